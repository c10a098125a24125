#!/usr/bin/env python3
"""Per-shape timing of the e4m3 kernels on BASELINE config 5's GEMM shapes: k_igemm8 (non-scaled fp8 MFMA, the fp16 rate) against
k_igemm_pp's e4m3 form (v_mfma_scale_f32_16x16x128_f8f6f4, twice the rate).  GPU box only.   usage: tools/pp_bench8.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.ff import fp8
from tinyfusers_amd.native import hip, lib
from tools.pp_bench import time_call, st


FLUSH = None


def bench(n, hw, cin, cout, k, label, cfgs, pp_cfgs=(), gn=0, flush=False):
    global FLUSH
    cfgs = list(cfgs) + list(pp_cfgs)
    if flush and FLUSH is None:
        FLUSH = T.DeviceArray.empty((384 << 20,), np.uint8, "row")
    pad = k // 2
    rng = np.random.default_rng(0)
    x8 = fp8.quantize(T.DeviceArray.from_numpy((rng.standard_normal((n, cin, hw, hw))).astype(np.float16)))
    wt = T.DeviceArray.from_numpy((rng.standard_normal((cout, cin, k, k)) * (cin * k * k) ** -0.5).astype(np.float16))
    w8, sc = fp8.pack_weight(wt, {})
    b = T.DeviceArray.from_numpy(rng.standard_normal(cout).astype(np.float16), np.float16, "row")
    y = T.DeviceArray.empty((n, cout, hw, hw))
    M, K = n * hw * hw, k * k * cin
    ws = T.DeviceArray.empty((min(4 * M * cout * 4, 1 << 30) + 16,), np.uint8, "row")
    flops = 2.0 * M * cout * K
    chunks = ctypes.c_int(0)
    e = T.DeviceArray.from_numpy(rng.standard_normal((n, cout)).astype(np.float16), np.float16, "row")
    pb = hip.tf_conv2d_gn_partial_bytes(n, gn) if gn else 0
    part = T.DeviceArray.empty((pb,), np.uint8, "row") if gn else None

    def fn():
        if flush:
            hip.tf_memset_async(FLUSH.ptr, 1, FLUSH.nbytes, st.handle)
        hip.tf_conv2d_fp8(y.ptr, x8.ptr, None, w8.ptr, sc.ptr, b.ptr, e.ptr if gn else None, cout if gn else 0, None, n, hw, hw, cin, 0, cout, k, k, 1, pad, 0, ws.ptr, ws.nbytes,
                          part.ptr if gn else None, pb, gn, ctypes.byref(chunks), st.handle)
    res = []
    for (bm, bn, sk, flags) in cfgs:
        lib.tf_gemm_force_config(bm, bn, sk); lib.tf_gemm_debug(flags)
        try:
            res.append((time_call(fn), bm, bn, sk, flags))
        except RuntimeError:
            res.append((float("inf"), bm, bn, sk, flags))
        finally:
            lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    print(f"{label:24s} M={M:6d} N={cout:5d} K={K:6d} | " + " ".join(f"{'PP' if f == 512 else 'k8'}{bm}x{bn}/{sk}:{us:.0f}us={flops / us / 1e6:.0f}TF" for us, bm, bn, sk, f in sorted(res)), flush=True)


if __name__ == "__main__":
    K8 = [(128, 128, 1, 0), (256, 64, 1, 0), (128, 64, 1, 0)]
    P = lambda *bns: [(bm, bn, sk, 512) for bm in (256, 192) for bn in bns if not (bm == 192 and bn == 256) for sk in (1, 2)]
    with T.use_stream(st):
        bench(8, 96, 320, 320, 3, "conv3x3 320@96", K8, P(160, 128))
        bench(8, 96, 320, 320, 3, "  + emb + gn stats", K8, P(160, 128), gn=32)
        bench(8, 96, 320, 320, 3, "  + flush (incl. ~70us memset)", K8, P(160, 128), gn=32, flush=True)
        bench(8, 48, 640, 640, 3, "conv3x3 640@48", K8, P(160, 128))
        bench(8, 24, 1280, 1280, 3, "conv3x3 1280@24", K8, P(160, 128))
        bench(8, 96, 640, 320, 3, "conv3x3 640->320@96", K8, P(160))
        bench(8, 96, 320, 2560, 1, "lin 320->2560 @73728", K8, P(160, 128))
        bench(8, 48, 2560, 640, 1, "lin 2560->640 @18432", K8, P(160, 128))
