#!/usr/bin/env python3
"""Per-shape timing of the persistent short-K kernel (k_gemm_c4, tf_gemm_debug(1024)) against the other kernels of the family on the
transformer blocks' linears (q/k/v, out-projection, GEGLU, FF2; ff/linear.py:112-121, ff/nn.py:5-23) at BASELINE config 2 (UNet batch 2,
64 x 64 latents) and config 5 (UNet batch 8, 96 x 96 latents).  GPU box only.
usage: tools/c4_bench.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from pp_bench import time_call, st


def bench_linear(m, n, k, act, label, cfgs):
    rng = np.random.default_rng(0)
    x = T.DeviceArray.from_numpy((rng.standard_normal((m, k)) * 0.5).astype(np.float16))
    w = T.DeviceArray.from_numpy((rng.standard_normal((n, k)) * k ** -0.5).astype(np.float16))
    b = T.DeviceArray.from_numpy(rng.standard_normal(n).astype(np.float16))
    no = n // 2 if act else n
    y = T.DeviceArray.empty((m, no))
    flops = 2.0 * m * n * k

    cs = T.DeviceArray.from_numpy(rng.standard_normal(n).astype(np.float32), np.float32, "row")

    def fn():
        if os.environ.get("C4_LN"):                        # the LayerNorm-folded form (tf_linear_ln_16): what the step launches for q|k|v, to_q and the GEGLU projection
            hip.tf_linear_ln_16(0, y.ptr, x.ptr, w.ptr, b.ptr, cs.ptr, None, m, no, k, act, 1e-5, st.handle)
        else:
            hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, b.ptr, None, m, no, k, act, None, 0, st.handle)   # (GEGLU: N is the output width, w holds 2 N rows)
    res = []
    for (bm, bn, sk, flags) in cfgs:
        lib.tf_gemm_force_config(bm, bn, sk); lib.tf_gemm_debug(flags)
        if os.environ.get('C4_TRACE'): print('  try', label, bm, bn, sk, flags, file=sys.stderr, flush=True)
        try:
            res.append((time_call(fn), bm, bn, sk, flags))
        except RuntimeError:
            res.append((float("inf"), bm, bn, sk, flags))
        finally:
            lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    name = {8: "deep", 16: "wide", 512: "PP", 1024: "C4", 1024 | 64: "C4m", 16384: "C8", 16384 | 64: "C8m", 32768: "AR"}
    base = min(r for r in res if not (r[4] & (1024 | 16384 | 32768)))
    c4 = min(r for r in res if r[4] & 1024)
    c8 = min(r for r in res if r[4] & 16384)
    ar = min([r for r in res if r[4] & 32768] or [(float("inf"),)])
    byt = 2.0 * (m * k + n * k + m * no)
    print(f"{label:26s} M={m:6d} N={n:5d} K={k:5d} | best other {base[0]:7.1f} us {flops/base[0]/1e6:5.0f} TF {base[1]}x{base[2]} {name[base[4]]:5s} | C4 {c4[0]:7.1f} us {flops/c4[0]/1e6:5.0f} TF "
          f"{byt/c4[0]/1e3:5.0f} GB/s | C8 {c8[0]:7.1f} us {flops/c8[0]/1e6:5.0f} TF | AR {ar[0]:7.1f} us {flops/ar[0]/1e6:5.0f} TF | " + " ".join(f"{name[f]}{bm}x{bn}:{us:.1f}" for us, bm, bn, sk, f in sorted(res)), flush=True)


if __name__ == "__main__":
    C = [(128, 128, 1, 16), (128, 128, 1, 8), (64, 128, 1, 16), (128, 64, 1, 16), (256, 128, 1, 512), (256, 160, 1, 512), (192, 128, 1, 512), (128, 128, 1, 1024), (128, 128, 1, 1024 | 64), (256, 128, 1, 16384), (256, 128, 1, 16384 | 64), (128, 128, 1, 32768)]
    if os.environ.get("C4_AR_ONLY"):
        C = [(128, 128, 1, 16), (128, 128, 1, 1024), (256, 128, 1, 16384), (128, 128, 1, 32768)]
    for tag, s in (("c2", 1), ("c5", 9)):
        if os.environ.get("C4_ONLY") and os.environ["C4_ONLY"] != tag:
            continue
        m0 = 8192 * s
        if os.environ.get("C4_K320"):                     # the K = 320 shapes only (k_gemm_ar's)
            bench_linear(m0, 2560, 320, 1, f"{tag} geglu 320", C)
            bench_linear(m0, 960, 320, 0, f"{tag} qkv 320", C)
            bench_linear(m0, 320, 320, 0, f"{tag} out 320", C)
            continue
        bench_linear(m0, 2560, 320, 1, f"{tag} geglu 320", C)
        bench_linear(m0, 320, 1280, 0, f"{tag} ff2 1280->320", C)
        bench_linear(m0, 960, 320, 0, f"{tag} qkv 320", C)
        bench_linear(m0, 320, 320, 0, f"{tag} out 320", C)
        bench_linear(m0 // 4, 5120, 640, 1, f"{tag} geglu 640", C)
        bench_linear(m0 // 4, 640, 2560, 0, f"{tag} ff2 2560->640", C)
        bench_linear(m0 // 4, 1920, 640, 0, f"{tag} qkv 640", C)
        bench_linear(m0 // 16, 10240, 1280, 1, f"{tag} geglu 1280", C)
        bench_linear(m0 // 16, 1280, 5120, 0, f"{tag} ff2 5120->1280", C)
        bench_linear(m0 // 16, 3840, 1280, 0, f"{tag} qkv 1280", C)
        bench_linear(m0 // 4, 640, 640, 0, f"{tag} out 640", C)
        bench_linear(m0 // 16, 1280, 1280, 0, f"{tag} out 1280", C)
        bench_linear(m0, 320, 1600, 0, f"{tag} ff2.proj_out 320", C)
        bench_linear(m0 // 4, 640, 3200, 0, f"{tag} ff2.proj_out 640", C)
        bench_linear(m0 // 16, 1280, 6400, 0, f"{tag} ff2.proj_out 1280", C)
