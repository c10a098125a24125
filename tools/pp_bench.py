#!/usr/bin/env python3
"""Per-shape timing of the 256-row ping-pong kernel (k_igemm_pp, tf_gemm_debug(512)) against the other kernels of the family on
BASELINE config 5's GEMM shapes (4 images, 96 x 96 latents: UNet batch 8).  GPU box only.
usage: tools/pp_bench.py [quick|all]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib

T.ensure_init(0)
st = T.Stream()


def time_call(fn, reps=6):
    ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
    hip.tf_event_create(ctypes.byref(ev0)); hip.tf_event_create(ctypes.byref(ev1))
    fn()
    hip.tf_stream_sync(st.handle)
    hip.tf_graph_begin_capture(st.handle)
    for _ in range(reps): fn()
    g = ctypes.c_void_p()
    hip.tf_graph_end_capture(st.handle, ctypes.byref(g))
    hip.tf_graph_launch(g, st.handle)
    hip.tf_event_record(ev0, st.handle)
    for _ in range(3): hip.tf_graph_launch(g, st.handle)
    hip.tf_event_record(ev1, st.handle)
    hip.tf_stream_sync(st.handle)
    ms = ctypes.c_float(); hip.tf_event_elapsed_ms(ctypes.byref(ms), ev0, ev1)
    hip.tf_graph_destroy(g)
    return ms.value * 1e3 / (reps * 3)


def bench_conv(n, hw, cin, cout, k, label, base_cfgs, pp_cfgs):
    pad = k // 2
    rng = np.random.default_rng(0)
    x = T.DeviceArray.from_numpy((rng.standard_normal((n, cin, hw, hw)) * 0.5).astype(np.float16))
    wt = T.DeviceArray.from_numpy((rng.standard_normal((cout, cin, k, k)) * (cin * k * k) ** -0.5).astype(np.float16))
    b = T.DeviceArray.from_numpy(rng.standard_normal(cout).astype(np.float16))
    y = T.DeviceArray.empty((n, cout, hw, hw))
    M, K = n * hw * hw, k * k * cin
    ws = T.DeviceArray.empty((min(4 * M * cout * 4, 1 << 30) + 16,), np.uint8, "row")
    flops = 2.0 * M * cout * K

    def fn():
        hip.tf_conv2d_f16(y.ptr, x.ptr, None, wt.ptr, b.ptr, None, 0, None, n, hw, hw, cin, 0, cout, k, k, 1, pad, 0, ws.ptr, ws.nbytes, st.handle)
    res = []
    for (bm, bn, sk, flags) in base_cfgs + pp_cfgs:
        lib.tf_gemm_force_config(bm, bn, sk); lib.tf_gemm_debug(flags)
        try:
            us = time_call(fn)
            res.append((us, bm, bn, sk, flags))
        except RuntimeError as e:
            res.append((float("inf"), bm, bn, sk, flags))
        finally:
            lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    name = {8: "deep", 16: "wide", 128: "patch", 256: "all8", 512: "PP"}
    base = min(r for r in res if r[4] != 512)
    pp = min(r for r in res if r[4] == 512)
    print(f"{label:26s} M={M:6d} N={cout:5d} K={K:6d} | best other {base[0]:8.1f} us {flops/base[0]/1e6:6.0f} TF {base[1]}x{base[2]}/{base[3]} {name[base[4]]:5s} | best PP {pp[0]:8.1f} us {flops/pp[0]/1e6:6.0f} TF {pp[1]}x{pp[2]}/{pp[3]} | " +
          " ".join(f"{name[f]}{bm}x{bn}/{sk}:{us:.0f}" for us, bm, bn, sk, f in sorted(res)), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    B3 = [(128, 128, 1, 256), (128, 160, 1, 8), (128, 160, 1, 128), (256, 128, 1, 8), (128, 128, 1, 16)]
    P = lambda *bns: [(bm, bn, sk, 512) for bm in (256, 192) for bn in bns if not (bm == 192 and bn == 256) for sk in (1, 2)]
    bench_conv(8, 96, 320, 320, 3, "conv3x3 320@96", B3, P(160))
    bench_conv(8, 48, 640, 640, 3, "conv3x3 640@48", B3, P(160, 128))
    bench_conv(8, 24, 1280, 1280, 3, "conv3x3 1280@24", B3 + [(128, 64, 1, 16)], P(160, 128, 256))
    bench_conv(8, 96, 320, 320, 1, "conv1x1 320@96", [(128, 128, 1, 16), (128, 64, 1, 16)], P(160))
    bench_conv(8, 96, 320, 2560, 1, "lin 320->2560 @73728", [(128, 128, 1, 16)], P(160, 128, 256))
    if mode == "all":
        bench_conv(8, 96, 640, 320, 3, "conv3x3 640->320@96", B3, P(160))
        bench_conv(8, 48, 1280, 640, 3, "conv3x3 1280->640@48", B3, P(160, 128))
        bench_conv(8, 48, 640, 5120, 1, "lin 640->5120 @18432", [(128, 128, 1, 16)], P(160, 128, 256))
        bench_conv(8, 24, 1280, 10240, 1, "lin 1280->10240 @4608", [(128, 128, 1, 16)], P(160, 128, 256))
        bench_conv(8, 96, 1600, 320, 1, "lin 1600->320 @73728", [(128, 128, 1, 16)], P(160))
        bench_conv(8, 48, 640, 640, 1, "conv1x1 640@48", [(128, 128, 1, 16), (128, 64, 1, 16)], P(160, 128))
