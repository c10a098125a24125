#!/usr/bin/env python3
"""k_igemm_pp3 (the patch form of the ping-pong kernel, tf_gemm_debug(2048)) against k_igemm_pp (512) and the deep-ring kernels on the 3x3
convolutions of BASELINE config 5 (4 images, 96 x 96 latents: UNet batch 8).  GPU box only.
usage: tools/pp3_bench.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pp_bench import T, hip, lib, st, time_call


def bench(n, hw, cin, cout, ups=0, c3=0):
    rng = np.random.default_rng(0)
    hs = hw >> ups
    x = T.DeviceArray.from_numpy((rng.standard_normal((n, cin, hs, hs)) * 0.5).astype(np.float16))
    wt = T.DeviceArray.from_numpy((rng.standard_normal((cout, 9 * cin + c3)) * (cin * 9) ** -0.5).astype(np.float16), np.float16, "row")     # KRSC rows (+ the 1x1 columns of the folded skip projection)
    b = T.DeviceArray.from_numpy(rng.standard_normal(cout).astype(np.float16))
    y = T.DeviceArray.empty((n, cout, hw, hw))
    x3 = T.DeviceArray.from_numpy((rng.standard_normal((n, c3, hw, hw)) * 0.5).astype(np.float16)) if c3 else None
    M, K = n * hw * hw, 9 * cin + c3
    ws = T.DeviceArray.empty((min(4 * M * cout * 4, 1 << 30) + 16,), np.uint8, "row")
    flops = 2.0 * M * cout * K

    def fn():
        hip.tf_conv2d_fused_f16(y.ptr, x.ptr, None, wt.ptr, b.ptr, None, 0, None, n, hs, hs, cin, 0, cout, 3, 3, 1, 1, ups, ws.ptr, ws.nbytes,
                                x3.ptr if c3 else None, None, c3, 0, None, 0, 0, None, st.handle)
    out = []
    cfgs = [(128, 160, 1, 8, 0), (256, 128, 1, 8, 0), (256, 128, 1, 512, 0), (128, 160, 1, 128, 0), (256, 160, 1, 512, 0), (192, 160, 1, 512, 0), (192, 128, 1, 512, 0), (192, 160, 2, 512, 0), (256, 160, 2, 512, 0),
            (192, 160, 1, 2048, 0), (192, 160, 1, 2048, 1)]      # (variant 6 takes its own tile width: 160 on 96-pixel rows, 128 on 48 / 24)
    for bm, bn, sk, flags, order in cfgs:
        lib.tf_gemm_force_config(bm, bn, sk); lib.tf_gemm_debug(flags | (64 if order else 32))
        try:
            us = time_call(fn)
        except RuntimeError:
            us = float("inf")
        finally:
            lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
        out.append(f"{ {8: 'deep', 128: 'patch', 512: 'pp', 2048: 'PP3'}[flags]}{bm}x{bn}/{sk}o{order}: {us:7.1f} us {flops / us / 1e6:5.0f} TF")
    print(f"conv3x3 {cin}->{cout} @{hw}{' (2x up-sampled input)' if ups else ''}{f' + 1x1 skip projection of {c3} channels' if c3 else ''} (M={M} K={K})\n   " + "\n   ".join(out), flush=True)


if __name__ == "__main__":
    bench(8, 96, 320, 320)
    bench(8, 96, 640, 320)
    bench(8, 48, 640, 640)
    bench(8, 48, 1280, 640)
    bench(8, 24, 1280, 1280)
    bench(8, 24, 2560, 1280)
    bench(8, 96, 320, 320, 0, 640)
    bench(8, 48, 640, 640, 0, 1280)
    bench(8, 24, 1280, 1280, 0, 2560)
    bench(8, 96, 640, 640, 1)
    bench(8, 48, 1280, 1280, 1)
