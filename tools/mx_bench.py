#!/usr/bin/env python3
"""Per-shape timing of the block-scaled e4m3 conv / linear (k_igemm_pp<F8>, tf_conv2d_mx8 / tf_linear_mx8) on BASELINE config 5's shapes (4 images,
96 x 96 latents: UNet batch 8), the tile the shipped table picks and the other admissible ones.  GPU box only.   usage: tools/mx_bench.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.ff import fp8
from tinyfusers_amd.native import hip, lib
from tools.pp_bench import time_call, st

rng = np.random.default_rng(0)


def conv(n, hw, cin, cout, k, label):
    x = fp8.quantize_mx(T.DeviceArray.from_numpy((rng.standard_normal((n, cin, hw, hw)) * 0.7).astype(np.float16), np.float16, "nhwc"))
    wt = T.DeviceArray.from_numpy((rng.standard_normal((cout, cin, k, k)) * (cin * k * k) ** -0.5).astype(np.float16), np.float16, "nhwc")
    w8, sc = fp8.pack_weight(wt, {})
    b = T.DeviceArray.from_numpy(rng.standard_normal(cout).astype(np.float16), np.float16, "row")
    M, K = n * hw * hw, k * k * cin
    res = []
    with T.use_stream(st):
        for bm, bn, flags in ((192, 160, 512), (192, 128, 512), (256, 128, 512), (256, 160, 512), (192, 128, 2048)):      # 2048: the patch form, k_igemm_pp3<F8>
            lib.tf_gemm_force_config(bm, bn, 1); lib.tf_gemm_debug(flags)
            try:
                us = time_call(lambda: fp8.conv2d_mx(x, w8, sc, b, (cout, cin, k, k), [k // 2, k // 2]))
                res.append((us, "PP3-" + str(bm) if flags == 2048 else bm, bn))
            except RuntimeError:
                pass
            finally:
                lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    res.sort()
    print(f"{label:24s} M={M:6d} N={cout:5d} K={K:6d} | " + "  ".join(f"{bm}x{bn}: {us:7.1f} us {2.0 * M * cout * K / us / 1e6:5.0f} TF" for us, bm, bn in res), flush=True)


if __name__ == "__main__":
    conv(8, 96, 320, 320, 3, "conv3x3 320@96 (H2)")
    conv(8, 96, 960, 320, 3, "conv3x3 960->320@96 (H2)")
    conv(8, 96, 640, 320, 3, "conv3x3 640->320@96")
    conv(8, 48, 640, 640, 3, "conv3x3 640@48")
    conv(8, 24, 1280, 1280, 3, "conv3x3 1280@24")
    conv(8, 48, 1280, 640, 3, "conv3x3 1280->640@48")
    conv(8, 24, 2560, 1280, 3, "conv3x3 2560->1280@24")
    conv(8, 48, 640, 5120, 1, "lin 640->5120 @18432")
