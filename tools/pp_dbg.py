#!/usr/bin/env python3
"""Ablation of k_igemm_pp on one conv shape (GPU box): where does a K tile's time go?  tf_gemm_debug bits: 1 no epilogue, 2 no MFMA,
4 no LDS-DMA inside the loop, 4096 no fragment reads (all on the ablation instance of the kernel; 512 selects the kernel)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import _ablation; _ablation.use()   # the ablation bits exist only in the -DTF_ABLATION library
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from tools.pp_bench import time_call, st


def run(n, hw, cin, cout, k, bn, sk, label, extra=0):
    pad = k // 2
    rng = np.random.default_rng(0)
    x = T.DeviceArray.from_numpy((rng.standard_normal((n, cin, hw, hw)) * 0.5).astype(np.float16))
    wt = T.DeviceArray.from_numpy((rng.standard_normal((cout, cin, k, k)) * (cin * k * k) ** -0.5).astype(np.float16))
    b = T.DeviceArray.from_numpy(rng.standard_normal(cout).astype(np.float16))
    y = T.DeviceArray.empty((n, cout, hw, hw))
    ws = T.DeviceArray.empty((min(4 * n * hw * hw * cout * 4, 1 << 30) + 16,), np.uint8, "row")
    M, K = n * hw * hw, k * k * cin
    tiles = ((M + 255) // 256) * ((cout + bn - 1) // bn) * sk
    rounds = (tiles + 255) // 256
    kt = (K // 64) // sk

    def fn():
        hip.tf_conv2d_f16(y.ptr, x.ptr, None, wt.ptr, b.ptr, None, 0, None, n, hw, hw, cin, 0, cout, k, k, 1, pad, 0, ws.ptr, ws.nbytes, st.handle)
    names = [("full", 0), ("no epilogue", 1), ("no MFMA", 2), ("no DMA", 4), ("no reads", 4096), ("no MFMA, no DMA", 6), ("no reads, no MFMA", 4098), ("no reads, no DMA", 4100),
             ("barriers only", 4102)]
    lib.tf_gemm_force_config(256, bn, sk)
    out = []
    try:
        for nm, f in names:
            lib.tf_gemm_debug(512 | f | extra)
            out.append((nm, time_call(fn)))
    finally:
        lib.tf_gemm_debug(0); lib.tf_gemm_force_config(0, 0, 0)
    print(f"{label} 256x{bn}/{sk}{' one phase per k-step' if extra else ''}: {tiles} blocks = {rounds} rounds x {kt} K tiles")
    for nm, us in out:
        print(f"   {nm:22s} {us:8.1f} us   {us * 1e3 / (rounds * kt):7.1f} ns per K tile per round")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "short":
        run(8, 96, 320, 2560, 1, 128, 1, "lin 320->2560 @73728")
        run(8, 96, 320, 320, 1, 160, 1, "conv1x1 320@96")
        run(8, 48, 640, 640, 1, 128, 1, "conv1x1 640@48")
        sys.exit(0)
    run(8, 96, 320, 320, 3, 160, 1, "conv3x3 320@96")
    run(8, 96, 320, 320, 3, 160, 1, "conv3x3 320@96", 8192)
    run(8, 48, 640, 640, 3, 128, 1, "conv3x3 640@48")
    run(8, 24, 1280, 1280, 3, 256, 2, "conv3x3 1280@24")
