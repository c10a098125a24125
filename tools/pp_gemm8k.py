#!/usr/bin/env python3
"""How good is k_igemm_pp as a GEMM kernel?  The programming guide's reference 256 x 256 template (8 waves, LDS-DMA ring, counted vmcnt, raw barriers)
delivers 1.32-1.34 PFLOP/s at 4096^3 and ~1.47 at 8192^3 on random bf16 operands.  This runs the ping-pong kernel on the same problems (fp16, random
operands, 256 x 256 / 256 x 160 / 192 x 128 tiles) -- the comparison separates "the kernel" from "the tiles the UNet's channel counts allow"."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from tools.pp_bench import time_call, st

rng = np.random.default_rng(0)
for size in (4096, 8192):
    M = N = K = size
    x = T.DeviceArray.from_numpy(rng.uniform(-1, 1, (M, K)).astype(np.float16), np.float16, "row")
    w = T.DeviceArray.from_numpy(rng.uniform(-1, 1, (N, K)).astype(np.float16), np.float16, "row")
    y = T.DeviceArray.empty((M, N), np.float16, "row")

    def fn():
        hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, None, None, M, N, K, 0, None, 0, st.handle)
    for bm, bn in ((256, 256), (256, 160), (256, 128), (192, 160), (192, 128)):
        lib.tf_gemm_force_config(bm, bn, 1); lib.tf_gemm_debug(512)
        try:
            us = time_call(fn, reps=3)
            bytes_per_ktile = (bm + bn) * 128
            flop_per_byte = 2.0 * bm * bn * 64 / bytes_per_ktile
            blocks = -(-M // bm) * -(-N // bn)
            rounds = -(-blocks // 256)
            ingest = rounds * (K // 64) * bytes_per_ktile / (us * 1e-6) / 1e9      # GB/s per CU while it has a block
            print(f"{size}^3 tile {bm}x{bn}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s | {flop_per_byte:5.1f} FLOP per ingested byte, {blocks} blocks = {rounds} rounds, "
                  f"{ingest:5.1f} GB/s LDS-DMA ingest per CU", flush=True)
        except RuntimeError as e:
            print(f"{size}^3 tile {bm}x{bn}: {e}")
        finally:
            lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
