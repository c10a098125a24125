#!/usr/bin/env python3
"""GEGLU / plain linear GEMMs of the transformer FF at the three levels, per tile config, graph-timed."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from tools.gemm_bench import time_call, st

for M, C in ((8192, 320), (2048, 640), (512, 1280)):
    N = 4 * C
    x = T.DeviceArray.from_numpy(np.random.randn(M, C).astype(np.float16), layout="row")
    w = T.DeviceArray.from_numpy((np.random.randn(2 * N, C) * C ** -0.5).astype(np.float16), layout="row")
    b = T.DeviceArray.from_numpy(np.random.randn(2 * N).astype(np.float16), layout="row")
    y = T.DeviceArray.empty((M, 2 * N), np.float16, "row")
    fl = 2.0 * M * 2 * N * C
    for act, nn in ((1, N), (0, 2 * N)):
        res = []
        for bm, bn in ((128, 128), (64, 128), (128, 64), (64, 64), (128, 160), (64, 160)):
            if act == 1 and bn % 64:
                continue
            for wide in (8, 16):
                lib.tf_gemm_force_config(bm, bn, 1)
                lib.tf_gemm_debug(wide)
                try:
                    us = time_call(lambda: hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, b.ptr, None, M, nn, C, act, None, 0, st.handle))
                except RuntimeError:
                    continue
                res.append((us, bm, bn, "wide" if wide == 16 else "deep"))
        lib.tf_gemm_debug(0); lib.tf_gemm_force_config(0, 0, 0)
        auto = time_call(lambda: hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, b.ptr, None, M, nn, C, act, None, 0, st.handle))
        res.sort()
        print(f"M={M} N={2*N} K={C} act={act}: auto {auto:.1f} us ({fl/auto/1e6:.0f} TF) | " + " ".join(f"{bm}x{bn}{v}:{us:.1f}" for us, bm, bn, v in res[:6]), flush=True)
