#!/usr/bin/env python3
"""What the LayerNorm fold costs inside k_gemm_c4 (GPU box): the transformer blocks' LN-consuming linears at config 5 and batch 1, plain
(tf_linear_f16) against folded (tf_linear_ln_f16), both tile orders.  Measured: +3 ... +12 % at config 5, +14 ... +20 % at batch 1.   usage: tools/ln_c4_bench.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from pp_bench import time_call, st
rng = np.random.default_rng(0)
for (m, n, k) in ((73728, 2560, 320), (18432, 5120, 640), (4608, 10240, 1280), (73728, 960, 320), (8192, 2560, 320)):
    act = 1 if n >= 2560 else 0
    no = n // 2 if act else n
    x = T.DeviceArray.from_numpy((rng.standard_normal((m, k)) * 0.5).astype(np.float16))
    w = T.DeviceArray.from_numpy((rng.standard_normal((n, k)) * k ** -0.5).astype(np.float16))
    b = T.DeviceArray.from_numpy(rng.standard_normal(n).astype(np.float16))
    cs = T.DeviceArray.from_numpy(rng.standard_normal(n).astype(np.float32), np.float32, "row")
    y = T.DeviceArray.empty((m, no))
    out = []
    for name, fn in (("plain", lambda: hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, b.ptr, None, m, no, k, act, None, 0, st.handle)),
                     ("ln", lambda: hip.tf_linear_ln_f16(y.ptr, x.ptr, w.ptr, b.ptr, cs.ptr, None, m, no, k, act, 1e-5, st.handle))):
        for flags, tag in ((1024, "c4"), (1024 | 64, "c4m")):
            lib.tf_gemm_force_config(128, 128, 1); lib.tf_gemm_debug(flags)
            try:
                out.append(f"{name}/{tag}:{time_call(fn):.1f}")
            finally:
                lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    print(m, n, k, " ".join(out), flush=True)
