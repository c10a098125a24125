#!/usr/bin/env python3
"""conv (+ split-K reduce) with and without the GroupNorm statistics riding along, graph-timed; plus the two GN kernels."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from tools.gemm_bench import time_call, st

CASES = [  # n, hw, cin, cout, k, (bm, bn, splitk)
    (2, 8, 1280, 1280, 3, (64, 160, 16)), (2, 16, 1280, 1280, 3, (128, 160, 8)), (2, 32, 640, 640, 3, (128, 160, 4)),
    (2, 64, 320, 320, 3, (64, 160, 1)), (2, 32, 640, 640, 1, (64, 128, 1)), (2, 16, 1280, 1280, 1, (64, 64, 1)),
]
for n, hw, cin, cout, k, force in CASES:
    x = T.DeviceArray.from_numpy(np.random.randn(n, cin, hw, hw).astype(np.float16) * 0.5)
    w = T.DeviceArray.from_numpy((np.random.randn(cout, cin, k, k) * (cin * k * k) ** -0.5).astype(np.float16))
    b = T.DeviceArray.from_numpy(np.random.randn(cout).astype(np.float16))
    y = T.DeviceArray.empty((n, cout, hw, hw))
    y2 = T.DeviceArray.empty((n, cout, hw, hw))
    gam = T.DeviceArray.from_numpy(np.ones(cout, np.float16), layout="row")
    M = n * hw * hw
    ws = T.DeviceArray.empty((32 * M * cout * 4 + 16,), np.uint8, "row")
    pb = hip.tf_conv2d_gn_partial_bytes(n, 32)
    part = T.DeviceArray.empty((pb,), np.uint8, "row")
    gws = T.DeviceArray.empty((hip.tf_group_norm_workspace(n, hw * hw, cout, 32),), np.uint8, "row")
    ch = ctypes.c_int(0)
    lib.tf_gemm_force_config(*force)
    args = (y.ptr, x.ptr, None, w.ptr, b.ptr, None, 0, None, n, hw, hw, cin, 0, cout, k, k, 1, k // 2, 0, ws.ptr, ws.nbytes)
    t0 = time_call(lambda: hip.tf_conv2d_f16(*args, st.handle))
    t1 = time_call(lambda: hip.tf_conv2d_fused_f16(*args, None, None, 0, 0, part.ptr, pb, 32, ctypes.byref(ch), st.handle))
    t2 = time_call(lambda: hip.tf_group_norm_f16(y2.ptr, y.ptr, None, gam.ptr, gam.ptr, n, hw * hw, cout, 0, 32, 1e-5, 1, gws.ptr, gws.nbytes, st.handle))
    t3 = time_call(lambda: hip.tf_group_norm_apply_f16(y2.ptr, y.ptr, gam.ptr, gam.ptr, part.ptr, max(ch.value, 1), n, hw * hw, cout, 32, 1e-5, 1, st.handle)) if ch.value else float("nan")
    lib.tf_gemm_force_config(0, 0, 0)
    print(f"M={M:5d} N={cout:4d} K={k*k*cin:5d} {force}: conv {t0:6.1f} us  conv+stats {t1:6.1f} us (chunks {ch.value:3d}) | GN stats+apply {t2:6.1f} us  apply only {t3:6.1f} us", flush=True)
