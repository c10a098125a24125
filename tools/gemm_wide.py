import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from tools.gemm_bench import time_call, st


def run(m, n, k, cfgs, label, act=0):
    x = T.DeviceArray.from_numpy((np.random.randn(m, k) * 0.5).astype(np.float16))
    w = T.DeviceArray.from_numpy((np.random.randn(n * (2 if act else 1), k) * k ** -0.5).astype(np.float16))
    b = T.DeviceArray.from_numpy(np.random.randn(n * (2 if act else 1)).astype(np.float16))
    y = T.DeviceArray.empty((m, n))
    out = []
    for bm, bn in cfgs:
        lib.tf_gemm_force_config(bm, bn, 1)
        for mode, flag in (("deep", 8), ("wide", 16)):
            if mode == "wide" and (bm, bn) == (128, 160): continue
            lib.tf_gemm_debug(flag)
            t = time_call(lambda: hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, b.ptr, None, m, n, k, act, None, 0, st.handle))
            out.append(f"{bm}x{bn} {mode} {t:6.1f}")
    lib.tf_gemm_debug(0); lib.tf_gemm_force_config(0, 0, 0)
    t = time_call(lambda: hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, b.ptr, None, m, n, k, act, None, 0, st.handle))
    fl = 2.0 * m * n * (2 if act else 1) * k
    print(f"{label:26s} M={m} N={n} K={k}: auto {t:6.1f} us ({fl/t/1e6:5.0f} TF) | " + " | ".join(out), flush=True)

C = [(128, 160), (128, 128), (64, 160), (64, 128), (64, 64)]
run(8192, 2560, 320, C[:4], "geglu GEMM (plain out)")
run(8192, 1280, 320, [(128, 128), (64, 128), (64, 64)], "geglu fused", act=1)
run(8192, 960, 320, C, "qkv 64^2")
run(8192, 320, 320, C, "lin 320 64^2")
run(8192, 320, 1280, C, "ff2 64^2")
run(2048, 5120, 640, C[:4], "geglu 32^2 (plain)")
run(2048, 1920, 640, C, "qkv 32^2")
run(2048, 640, 640, C, "lin 640 32^2")
run(512, 10240, 1280, C[:4], "geglu 16^2 (plain)")
run(512, 3840, 1280, C, "qkv 16^2")
run(512, 1280, 1280, C, "lin 1280 16^2")
run(154, 24960, 768, C, "kv_all")
