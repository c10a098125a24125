import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import _ablation; _ablation.use()   # the ablation bits exist only in the -DTF_ABLATION library
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from tools.gemm_bench import time_call, st

x = T.DeviceArray.from_numpy(np.random.randn(8192, 320).astype(np.float16))
y = T.DeviceArray.empty((8192, 320))
print("silu 64 elems      : %.2f us" % time_call(lambda: hip.tf_silu_f16(y.ptr, x.ptr, 64, st.handle), 200))
print("silu 2.6M elems    : %.2f us" % time_call(lambda: hip.tf_silu_f16(y.ptr, x.ptr, 8192 * 320, st.handle), 200))
w = T.DeviceArray.from_numpy(np.random.randn(320, 320).astype(np.float16) * 0.05)
for (m, n, k, bm, bn) in ((64, 160, 64, 64, 160), (64, 160, 320, 64, 160), (8192, 320, 64, 64, 160), (8192, 320, 320, 64, 160), (8192, 320, 320, 64, 64), (2048, 320, 320, 64, 64)):
    lib.tf_gemm_force_config(bm, bn, 1)
    for dbg in (0, 7):
        lib.tf_gemm_debug(dbg)
        t = time_call(lambda: hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, None, None, m, n, k, 0, None, 0, st.handle), 200)
        print(f"gemm M={m} N={n} K={k} tile {bm}x{bn} dbg={dbg}: {t:.2f} us")
lib.tf_gemm_debug(0); lib.tf_gemm_force_config(0, 0, 0)
