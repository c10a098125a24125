import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from tools.gemm_bench import time_call, st
x = T.DeviceArray.from_numpy(np.random.randn(8192, 320).astype(np.float16))
y = T.DeviceArray.empty((8192, 320)); z = T.DeviceArray.empty((8192, 320))
g = T.DeviceArray.from_numpy(np.ones(320, np.float16)); b = T.DeviceArray.from_numpy(np.zeros(320, np.float16))
w = T.DeviceArray.from_numpy((np.random.randn(320, 320) * 0.05).astype(np.float16))
ws = T.DeviceArray.empty((hip.tf_group_norm_workspace(2, 4096, 320, 32),), np.uint8, "row") if False else None
nb = lib.tf_group_norm_workspace(2, 4096, 320, 32)
ws = T.DeviceArray.empty((nb,), np.uint8, "row")
s = st.handle
ln = lambda: hip.tf_layer_norm_f16(y.ptr, x.ptr, g.ptr, b.ptr, 8192, 320, 1e-5, s)
silu = lambda: hip.tf_silu_f16(z.ptr, y.ptr, 8192 * 320, s)
add = lambda: hip.tf_add_f16(z.ptr, x.ptr, y.ptr, 8192 * 320, s)
gn = lambda: hip.tf_group_norm_f16(z.ptr, x.ptr, None, g.ptr, b.ptr, 2, 4096, 320, 0, 32, 1e-5, 1, ws.ptr, nb, s)
gemm = lambda: hip.tf_linear_f16(z.ptr, x.ptr, w.ptr, b.ptr, None, 8192, 320, 320, 0, None, 0, s)
print("LN alone            %.2f us" % time_call(ln))
print("silu alone          %.2f us" % time_call(silu))
print("add alone           %.2f us" % time_call(add))
print("GN (2 kernels)      %.2f us" % time_call(gn))
print("gemm 8192x320x320   %.2f us" % time_call(gemm))
def mix():
    ln(); silu(); add()
print("LN+silu+add         %.2f us (sum of 3)" % time_call(mix))
def mix2():
    ln(); gemm(); gn()
print("LN+gemm+GN          %.2f us (sum of 4 kernels)" % time_call(mix2))
