import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from tinyfusers_amd.ff.layer_norm import LayerNorm
from tinyfusers_amd.ff.linear import fold_layer_norm, linear_ln_f16, linear_f16
from tools.gemm_bench import time_call, st
for (m, n, k) in ((8192, 960, 320), (8192, 320, 320), (2048, 1920, 640), (2048, 640, 640), (512, 3840, 1280), (512, 1280, 1280)):
    x = T.DeviceArray.from_numpy((np.random.randn(m, k)).astype(np.float16))
    w = T.DeviceArray.from_numpy((np.random.randn(n, k) * k ** -0.5).astype(np.float16))
    ln = LayerNorm(k)
    f = fold_layer_norm(w, None, ln)
    y = T.DeviceArray.empty((m, n)); xn = T.DeviceArray.empty((m, k))
    with T.use_stream(st):
        linear_f16(x, w, f[1]); linear_ln_f16(x, f, ln.eps)      # autotune both
        t0 = time_call(lambda: hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, f[1].ptr, None, m, n, k, 0, None, 0, st.handle))
        t1 = time_call(lambda: hip.tf_linear_ln_f16(y.ptr, x.ptr, f[0].ptr, f[1].ptr, f[2].ptr, None, m, n, k, 0, 1e-5, st.handle))
        t2 = time_call(lambda: hip.tf_layer_norm_f16(xn.ptr, x.ptr, ln.weight.ptr, ln.bias.ptr, m, k, 1e-5, st.handle))
    print(f"M={m} N={n} K={k}: plain gemm {t0:6.1f} us | ln-folded gemm {t1:6.1f} us | separate LN kernel {t2:5.1f} us")
