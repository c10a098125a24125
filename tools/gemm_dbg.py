import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import _ablation; _ablation.use()   # the ablation bits exist only in the -DTF_ABLATION library
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from tools.gemm_bench import time_call, st


def run(n, h, w, cin, cout, k, bm, bn, sk, label):
    pad = k // 2
    x = T.DeviceArray.from_numpy(np.random.randn(n, cin, h, w).astype(np.float16) * 0.5)
    wt = T.DeviceArray.from_numpy((np.random.randn(cout, cin, k, k) * (cin * k * k) ** -0.5).astype(np.float16))
    b = T.DeviceArray.from_numpy(np.random.randn(cout).astype(np.float16))
    y = T.DeviceArray.empty((n, cout, h, w))
    ws = T.DeviceArray.empty((32 * n * h * w * cout * 4 + 16,), np.uint8, "row")
    lib.tf_gemm_force_config(bm, bn, sk)
    out = []
    for dbg in (0, 1, 2, 3, 4, 5, 6, 7):
        lib.tf_gemm_debug(dbg)
        def fn():
            hip.tf_conv2d_f16(y.ptr, x.ptr, None, wt.ptr, b.ptr, None, 0, None, n, h, w, cin, 0, cout, k, k, 1, pad, 0, ws.ptr, ws.nbytes, st.handle)
        out.append(time_call(fn))
    lib.tf_gemm_debug(0); lib.tf_gemm_force_config(0, 0, 0)
    print(f"{label:24s} {bm}x{bn}/{sk}: full {out[0]:6.1f} | nostore {out[1]:6.1f} | nomfma {out[2]:6.1f} | nomfma+nostore {out[3]:6.1f} | noload {out[4]:6.1f} | noload+nostore {out[5]:6.1f} | noload+nomfma {out[6]:6.1f} | nothing {out[7]:6.1f}")

if __name__ == "__main__":
    run(2, 64, 64, 320, 320, 1, 64, 160, 1, "conv1x1 320@64")
    run(2, 64, 64, 320, 320, 3, 64, 160, 1, "conv3x3 320@64")
    run(2, 64, 64, 320, 320, 3, 128, 160, 1, "conv3x3 320@64")
    run(2, 32, 32, 640, 640, 3, 128, 160, 4, "conv3x3 640@32")
    run(2, 32, 32, 640, 640, 1, 64, 64, 1, "conv1x1 640@32")
    run(2, 16, 16, 1280, 1280, 3, 128, 160, 8, "conv3x3 1280@16")
    print("--- small-K shapes")
    run(2, 64, 64, 320, 2560, 1, 128, 128, 1, "geglu 320->2560@64")
    run(2, 64, 64, 320, 2560, 1, 64, 128, 1, "geglu 320->2560@64")
    run(2, 64, 64, 320, 2560, 1, 128, 160, 1, "geglu 320->2560@64")
    run(2, 64, 64, 320, 960, 1, 128, 160, 1, "qkv 320->960@64")
    run(2, 64, 64, 320, 960, 1, 64, 160, 1, "qkv 320->960@64")
    run(2, 64, 64, 1280, 320, 1, 64, 160, 1, "ff2 1280->320@64")
    run(2, 32, 32, 640, 640, 1, 64, 128, 1, "lin 640@32")
    run(2, 16, 16, 1280, 1280, 1, 64, 64, 1, "lin 1280@16")
