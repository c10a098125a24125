"""3x3 convolutions of the SD-1.x step: tap-by-tap kernel (deep ring) vs k_igemm_patch on the same forced tile, graph-timed.
    python tools/patch_bench.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from tools.gemm_bench import time_call, st


def run(n, hw, cin, cout, bm, bn, sk, label, c3=0):
    x = T.DeviceArray.from_numpy(np.random.randn(n, cin, hw, hw).astype(np.float16) * 0.5)
    k = 9 * cin + c3
    wt = T.DeviceArray.from_numpy((np.random.randn(cout, k) * k ** -0.5).astype(np.float16), layout="row")
    b = T.DeviceArray.from_numpy(np.random.randn(cout).astype(np.float16))
    x3 = T.DeviceArray.from_numpy(np.random.randn(n, c3, hw, hw).astype(np.float16)) if c3 else None
    y = T.DeviceArray.empty((n, cout, hw, hw))
    ws = T.DeviceArray.empty((32 * n * hw * hw * cout * 4 + 16,), np.uint8, "row")
    lib.tf_gemm_force_config(bm, bn, sk)
    out = {}
    for flags in (8, 128, 256):
        lib.tf_gemm_debug(flags)
        def fn():
            hip.tf_conv2d_fused_f16(y.ptr, x.ptr, None, wt.ptr, b.ptr, None, 0, None, n, hw, hw, cin, 0, cout, 3, 3, 1, 1, 0, ws.ptr, ws.nbytes,
                                    x3.ptr if c3 else None, None, c3, 0, None, 0, 0, None, st.handle)
        out[flags] = time_call(fn)
    lib.tf_gemm_debug(0); lib.tf_gemm_force_config(0, 0, 0)
    fl = 2.0 * n * hw * hw * cout * k
    print(f"{label:22s} {bm:3d}x{bn}/{sk:<2d}: taps {out[8]:6.1f} us ({fl / out[8] / 1e6:6.1f} TF)   patch {out[128]:6.1f} us {out[8] / out[128]:.3f}x   all8 {out[256]:6.1f} us {out[8] / out[256]:.3f}x", flush=True)


if __name__ == "__main__":
    run(2, 64, 320, 320, 64, 160, 1, "320@64")
    run(2, 64, 320, 320, 128, 160, 1, "320@64")
    run(2, 64, 640, 320, 128, 160, 2, "640->320@64")
    run(2, 64, 960, 320, 128, 160, 2, "960->320@64")
    run(2, 64, 320, 320, 64, 160, 1, "320@64 + skip 640", c3=640)
    run(2, 32, 640, 640, 128, 160, 4, "640@32")
    run(2, 32, 640, 640, 64, 160, 2, "640@32")
    run(2, 32, 320, 640, 128, 160, 2, "320->640@32")
    run(2, 32, 1280, 640, 128, 160, 4, "1280->640@32")
    run(2, 16, 1280, 1280, 128, 160, 8, "1280@16")
    run(2, 16, 1280, 1280, 64, 160, 4, "1280@16")
    run(2, 16, 2560, 1280, 128, 160, 8, "2560->1280@16")
    run(2, 8, 1280, 1280, 64, 160, 16, "1280@8")
    run(2, 8, 2560, 1280, 64, 160, 16, "2560->1280@8")
