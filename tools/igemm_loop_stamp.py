#!/usr/bin/env python3
"""Inside the K loop of k_igemm (round 5): per-K-tile stamps (s_memtime, core clock cycles) of block 0's loader wave 4 and consumer wave 0, from the diagnostic build

    python -m tinyfusers_amd.build --tag stamp2 -DTF_IGEMM_STAMP=2
    TF_LIB_PATH=tinyfusers_amd/lib/libtinyfusers_hip_stamp2.so TF_LIB_ALLOW_MISSING=1 python tools/igemm_loop_stamp.py

Loader per tile: [wait] top -> tile it + 1 landed (counted vmcnt) | [barrier] -> behind barrier(it) | [issue] -> the stage of tile it + NS issued | [rest] -> next top.
Consumer per tile: [barrier] fragments of tile it in registers -> behind barrier(it) | [read + MFMA] -> fragments of tile it + 1 in registers.
GPU box only; the shipped library holds no stamp code."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pp_bench import T, hip, lib, st

assert "stamp2" in os.environ.get("TF_LIB_PATH", "")
rng = np.random.default_rng(0)
flush = T.DeviceArray.empty((384 << 20,), np.uint8, "row")


def read():
    raw = np.zeros((256, 8), np.uint32)
    assert lib.tf_debug_loop_stamps(raw.ctypes.data_as(ctypes.c_void_p)) == 0
    return raw.astype(np.int64)


def report(label, fn, nt, cold):
    for _ in range(3):
        fn()
    hip.tf_stream_sync(st.handle)
    import time
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(20):
            fn()
        hip.tf_stream_sync(st.handle)
    if cold:
        hip.tf_memset_async(flush.ptr, 1, flush.nbytes, st.handle)
    fn()
    hip.tf_stream_sync(st.handle)
    s = read()[: max(0, min(256, nt - 4) - 2)]
    if len(s) < 3:
        print(f"{label}: too few K tiles"); return
    d = lambda a, b: ((b - a) & 0xffffffff).astype(np.float64)
    lw, lb, li = d(s[:, 0], s[:, 1]), d(s[:, 1], s[:, 2]), d(s[:, 2], s[:, 3])
    lr = d(s[:-1, 3], s[1:, 0])
    cb = d(s[:, 4], s[:, 5])
    cm = d(s[:-1, 5], s[1:, 4])
    per = d(s[:-1, 0], s[1:, 0])
    m = lambda a: f"{np.median(a):6.0f} (p10 {np.percentile(a, 10):5.0f}, p90 {np.percentile(a, 90):5.0f})"
    print(f"=== {label}, {'cold' if cold else 'warm'}: {len(s)} K tiles of block 0; cycles per tile {m(per)}")
    print(f"  loader  : wait for tile it+1 {m(lw)} | barrier {m(lb)} | issue stage it+NS {m(li)} | rest {m(lr)}")
    print(f"  consumer: barrier {m(cb)} | fragment reads + MFMAs {m(cm)}")


def conv(n, hw, cin, cout, bm, bn, split, flag):
    x = T.DeviceArray.from_numpy((rng.standard_normal((n, cin, hw, hw)) * 0.5).astype(np.float16))
    wt = T.DeviceArray.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * (cin * 9) ** -0.5).astype(np.float16))
    y = T.DeviceArray.empty((n, cout, hw, hw))
    nb = hip.tf_conv2d_workspace(n, hw, hw, cin, 0, cout, 3, 3, 1, 1, 0)
    ws = T.DeviceArray.empty((max(nb, 16),), np.uint8, "row")

    def fn():
        lib.tf_gemm_force_config(bm, bn, split); lib.tf_gemm_debug(flag | 32)
        hip.tf_conv2d_f16(y.ptr, x.ptr, None, wt.ptr, None, None, 0, None, n, hw, hw, cin, 0, cout, 3, 3, 1, 1, 0, ws.ptr, nb, st.handle)
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    nt = 9 * cin // 64 // split
    for cold in (False, True):
        report(f"conv3x3 {cin}->{cout} @{hw} batch {n}, tile {bm}x{bn}, split {split}, flag {flag}", fn, nt, cold)
    return x, wt, y, ws


def linear(M, N, K, bm, bn, flag):
    x = T.DeviceArray.from_numpy((rng.standard_normal((M, K)) * 0.5).astype(np.float16), np.float16, "row")
    w = T.DeviceArray.from_numpy((rng.standard_normal((N, K)) * K ** -0.5).astype(np.float16), np.float16, "row")
    y = T.DeviceArray.empty((M, N), np.float16, "row")

    def fn():
        lib.tf_gemm_force_config(bm, bn, 1); lib.tf_gemm_debug(flag | 32)
        hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, None, None, M, N, K, 0, None, 0, st.handle)
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    for cold in (False, True):
        report(f"linear {M} x {N} x {K}, tile {bm}x{bn}, flag {flag}", fn, K // 64, cold)
    return x, w, y


if __name__ == "__main__":
    keep = [conv(2, 64, 320, 320, 64, 160, 1, 8), conv(2, 64, 320, 320, 64, 160, 1, 256), conv(2, 32, 640, 640, 64, 128, 1, 256), conv(2, 16, 1280, 1280, 64, 160, 4, 256),
            linear(512, 1280, 5120, 64, 64, 256), linear(512, 1280, 5120, 64, 64, 8), linear(2048, 640, 2560, 64, 128, 256)]
