#!/usr/bin/env python3
"""Back-to-back timing of single GEMM/conv shapes through the C-ABI, per tile configuration.
usage: gemm_bench.py [quick|all]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib

T.ensure_init(0)
st = T.Stream()
CFGS = [(128, 160), (64, 160), (128, 128), (64, 128), (128, 64), (64, 64)]


def time_call(fn, reps=20):
    """Average device time per call: `reps` calls captured into a HIP graph (no host launch bound), replayed 5x."""
    ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
    hip.tf_event_create(ctypes.byref(ev0)); hip.tf_event_create(ctypes.byref(ev1))
    for _ in range(2): fn()
    hip.tf_stream_sync(st.handle)
    hip.tf_graph_begin_capture(st.handle)
    for _ in range(reps): fn()
    g = ctypes.c_void_p()
    hip.tf_graph_end_capture(st.handle, ctypes.byref(g))
    hip.tf_graph_launch(g, st.handle)
    hip.tf_event_record(ev0, st.handle)
    for _ in range(5): hip.tf_graph_launch(g, st.handle)
    hip.tf_event_record(ev1, st.handle)
    hip.tf_stream_sync(st.handle)
    ms = ctypes.c_float(); hip.tf_event_elapsed_ms(ctypes.byref(ms), ev0, ev1)
    hip.tf_graph_destroy(g)
    return ms.value * 1e3 / (reps * 5)


def bench_conv(n, h, w, cin, cout, k, stride=1, splits=(1,), cfgs=CFGS, label=""):
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = T.DeviceArray.from_numpy(np.random.randn(n, cin, h, w).astype(np.float16) * 0.5)
    wt = T.DeviceArray.from_numpy((np.random.randn(cout, cin, k, k) * (cin * k * k) ** -0.5).astype(np.float16))
    b = T.DeviceArray.from_numpy(np.random.randn(cout).astype(np.float16))
    y = T.DeviceArray.empty((n, cout, ho, wo))
    M, K = n * ho * wo, k * k * cin
    ws = T.DeviceArray.empty((32 * M * cout * 4 + 16,), np.uint8, "row")
    flops = 2.0 * M * cout * K
    res = []
    for bm, bn in cfgs:
        for sk in splits:
            lib.tf_gemm_force_config(bm, bn, sk)
            def fn():
                hip.tf_conv2d_f16(y.ptr, x.ptr, None, wt.ptr, b.ptr, None, 0, None, n, h, w, cin, 0, cout, k, k, stride, pad, 0, ws.ptr, ws.nbytes, st.handle)
            us = time_call(fn)
            res.append((us, bm, bn, sk))
    lib.tf_gemm_force_config(0, 0, 0)
    def fn():
        hip.tf_conv2d_f16(y.ptr, x.ptr, None, wt.ptr, b.ptr, None, 0, None, n, h, w, cin, 0, cout, k, k, stride, pad, 0, ws.ptr, ws.nbytes, st.handle)
    auto = time_call(fn)
    res.sort()
    best = res[0]
    print(f"{label:28s} M={M:5d} N={cout:5d} K={K:5d}  heuristic {auto:7.1f} us {flops/auto/1e6:6.0f} TF | best {best[0]:7.1f} us {flops/best[0]/1e6:6.0f} TF  {best[1]}x{best[2]} sk{best[3]} | " +
          " ".join(f"{bm}x{bn}/{sk}:{us:.0f}" for us, bm, bn, sk in res[1:6]), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    S = (1, 2, 4, 8, 16)
    bench_conv(2, 64, 64, 320, 320, 3, splits=(1, 2), label="conv3x3 320@64")
    bench_conv(2, 64, 64, 320, 320, 1, splits=(1,), label="conv1x1 320@64")
    bench_conv(2, 32, 32, 640, 640, 3, splits=(1, 2, 4), label="conv3x3 640@32")
    bench_conv(2, 32, 32, 640, 640, 1, splits=(1, 2), label="conv1x1 640@32")
    bench_conv(2, 16, 16, 1280, 1280, 3, splits=(1, 2, 4, 8), label="conv3x3 1280@16")
    bench_conv(2, 16, 16, 1280, 1280, 1, splits=(1, 2, 4), label="conv1x1 1280@16")
    bench_conv(2, 8, 8, 1280, 1280, 3, splits=(4, 8, 16, 32), label="conv3x3 1280@8")
    if mode == "all":
        bench_conv(2, 64, 64, 640, 320, 3, splits=(1, 2), label="conv3x3 640->320@64")
        bench_conv(2, 64, 64, 640, 640, 3, splits=(1,), label="conv3x3 640@64")
        bench_conv(2, 32, 32, 1280, 640, 3, splits=(1, 2, 4), label="conv3x3 1280->640@32")
        bench_conv(2, 16, 16, 2560, 1280, 3, splits=(1, 2, 4, 8), label="conv3x3 2560->1280@16")
        bench_conv(2, 8, 8, 2560, 1280, 3, splits=(4, 8, 16, 32), label="conv3x3 2560->1280@8")
        bench_conv(2, 64, 64, 320, 2560, 1, splits=(1,), label="geglu-ish 320->2560@4096")
        bench_conv(2, 64, 64, 1280, 320, 1, splits=(1, 2), label="ff out 1280->320@4096")
        bench_conv(2, 8, 8, 1280, 1280, 1, splits=(1, 2, 4, 8), label="conv1x1 1280@8")
