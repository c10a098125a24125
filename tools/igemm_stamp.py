#!/usr/bin/env python3
"""Where the 10-20 us of a small batch-1 GEMM launch go: phase stamps of EVERY block of a k_igemm launch on the constant
100 MHz clock (s_memrealtime: one time base for all CUs), from the diagnostic build

    python -m tinyfusers_amd.build --tag stamp16 -DTF_IGEMM_STAMP=1
    TF_LIB_PATH=tinyfusers_amd/lib/libtinyfusers_hip_stamp16.so TF_LIB_ALLOW_MISSING=1 python tools/igemm_stamp.py > profiles/r05_small_gemm_stamps.txt

Wave 0 (a consumer) and wave 4 (a loader) of every block stamp: entry | K tile 0 landed in LDS | K loop done (barrier X) | epilogue stores issued |
stores drained (s_waitcnt vmcnt(0)).  All times in microseconds relative to the FIRST block's entry; 10 ns resolution.  The three launches are the
step's own configurations (tile, split, variant from the shipped table): 8192 x 320 x 320 and 512 x 1280 x 1280 (20 launches per step each) and the
3 x 3 conv 128 x 1280 x 11520 of the 8 x 8 level (split-K 16).  Each is measured warm (back to back: operands in L2 / Infinity Cache) and cold (behind a
384 MiB memset: every layer's weights come from HBM in the real step).  GPU box only; the shipped library holds no stamp code."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pp_bench import T, hip, lib, st

assert "stamp16" in os.environ.get("TF_LIB_PATH", ""), "run with TF_LIB_PATH=.../libtinyfusers_hip_stamp16.so"
flush = T.DeviceArray.empty((384 << 20,), np.uint8, "row")
rng = np.random.default_rng(0)


def event_us(fn, cold, reps=9):
    ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
    hip.tf_event_create(ctypes.byref(ev0)); hip.tf_event_create(ctypes.byref(ev1))
    ts = []
    for r in range(reps):
        if cold:
            hip.tf_memset_async(flush.ptr, r, flush.nbytes, st.handle)
        hip.tf_event_record(ev0, st.handle)
        fn()
        hip.tf_event_record(ev1, st.handle)
        hip.tf_stream_sync(st.handle)
        ms = ctypes.c_float(); hip.tf_event_elapsed_ms(ctypes.byref(ms), ev0, ev1)
        ts.append(ms.value * 1e3)
    return float(np.median(ts))


def pct(a):
    return "min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())


def report(label, fn, blocks, flops, in_bytes, out_bytes):
    print(f"=== {label}: {blocks} blocks, {flops / 1e9:.2f} GFLOP, operands {in_bytes / 1e6:.2f} MB in / {out_bytes / 1e6:.2f} MB out "
          f"(MFMA floor {flops / 2.5e15 * 1e6:.2f} us, HBM floor {(in_bytes + out_bytes) / 8e12 * 1e6:.2f} us)")
    for cold in (False, True):
        for _ in range(3):
            fn()
        hip.tf_stream_sync(st.handle)
        ev = event_us(fn, cold)
        raws = []
        for r in range(7):                                   # seven stamped launches: report the one with the median span
            if cold:
                hip.tf_memset_async(flush.ptr, r, flush.nbytes, st.handle)
            fn()
            hip.tf_stream_sync(st.handle)
            raw = np.zeros((blocks, 8), np.uint64)
            assert lib.tf_debug_stamps(raw.ctypes.data_as(ctypes.c_void_p), blocks) == 0
            raws.append(raw.astype(np.int64))
        spans = [(r[:, [3, 7]].max() - r[:, [0, 4]].min()) * 0.01 for r in raws]
        s = raws[int(np.argsort(spans)[len(spans) // 2])].astype(np.float64) * 0.01
        t0 = s[:, [0, 4]].min()
        s -= t0
        c_entry, c_kdone, c_issued, c_drained, l_entry, l_tile0, l_last, l_left = (s[:, i] for i in range(8))
        span = max(c_drained.max(), l_left.max())
        print(f"--- {'COLD (behind a 384 MiB memset)' if cold else 'WARM (back to back)'}: event bracket {ev:.2f} us (incl. ~2-3 us of bracket overhead); first entry -> last store drained {span:.2f} us")
        print(f"  block entry (dispatch skew)            {pct(c_entry)}")
        print(f"  entry -> K tile 0 landed (loader)      {pct(l_tile0 - l_entry)}")
        print(f"  tile 0 landed -> K loop done           {pct(c_kdone - l_tile0)}")
        print(f"  K loop done -> epilogue stores issued  {pct(c_issued - c_kdone)}")
        print(f"  stores issued -> drained               {pct(c_drained - c_issued)}")
        print(f"  block lifetime (entry -> drained)      {pct(np.maximum(c_drained, l_left) - c_entry)}")
        print(f"  time of the block that ends last: entry {c_entry[np.argmax(c_drained)]:.2f}, tile0 {l_tile0[np.argmax(c_drained)]:.2f}, K loop done {c_kdone[np.argmax(c_drained)]:.2f}, drained {c_drained.max():.2f}")


def linear_case(M, N, K, bm, bn, variant_flag, order):
    x = T.DeviceArray.from_numpy((rng.standard_normal((M, K)) * 0.5).astype(np.float16), np.float16, "row")
    w = T.DeviceArray.from_numpy((rng.standard_normal((N, K)) * K ** -0.5).astype(np.float16), np.float16, "row")
    b = T.DeviceArray.from_numpy(rng.standard_normal(N).astype(np.float16), np.float16, "row")
    res = T.DeviceArray.from_numpy(rng.standard_normal((M, N)).astype(np.float16), np.float16, "row")
    y = T.DeviceArray.empty((M, N), np.float16, "row")

    def fn():
        lib.tf_gemm_force_config(bm, bn, 1); lib.tf_gemm_debug(variant_flag | (64 if order else 32))
        hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, b.ptr, res.ptr, M, N, K, 0, None, 0, st.handle)
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    blocks = -(-M // bm) * -(-N // bn)
    report(f"linear {M} x {N} x {K} (+ bias + residual), tile {bm}x{bn}, variant flag {variant_flag}", fn, blocks, 2.0 * M * N * K, (M * K + N * K + M * N) * 2, M * N * 2)
    return x, w, b, res, y


def conv_case(n, hw, cin, cout, bm, bn, split, variant_flag, order):
    x = T.DeviceArray.from_numpy((rng.standard_normal((n, cin, hw, hw)) * 0.5).astype(np.float16))
    wt = T.DeviceArray.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * (cin * 9) ** -0.5).astype(np.float16))
    b = T.DeviceArray.from_numpy(rng.standard_normal(cout).astype(np.float16), np.float16, "row")
    y = T.DeviceArray.empty((n, cout, hw, hw))
    nb = hip.tf_conv2d_workspace(n, hw, hw, cin, 0, cout, 3, 3, 1, 1, 0)
    ws = T.DeviceArray.empty((max(nb, 16),), np.uint8, "row")
    M, K = n * hw * hw, 9 * cin

    def fn():
        lib.tf_gemm_force_config(bm, bn, split); lib.tf_gemm_debug(variant_flag | (64 if order else 32))
        hip.tf_conv2d_f16(y.ptr, x.ptr, None, wt.ptr, b.ptr, None, 0, None, n, hw, hw, cin, 0, cout, 3, 3, 1, 1, 0, ws.ptr, nb, st.handle)
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    blocks = -(-M // bm) * -(-cout // bn) * split
    report(f"conv 3x3 {cin}->{cout} @{hw}x{hw} batch {n} = {M} x {cout} x {K}, tile {bm}x{bn}, split-K {split} (stamps: the GEMM kernel; its reduce launch follows), variant flag {variant_flag}",
           fn, blocks, 2.0 * M * cout * K, (M * cin + cout * K) * 2, M * cout * 2 * split)
    return x, wt, b, y, ws


if __name__ == "__main__":
    print(f"# tools/igemm_stamp.py, library {os.environ.get('TF_LIB_PATH')}; times in us relative to the first block's entry (s_memrealtime, 10 ns ticks)")
    keep = []
    keep.append(linear_case(8192, 320, 320, 64, 160, 256, 0))        # the step: 64x160, variant 3 (ALL8)
    keep.append(linear_case(512, 1280, 1280, 64, 64, 256, 0))        # the step: 64x64, variant 3
    keep.append(conv_case(2, 8, 1280, 1280, 64, 160, 16, 256, 0))    # the step: 64x160, split 16, variant 3
    keep.append(linear_case(2048, 640, 640, 64, 128, 256, 0))
    keep.append(conv_case(2, 64, 320, 320, 64, 160, 1, 256, 0))      # 8192 x 320 x 2880: the 64 x 64 level's 3x3 conv
    keep.append(conv_case(2, 16, 1280, 1280, 128, 160, 8, 128, 0))   # 512 x 1280 x 11520: variant 2 (patch), split 8
