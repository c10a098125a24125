#!/usr/bin/env python3
"""Where k_gemm_ar's time goes: cycle sums (s_memtime) per block of consumer wave 0 (whole run | until barrier P | at barriers | epilogues) and loader wave 4
(whole run | counted waits | at barriers | stage issue), diagnostic build
    python -m tinyfusers_amd.build --tag stamp3 -DTF_IGEMM_STAMP=3
    TF_LIB_PATH=tinyfusers_amd/lib/libtinyfusers_hip_stamp3.so TF_LIB_ALLOW_MISSING=1 python tools/ar_stamp.py
GPU box only; the shipped library holds no stamp code."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pp_bench import T, hip, lib, st, time_call

assert "stamp3" in os.environ.get("TF_LIB_PATH", "")
rng = np.random.default_rng(0)


def case(m, n, k, act, label):
    x = T.DeviceArray.from_numpy((rng.standard_normal((m, k)) * 0.5).astype(np.float16))
    w = T.DeviceArray.from_numpy((rng.standard_normal((n, k)) * k ** -0.5).astype(np.float16))
    b = T.DeviceArray.from_numpy(rng.standard_normal(n).astype(np.float16))
    no = n // 2 if act else n
    y = T.DeviceArray.empty((m, no))

    def fn():
        lib.tf_gemm_force_config(128, 128, 1); lib.tf_gemm_debug(32768)
        hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, b.ptr, None, m, no, k, act, None, 0, st.handle)
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    us = time_call(fn)
    fn(); hip.tf_stream_sync(st.handle)
    tiles = -(-m // 128) * -(-n // 128)
    blocks = min(256, tiles)
    tpb = -(-tiles // blocks); blocks = -(-tiles // tpb)
    raw = np.zeros((4096 + blocks, 8), np.uint64)
    assert lib.tf_debug_stamps(raw.ctypes.data_as(ctypes.c_void_p), 4096 + blocks) == 0
    r = raw[:blocks].astype(np.float64)
    e = np.median(raw[4096:4096 + blocks].astype(np.float64), axis=0) / tpb
    el = np.median(raw[4096:4096 + blocks].astype(np.float64), axis=0)
    steps = tpb * (k // 64)
    med = np.median(r, axis=0)
    print(f"{label}: {m} x {n} x {k}, {us:.1f} us; {blocks} blocks x {tpb} tiles ({steps} K steps); medians over blocks, cycles (per K step | per tile)")
    print(f"  consumer wave 0: whole run {med[0]:9.0f} ({med[0]/steps:6.0f} | {med[0]/tpb:6.0f})  until barrier P {med[1]:7.0f}  at barriers {med[2]:9.0f} ({med[2]/steps:6.0f})  tile ends {med[3]:9.0f} (     - | {med[3]/tpb:6.0f})"
          f"  -> reads + MFMA + loop {(med[0]-med[1]-med[2]-med[3])/steps:6.0f} per step")
    print(f"  consumer tail (the run's last tile stored by the consumers) {el[7]:7.0f};  loader store duty (patch read + stores) {el[6]:9.0f} ({el[6]/steps:6.0f} per step)")
    print(f"  loader   wave 4: whole run {med[4]:9.0f} ({med[4]/steps:6.0f} | {med[4]/tpb:6.0f})  counted waits  {med[5]:9.0f} ({med[5]/steps:6.0f})  at barriers {med[6]:9.0f} ({med[6]/steps:6.0f})  stage issue {med[7]:9.0f} ({med[7]/steps:6.0f})")


if __name__ == "__main__":
    case(73728, 2560, 320, 1, "c5 geglu 320")
    case(73728, 960, 320, 0, "c5 qkv 320")
    case(73728, 320, 320, 0, "c5 out 320")
    case(8192, 2560, 320, 1, "c2 geglu 320")
