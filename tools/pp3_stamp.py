#!/usr/bin/env python3
"""Where a K tile of k_igemm_pp3 spends its cycles: s_memtime stamps of wave 0 (first half) and wave 4 (second half) of block 0 around the phases of
the nine tiles of channel slab 1, from the diagnostic build (python -m tinyfusers_amd.build --tag stamp -DTF_PP3_STAMP=1; its fences forbid overlaps
the shipped kernel has: read the SHARES).  GPU box only.
The build --tag clock -DTF_PP3_STAMP=2 holds only the (s_memtime, s_memrealtime) pair around block 0's K loop: the in-kernel clock of the loop as shipped.
usage: TF_LIB_PATH=tinyfusers_amd/lib/libtinyfusers_hip_{stamp,clock}.so tools/pp3_stamp.py [cin cout hw bn]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pp_bench import T, hip, lib, st

cin, cout, hw, bn = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (640, 320, 96, 160)
n = 8
rng = np.random.default_rng(0)
x = T.DeviceArray.from_numpy((rng.standard_normal((n, cin, hw, hw)) * 0.5).astype(np.float16))
wt = T.DeviceArray.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * (cin * 9) ** -0.5).astype(np.float16))
y = T.DeviceArray.empty((n, cout, hw, hw))
ws = T.DeviceArray.from_numpy(np.zeros(1 << 20, np.uint8), np.uint8, "row")
lib.tf_gemm_force_config(192, bn, 1); lib.tf_gemm_debug(2048 | 32)
import time
t0 = time.time()
while time.time() - t0 < 2.5:      # the clock settles under sustained load
  for _ in range(50):
    hip.tf_conv2d_f16(y.ptr, x.ptr, None, wt.ptr, None, None, 0, None, n, hw, hw, cin, 0, cout, 3, 3, 1, 1, 0, ws.ptr, ws.nbytes, st.handle)
  hip.tf_stream_sync(st.handle)
raw = np.empty(128, np.uint64)
hip.tf_memcpy(raw.ctypes.data, ws.ptr, raw.nbytes, 2)      # device -> host
s = raw.reshape(2, 64)[:, :54].reshape(2, 9, 6).astype(np.int64)
c = raw[56:60].astype(np.int64)
print(f"conv3x3 {cin}->{cout} @{hw}, tile 192x{bn}: K loop of block 0 = {c[2] - c[0]} core cycles in {(c[3] - c[1]) * 10} ns: in-kernel clock {(c[2] - c[0]) / ((c[3] - c[1]) * 10.0):.2f} GHz "
      f"({'clock-only build: the shipped loop' if 'clock' in os.environ.get('TF_LIB_PATH', '') else 'stamped build'}), {(c[2] - c[0]) / (cin // 64 * 9):.0f} cycles per K tile")
if "clock" in os.environ.get("TF_LIB_PATH", ""):
    sys.exit(0)
names = ["reads + DMA issue", "vmcnt / LDS wait", "barrier 1", "MFMA", "vmcnt wait", "barrier 2"]
for w in range(2):
    d = np.diff(np.concatenate([s[w], np.roll(s[w][:, :1], -1, 0)], 1), axis=1)[:8]      # (tile, segment); the last tile's barrier-2 segment needs the next slab's stamp
    print(f"wave {4 * w} ({'first' if w == 0 else 'second'} half): cycles per segment, tiles of slab 1 (stamp cost ~40 each, not subtracted)")
    for t in range(8):
        print(f"   tap {t}: " + "  ".join(f"{names[k]} {d[t, k]:5d}" for k in range(6)) + f"   | tile {d[t].sum():5d}")
    print("   mean : " + "  ".join(f"{names[k]} {d[:, k].mean():5.0f}" for k in range(6)) + f"   | tile {d.sum(1).mean():5.0f}")
