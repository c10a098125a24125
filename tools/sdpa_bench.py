#!/usr/bin/env python3
"""Graph-timed SDPA shapes of the SD-1.5 step (self + cross attention at every level) through tf_sdpa_f16."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip
from tools.gemm_bench import time_call, st

SHAPES = [("self 64^2 d40", 2, 8, 4096, 4096, 40), ("self 32^2 d80", 2, 8, 1024, 1024, 80), ("self 16^2 d160", 2, 8, 256, 256, 160),
          ("self 8^2 d160", 2, 8, 64, 64, 160), ("cross 64^2 d40", 2, 8, 4096, 77, 40), ("cross 32^2 d80", 2, 8, 1024, 77, 80),
          ("cross 16^2 d160", 2, 8, 256, 77, 160), ("vae 64^2 d512", 1, 1, 4096, 4096, 512),
          ("c5 self 96^2 d40", 8, 8, 9216, 9216, 40), ("c5 cross 96^2 d40", 8, 8, 9216, 77, 40), ("c5 self 48^2 d80", 8, 8, 2304, 2304, 80)]


def main():
    rng = np.random.default_rng(0)
    tot = 0.0
    for name, b, nh, tq, tk, hs in SHAPES:
        if hs > 160 or (len(sys.argv) > 1 and sys.argv[1] == "d40" and hs != 40):
            continue
        c = nh * hs
        q = T.DeviceArray.from_numpy(rng.standard_normal((b, tq, c)).astype(np.float16), layout="row")
        k = T.DeviceArray.from_numpy(rng.standard_normal((b, tk, c)).astype(np.float16), layout="row")
        v = T.DeviceArray.from_numpy(rng.standard_normal((b, tk, c)).astype(np.float16), layout="row")
        o = T.DeviceArray.empty((b, tq, c), np.float16, "row")

        def fn():
            hip.tf_sdpa_f16(o.ptr, q.ptr, k.ptr, v.ptr, b, nh, tq, tk, hs, tq * c, hs, c, tk * c, hs, c, tk * c, hs, c, tq * c, hs, c, 0, st.handle)
        us = time_call(fn)
        fl = 4.0 * b * nh * tq * tk * hs
        print(f"{name:18s} B={b} NH={nh} Tq={tq:5d} Tk={tk:5d} d={hs:3d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
