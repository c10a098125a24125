#!/usr/bin/env python3
"""Ablation (tf_gemm_debug) of the short-K, many-tile GEMMs: WIDE vs deep, per tile size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import _ablation; _ablation.use()   # the ablation bits exist only in the -DTF_ABLATION library
from tinyfusers_amd.native import lib
from tools.gemm_dbg import run
for wide in (16, 8):
    print("=== forced", "WIDE" if wide == 16 else "deep")
    import tools.gemm_dbg as G
    orig = lib.tf_gemm_debug
    # gemm_dbg.run sets debug flags 0..7; OR the ring-variant bit into every call
    G.lib.tf_gemm_debug = lambda f, _o=orig, _w=wide: _o(f | _w)
    run(2, 64, 64, 320, 2560, 1, 128, 128, 1, "geglu-ish 320->2560@64")
    run(2, 64, 64, 320, 2560, 1, 64, 128, 1, "geglu-ish 320->2560@64")
    run(2, 64, 64, 320, 960, 1, 128, 128, 1, "qkv 320->960@64")
    run(2, 64, 64, 320, 320, 1, 64, 64, 1, "lin 320@64")
    run(2, 32, 32, 640, 5120, 1, 128, 128, 1, "geglu-ish 640->5120@32")
    G.lib.tf_gemm_debug = orig
