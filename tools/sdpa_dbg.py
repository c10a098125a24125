#!/usr/bin/env python3
"""Ablation of k_sdpa_dma<40, 2> (the d = 40 self-attention; GPU box): where does a 64-key tile's time go?  Runs tools/sdpa_bench.py's d = 40
shapes once per TF_SDPA_DBG setting (a separate process each: the flag is read when the library loads; every setting is its own
compile-time instance of the kernel, a run-time switch inside the loop costs more than anything it switches off; results are wrong by design).
bits: 1 no exp, 2 no P.V MFMAs, 4 no Q.K MFMAs, 8 no barrier, 16 no DMA in the loop, 32 no V reads, 64 no max."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import _ablation
ABL = _ablation.use()   # TF_SDPA_DBG is read by the -DTF_ABLATION library only; the shipped one ignores it
for flags, name in ((0, "full"), (1, "no exp"), (64, "no max"), (65, "no exp, no max"), (2, "no P.V MFMA"), (4, "no Q.K MFMA"), (6, "no MFMA at all"), (32, "no V reads"), (34, "no V reads, no P.V MFMA"),
                    (8, "no barrier"), (16, "no DMA in the loop"), (24, "no barrier, no DMA"), (6 | 32 | 24, "softmax VALU only"), (1 | 64 | 24, "MFMA + LDS reads only"),
                    (128, "same code, ablation build")):
    env = dict(os.environ, TF_SDPA_DBG=str(flags), TF_SDPA_NW="4", TF_LIB_PATH=ABL)   # the ablation instances are 4-wave blocks (the form the 8-wave one was derived from)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sdpa_bench.py"), "d40"], capture_output=True, text=True, env=env, timeout=300)
    rows = [l for l in r.stdout.splitlines() if "self" in l and "d= 40" in l]
    print(f"{name:28s} " + " | ".join(l.split("d= 40")[0].split()[0] + " " + l.split("d= 40")[0].split()[1] + " " + l.split("d= 40")[1].strip() for l in rows), flush=True)
    if r.returncode:
        print(r.stderr[-500:])
