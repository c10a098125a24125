#!/usr/bin/env python3
"""Per-shape majority vote over several tuner tables (tools/tune_best.sh leaves gpurun_out/tune_<i>.txt): many shapes have two
configurations within the timing noise, and a single tuning run picks between them at random.
    python tools/tune_consensus.py out.txt table1.txt table2.txt ...   (ties: the earliest table wins)"""
import sys
from collections import Counter, OrderedDict

out, paths = sys.argv[1], sys.argv[2:]
votes = OrderedDict()
for p in paths:
    for line in open(p):
        f = line.split()
        if len(f) == 15:
            votes.setdefault(tuple(f[:10]), []).append(tuple(f[10:]))
with open(out, "w") as fo:
    for k in sorted(votes, key=lambda k: tuple(int(x) for x in k)):
        c = Counter(votes[k])
        best = max(c.values())
        pick = next(v for v in votes[k] if c[v] == best)
        fo.write(" ".join(k) + " " + " ".join(pick) + "\n")
print(f"{len(votes)} shapes from {len(paths)} tables -> {out}")
