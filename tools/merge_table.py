#!/usr/bin/env python3
"""Merge tuner tables: rows of the later files replace rows of the earlier ones with the same ten key fields.   usage: tools/merge_table.py out.txt base.txt new1.txt [new2.txt ...]"""
import sys
out, paths = sys.argv[1], sys.argv[2:]
rows = {}
for p in paths:
    for line in open(p):
        f = line.split()
        if len(f) == 15:
            rows[tuple(int(x) for x in f[:10])] = f[10:]
with open(out, "w") as fo:
    for k in sorted(rows):
        fo.write(" ".join(str(x) for x in k) + " " + " ".join(rows[k]) + "\n")
print(f"{len(rows)} rows -> {out}")
