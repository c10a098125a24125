#!/usr/bin/env python3
"""HBM traffic per launch of the k_igemm family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
Units/corrections per MI355X_MICROARCH.md (HBM section): counters are in KiB-granular units of 1024 B ... the
FETCH_SIZE of a wide coalesced stream reads exactly half of the real bytes on gfx950 -> doubled.
usage: pmc_summary.py <fetch_dir> <write_dir> [out.json]"""
import csv, glob, hashlib, json, os, sys

def load(d, counter):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    tot, n = {}, {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"]
        fam = "k_igemm" if ("k_igemm" in k or "k_gemm_" in k) else "k_sdpa" if "k_sdpa" in k else "k_gn" if "k_gn" in k else "k_layer_norm" if "layer_norm" in k else "k_splitk_reduce" if "splitk" in k else "other"
        tot[fam] = tot.get(fam, 0.0) + float(r["Counter_Value"]); n[fam] = n.get(fam, 0) + 1
    return tot, n

ft, fn = load(sys.argv[1], "FETCH_SIZE")
wt, wn = load(sys.argv[2], "WRITE_SIZE")
out = {}
for fam in sorted(ft):
    fetch_b = ft[fam] * 1024 * 2.0          # KiB units; gfx950 FETCH_SIZE = 1/2 of a wide coalesced stream
    write_b = wt.get(fam, 0.0) * 1024
    out[fam] = {"launches": fn[fam], "fetch_bytes_per_launch": fetch_b / fn[fam], "write_bytes_per_launch": write_b / max(1, wn.get(fam, 1)),
                "hbm_bytes_per_launch": fetch_b / fn[fam] + write_b / max(1, wn.get(fam, 1))}
# stamp: hash of the kernel sources this profile was taken on (bench.py reports `traffic` only while it still matches)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash          # ONE definition of the stamp (it covers csrc/*.hip, *.h and *.inc)
out["csrc_sha16"] = csrc_hash()
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
