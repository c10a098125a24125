#!/usr/bin/env python3
"""MFMA / issue / LDS counters per kernel family from one rocprofv3 --pmc pass (SQ block, <= 8 counters).
usage: pmc_sq_summary.py <dir> [out.json]"""
import csv, glob, json, sys
d = sys.argv[1]
f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
fam_of = lambda k: ("k_igemm" if ("k_igemm" in k or "k_gemm_" in k) else "k_sdpa" if "k_sdpa" in k else "k_gn_apply" if "gn_apply" in k else "k_gn_stats" if "gn_stats" in k
                    else "k_splitk_reduce" if "splitk" in k else "other")
# kernel durations of the same run (q_kernel_trace.csv) -> SIMD-cycles available per family at the nominal 2.4 GHz, 1024 SIMDs
CLK, SIMDS = 2.4e9, 256 * 4
dur = {}
kt = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
for r in csv.DictReader(open(kt)):
    fam = fam_of(r["Kernel_Name"])
    dur[fam] = dur.get(fam, 0.0) + (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
tot, n = {}, {}
for r in csv.DictReader(open(f)):
    fam = fam_of(r["Kernel_Name"])
    c = r["Counter_Name"]
    tot.setdefault(fam, {}).setdefault(c, 0.0)
    tot[fam][c] += float(r["Counter_Value"])
    n.setdefault(fam, {}).setdefault(c, 0)
    n[fam][c] += 1
out = {}
for fam, cs in sorted(tot.items()):
    o = {"launches": max(n[fam].values())}
    for c, v in cs.items():
        o[c] = v
    wc = cs.get("SQ_WAVE_CYCLES") or 0
    bc = cs.get("SQ_BUSY_CYCLES") or 0
    o["kernel_seconds"] = dur.get(fam, 0.0)
    if cs.get("SQ_VALU_MFMA_BUSY_CYCLES") and dur.get(fam):
        # MFMA-pipe busy cycles (summed over the chip's SIMDs) / SIMD-cycles the family's launches lasted (2.4 GHz nominal)
        o["mfma_busy_frac"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur[fam] * CLK * SIMDS), 4)
    if cs.get("SQ_INSTS_VALU_MFMA_MOPS_F16"):
        o["mfma_flop_executed"] = cs["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512.0          # one MOP = 512 FLOP (checked against 2MNK)
        if dur.get(fam):
            o["mfma_tflops_executed"] = round(o["mfma_flop_executed"] / dur[fam] / 1e12, 1)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in cs:
                o[c.lower() + "_per_wave_cycle"] = round(cs[c] / wc, 4)
    out[fam] = o
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
