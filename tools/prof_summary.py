#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv per step: prof_summary.py <dir> <steps_in_run> [family.json]
family.json: the k_igemm family's launches / ms per step / average launch, stamped with the hash of the kernel sources (bench.py quotes it next
to its live HIP-event figure while the hash still matches)."""
import csv, glob, json, os, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = (glob.glob(d + "/*/*kernel_stats.csv") + glob.glob(d + "/*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
tot = 0
out = []
for r in rows:
    n = r["Name"]
    if "rocclr" in n or "k_prof_spin" in n: continue      # (runtime helpers; the spin kernel is the instrumented pass's event-overhead calibration)
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    out.append("%-64s calls/step %6.1f  ms/step %7.3f  avg_us %8.2f" % (n[:64], int(r["Calls"]) / steps, ms, float(r["AverageNs"]) / 1e3))
print("\n".join(out))
print("sum of kernel durations: %.3f ms/step over %d kernel launches/step" % (tot, sum(int(r["Calls"]) for r in rows if "rocclr" not in r["Name"] and "k_prof_spin" not in r["Name"]) / steps))
if len(sys.argv) > 3:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import csrc_hash
    fam = [r for r in rows if "k_igemm" in r["Name"] or "k_gemm_" in r["Name"]]      # (k_gemm_c4 / k_gemm_c8 / k_gemm_ar: the family's persistent short-K kernels)
    red = [r for r in rows if "k_splitk_reduce" in r["Name"]]
    calls = sum(int(r["Calls"]) for r in fam)
    ns = sum(float(r["TotalDurationNs"]) for r in fam)
    rcalls = sum(int(r["Calls"]) for r in red)
    rns = sum(float(r["TotalDurationNs"]) for r in red)
    json.dump({"csrc_sha16": csrc_hash(), "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-config5",
               "k_igemm": {"launches_per_step": calls / steps, "ms_per_step": ns / 1e6 / steps, "avg_launch_us": ns / 1e3 / max(1, calls)},
               "k_splitk_reduce": {"launches_per_step": rcalls / steps, "ms_per_step": rns / 1e6 / steps, "avg_launch_us": rns / 1e3 / max(1, rcalls)},
               "k_igemm_plus_reduce": {"launches_per_step": (calls + rcalls) / steps, "ms_per_step": (ns + rns) / 1e6 / steps,
                                       "avg_launch_us": (ns + rns) / 1e3 / max(1, calls)}},
              open(sys.argv[3], "w"), indent=1)
