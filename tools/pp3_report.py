#!/usr/bin/env python3
"""Assembles profiles/r04_pp3.txt: the final same-box tables of tools/pp3_bench.py and tools/mx_bench.py (gpurun_out/pp3_bench_final.log,
gpurun_out/mx_bench_final.log, produced on the GPU box) plus the development A/Bs kept under profiles/r04_pp3_logs/.   usage: tools/pp3_report.py"""
import os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from bench import csrc_hash
L = os.path.join(R, "profiles", "r04_pp3_logs")
G = os.path.join(R, "gpurun_out")


def best(path):
    cur, shape, res = None, None, {}
    for l in open(path):
        if l.startswith("== lib"):
            cur = l.strip()[3:]
            while cur in res:
                cur += "'"
            res[cur] = {}
        elif l.startswith("conv3x3"):
            shape = l.split("(")[0].strip()
        else:
            m = re.match(r"\s+PP3192x(\d+)/1o(\d):\s+([\d.]+) us\s+(\d+) TF", l)
            if m and cur:
                res[cur].setdefault(shape, []).append(int(m.group(4)))
    return res


def table(path, out):
    r = best(path)
    shapes = list(next(v for v in r.values() if v).keys())
    out.append("%-14s " % "library" + " ".join("%22s" % s.replace("conv3x3 ", "") for s in shapes))
    for k, v in r.items():
        if v:
            out.append("%-14s " % k + " ".join("%22d" % max(v[s]) for s in shapes if s in v))


out = [f"# k_igemm_pp3 -- the patch form of the ping-pong kernel (round 4).  MI355X, one box per block unless said otherwise; csrc {csrc_hash()} for the tables of sections 1 and 2.",
       "# Times: tools/pp_bench.py's graph-replayed launches (6 per graph x 3 replays), us per launch; TF = 2 M N K / time.\n",
       "## 1. Final sources, same box: tools/pp3_bench.py -- BASELINE config 5's 3x3 convolutions (UNet batch 8), variant 6 (PP3) against k_igemm_pp (pp), the deep ring (deep) and",
       "##    k_igemm_patch (patch); <bm>x<bn>/<split-K>o<tile order>.  PP3 takes its own tile width (160 on 96-pixel rows, 128 on 48 / 24).",
       open(os.path.join(G, "pp3_bench_final.log")).read(),
       "## 2. Final sources, same box: tools/mx_bench.py -- block-scaled e4m3 (k_igemm_pp<F8> tiles against PP3 = k_igemm_pp3<F8>; sorted by time)",
       open(os.path.join(G, "mx_bench_final.log")).read(),
       "## 3. In-kernel clock and phase stamps (tools/pp3_stamp.py; diagnostic builds --tag clock -DTF_PP3_STAMP=2 / --tag stamp -DTF_PP3_STAMP=1; fp16, before the e4m3 form was added)",
       open(os.path.join(L, "stamps.log")).read(),
       "## 4. Development A/Bs (each block one box; TF of the better tile order)",
       "# 4a. first version (run-time row length, the aligned tiles' swizzle on the patch) against k_igemm_pp, same box"]
for l in open(os.path.join(L, "v1_vs_pp.log")):
    if l.startswith("conv") or "pp192x160/1" in l or "pp192x128/1" in l or "PP3" in l:
        out.append(l.rstrip())
out += ["\n# 4b. same box: lib = row length as template parameter (immediate patch offsets) + shift-invariant patch swizzle + up to 4 ring slots; _pp3ns3 = the same with 3 slots",
        "#     everywhere (shipped); _pp3v1 = first version with the shift-invariant swizzle; _pp3v1s0 = first version with the aligned tiles' swizzle"]
table(os.path.join(L, "swizzle_immediates_ring.log"), out)
out.append("\n# 4c. same box: part of a tile's loads issued between its MFMAs instead of in front of the barrier: w<weight pieces moved>p<patch piece moved>")
table(os.path.join(L, "loads_between_mfmas.log"), out)
out += ["\n## 5. The step on BASELINE config 5's per-GPU shape (4 images, 96 x 96 latents), ms per step, same box per block (tools/retune_pp3.sh: the table without variant-6 rows",
        "##    against the table re-tuned with the patch kernel as a candidate)",
        "# fp16, 3x3 / stride 1 rows (before up-sampling and skip-projection support):",
        "".join(l for l in open(os.path.join(L, "retune_fp16.log")) if "ms/step" in l),
        "# fp8 policy, block-scaled 3x3 rows of the 48 / 24-pixel levels:",
        "".join(l for l in open(os.path.join(L, "retune_fp8.log")) if "ms/step" in l)]
open(os.path.join(R, "profiles", "r04_pp3.txt"), "w").write("\n".join(out) + "\n")
print("wrote profiles/r04_pp3.txt")
