#!/bin/bash
# tools/ab3.sh rounds VAR=val1 VAR=val2 ... : alternates bench runs over the listed settings on one box
R=$1; shift
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  for v in "$@"; do
    ( export $v; exec python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-roofline --no-e2e --no-config5 > gpurun_out/ab.json 2> gpurun_out/ab.err )
    python - <<PY
import json
try:
    print("round $i  [$v]  %.4f ms/step" % json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])["ms_per_step"])
except Exception as e:
    print("round $i  [$v]  failed:", e, open("gpurun_out/ab.err").read()[-300:])
PY
  done
done
