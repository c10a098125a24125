#!/bin/bash
# A/B on one box: tools/ab.sh "ENV_A=.." "ENV_B=.." [rounds]  -- alternates bench.py runs, prints ms/step of each
A="$1"; B="$2"; R=${3:-3}
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  for v in "$A" "$B"; do
    ( export $v; exec python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/ab.json 2> gpurun_out/ab.err )
    python - <<PY
import json
try:
    print("round $i  [$v]  %.4f ms/step" % json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])["ms_per_step"])
except Exception as e:
    print("round $i  [$v]  failed:", e, open("gpurun_out/ab.err").read()[-300:])
PY
  done
done
