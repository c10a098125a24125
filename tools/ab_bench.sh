#!/bin/bash
# tools/ab_bench.sh ROUNDS "LABEL=ENV ASSIGNMENTS"...: same-box A/B of the denoising step (bench.py, 200 graph-replayed steps per run) over
# environment settings, alternating; e.g.  tools/ab_bench.sh 3 "base=TF_X=0" "nt=TF_LIB_PATH=$PWD/tinyfusers_amd/lib/libtinyfusers_hip_wnt.so"
R=$1; shift
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  for spec in "$@"; do
    label=${spec%%=*}; envs=${spec#*=}
    ( export $envs; exec python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-roofline --no-e2e --no-config5 $AB_BENCH_ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err )
    python - "$label[$i]" <<'PY'
import json, sys
try:
    print("%-14s %.4f ms/step" % (sys.argv[1], json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])["ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "failed:", e, open("gpurun_out/ab.err").read()[-300:])
PY
  done
done
