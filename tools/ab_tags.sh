#!/bin/bash
# tools/ab_tags.sh ROUNDS "TAG1 TAG2 ..." [bench args...]: same-box A/B of the shipped library against tagged builds of the SAME sources (python -m tinyfusers_amd.build --tag T -D...),
# alternating, one bench process per measurement, identical settings and table.  GPU box only.
R=$1; TAGS=$2; shift 2
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  for t in shipped $TAGS; do
    if [ $t = shipped ]; then lib=$PWD/tinyfusers_amd/lib/libtinyfusers_hip.so; else lib=$PWD/tinyfusers_amd/lib/libtinyfusers_hip_$t.so; fi
    ( export TF_LIB_PATH=$lib TF_LIB_ALLOW_MISSING=1; exec python bench.py --steps ${AB_STEPS:-200} --warmup 10 --no-cpu-baseline --no-roofline --no-e2e --no-config5 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err )
    python - "$t[$i]" <<'PY'
import json, sys
try: print("%-14s %.4f ms/step" % (sys.argv[1], json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])["ms_per_step"]), flush=True)
except Exception as e: print(sys.argv[1], "failed:", e, open("gpurun_out/ab.err").read()[-300:], flush=True)
PY
  done
done
