#!/bin/bash
# N more fresh tuning runs of the step (the shipped table is ignored); tables land in gpurun_out/tune_m<i>.txt for tools/tune_consensus.py
export TF_GEMM_TUNE_TABLE=
for i in $(seq 1 ${1:-6}); do
  rm -f gpurun_out/tune_m$i.txt
  v=$(python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-e2e --no-config5 --tune-cache gpurun_out/tune_m$i.txt 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "tuning run $i: $v ms/step"
done
