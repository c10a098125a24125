#!/bin/bash
# Fresh tuning of BASELINE config 5's per-GPU shape (4 images, 96 x 96 latents): N runs that ignore the shipped table for the shapes of this step, the tuner
# cache of the fastest run kept, then an A/B against the shipped table on the same box.  GPU box only.   usage: tools/tune_c5.sh [fp16|fp8] [runs]
DT=${1:-fp16}; N=${2:-3}
T=tinyfusers_amd/gemm_tune_gfx950.txt
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-config5 --images 4 --latent 96 --dtype $DT"
best=999
for i in $(seq 1 $N); do
  rm -f gpurun_out/tune_c5f_${DT}_$i.txt
  v=$(TF_GEMM_TUNE_TABLE= $B --tune-cache gpurun_out/tune_c5f_${DT}_$i.txt 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "fresh tuning run $i: $v ms/step"
  if python -c "import sys; sys.exit(0 if float('$v') < float('$best') else 1)"; then best=$v; cp gpurun_out/tune_c5f_${DT}_$i.txt gpurun_out/tune_c5f_${DT}_best.txt; fi
done
for i in 1 2; do
  for t in $T gpurun_out/tune_c5f_${DT}_best.txt; do
    v=$(TF_GEMM_TUNE_TABLE=$PWD/$t $B 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$t: $v ms/step"
  done
done
