#!/bin/bash
# Re-tune the 3x3 / stride-1 rows of BASELINE config 5's per-GPU shape (4 images, 96 x 96 latents; fp16 and the block-scaled e4m3 policy) so that the
# patch form of the ping-pong kernel (variant 6) competes, then A/B the step with the shipped table against the re-tuned one on the same box.
# GPU box only; results under gpurun_out/.   usage: tools/retune_pp3.sh [fp16|fp8]
set -e
DT=${1:-fp16}
T=tinyfusers_amd/gemm_tune_gfx950.txt
if [ $DT = fp16 ]; then awk '!($6==3 && $7==1 && $8==0 && ($1==73728||$1==18432||$1==4608) && $10<64)' $T > gpurun_out/tune_partial_$DT.txt
else awk '!($6==3 && $7==1 && $8==0 && ($1==73728||$1==18432||$1==4608) && $10>=512)' $T > gpurun_out/tune_partial_$DT.txt; fi
rm -f gpurun_out/tune_c5_$DT.txt
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-config5 --images 4 --latent 96 --dtype $DT"
TF_GEMM_TUNE_TABLE=$PWD/gpurun_out/tune_partial_$DT.txt $B --tune-cache gpurun_out/tune_c5_$DT.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tuning run', d['ms_per_step'], 'ms/step')"
for i in 1 2; do
  for t in $T gpurun_out/tune_c5_$DT.txt; do
    TF_GEMM_TUNE_TABLE=$PWD/$t $B | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], 'ms/step')"
  done
done
