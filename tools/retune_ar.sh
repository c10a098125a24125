#!/bin/bash
# Re-tune the rows k_gemm_ar (variant 8) can take -- linear / 1x1, K = 256 / 320, no residual -- of BASELINE config 2 (default bench) and config 5's per-GPU shape
# (4 images, 96 x 96 latents), then A/B the step with the shipped table against the re-tuned one on the same box.  GPU box only; results under gpurun_out/.
set -e
T=tinyfusers_amd/gemm_tune_gfx950.txt
awk '!(($3==320||$3==256) && $6==1 && $7==1 && $8==0 && int($10/2)%2==0 && int($10/4)%2==0 && int($10/16)%2==0 && $10<64)' $T > gpurun_out/tune_partial_ar.txt
echo "rows dropped: $(( $(wc -l < $T) - $(wc -l < gpurun_out/tune_partial_ar.txt) ))"
rm -f gpurun_out/tune_ar_c2.txt gpurun_out/tune_ar_c5.txt
B2="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-e2e --no-config5"
B5="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-config5 --images 4 --latent 96"
TF_GEMM_TUNE_TABLE=$PWD/gpurun_out/tune_partial_ar.txt $B2 --tune-cache gpurun_out/tune_ar_c2.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tuning run c2', d['ms_per_step'], 'ms/step')"
TF_GEMM_TUNE_TABLE=$PWD/gpurun_out/tune_ar_c2.txt $B5 --tune-cache gpurun_out/tune_ar_c5.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tuning run c5', d['ms_per_step'], 'ms/step')"
diff <(sort $T) <(sort gpurun_out/tune_ar_c5.txt) | grep '^[<>]' || true
for i in 1 2 3; do
  for t in $T gpurun_out/tune_ar_c5.txt; do
    TF_GEMM_TUNE_TABLE=$PWD/$t $B2 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 $t', d['ms_per_step'], 'ms/step')"
    TF_GEMM_TUNE_TABLE=$PWD/$t $B5 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 $t', d['ms_per_step'], 'ms/step')"
  done
done
