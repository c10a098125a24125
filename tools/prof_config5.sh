#!/bin/bash
# rocprofv3 kernel statistics of BASELINE config 5's per-GPU shape (4 images, 96 x 96 latents), fp16 and fp8 (the last block of
# tools/refresh_profiles.sh on its own); results land in gpurun_out/kernel_stats_images4_latent96_<dtype>.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dt in ${1:-fp16 fp8}; do
  rm -rf $O/prof_stats5
  rocprofv3 --kernel-trace --stats -d $O/prof_stats5 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-config5 --images 4 --latent 96 --dtype $dt > $O/prof_stats5.log 2>&1
  echo "# csrc $(python3 -c "import sys; sys.path.insert(0, '$R'); from bench import csrc_hash; print(csrc_hash())")  rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-config5 --images 4 --latent 96 --dtype $dt   (9 steps executed: 2 in compile() + 2 warm-up + 5 timed)" > $O/kernel_stats_images4_latent96_$dt.txt
  python3 $R/tools/prof_summary.py $O/prof_stats5 9 >> $O/kernel_stats_images4_latent96_$dt.txt
  rm -rf $O/prof_stats5
done
