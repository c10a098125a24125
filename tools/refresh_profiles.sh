#!/bin/bash
# Regenerates the evidence under profiles/ on the GPU box (run from the repo root; results land in gpurun_out/, copy them over):
#   bench line, rocprofv3 kernel trace + stats of the same command, per-shape GEMM table, PMC passes (traffic: FETCH_SIZE and
#   WRITE_SIZE in separate runs, stamped with the kernel-source hash by tools/pmc_summary.py; SQ counters with the kernel trace),
#   same-box A/B of the GroupNorm fusions, the config-5 shape in fp16 and fp8.  rocprofv3 gets the program itself after `--`.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq
rocprofv3 --kernel-trace --stats -d $O/prof_stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-config5 > $O/prof_stats.log 2>&1
python3 $R/tools/prof_summary.py $O/prof_stats 32 $O/kernel_family.json > $O/kernel_stats_per_step.txt      # steps executed: 2 in compile() + 5 warm-up + 20 timed + 5 instrumented
cp $(ls $O/prof_stats/*/*kernel_stats.csv $O/prof_stats/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
python3 $R/tools/step_profile.py $O/gemm_shapes.csv > $O/gemm_shapes.txt
rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-config5 > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/prof_write --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-config5 > $O/prof_write.log 2>&1
python3 $R/tools/pmc_summary.py $O/prof_fetch $O/prof_write $O/pmc_traffic.json
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
  -d $O/prof_sq --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-config5 > $O/prof_sq.log 2>&1
python3 $R/tools/pmc_sq_summary.py $O/prof_sq $O/pmc_sq.json
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq
[ -n "$QUICK" ] && { ls -la $O; exit 0; }      # QUICK=1: the bench line, kernel stats and PMC passes only
# same-box A/B of the switches that change the step (ms per step, 60 graph-replayed steps each, two rounds)
: > $O/ab_fusions.txt
for i in 1 2; do
  for cfg in "" "TF_FUSE_GROUP_NORM=0" "TF_FUSE_REDUCE_NORM=0" "TF_HOIST_STEP_INVARIANTS=0" "TF_SPLITK_PARTIALS=32"; do
    v=$(env $cfg python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-e2e --no-config5 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $i  ${cfg:-default}: $v ms/step" >> $O/ab_fusions.txt
  done
done
# BASELINE config 5's per-GPU shape (4 images, 96 x 96 latents), fp16 and fp8: bench line, per-shape GEMM table, rocprofv3 kernel statistics
for dt in fp16 fp8; do
  python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-config5 --images 4 --latent 96 --dtype $dt > $O/bench_images4_latent96_$dt.json 2>/dev/null
  python3 $R/tools/step_profile.py $O/gemm_shapes_images4_latent96_$dt.csv 4 96 $dt > $O/gemm_shapes_images4_latent96_$dt.txt 2>/dev/null
  rm -rf $O/prof_stats5
  rocprofv3 --kernel-trace --stats -d $O/prof_stats5 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-config5 --images 4 --latent 96 --dtype $dt > $O/prof_stats5.log 2>&1
  echo "# csrc $(python3 -c "import sys; sys.path.insert(0, '$R'); from bench import csrc_hash; print(csrc_hash())")  rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-config5 --images 4 --latent 96 --dtype $dt   (9 steps executed: 2 in compile() + 2 warm-up + 5 timed)" > $O/kernel_stats_images4_latent96_$dt.txt
  python3 $R/tools/prof_summary.py $O/prof_stats5 9 >> $O/kernel_stats_images4_latent96_$dt.txt
  rm -rf $O/prof_stats5
  # HBM traffic of the same shape: FETCH_SIZE and WRITE_SIZE in separate passes, 6 steps each (2 in compile() + 1 warm-up + 3 timed)
  rm -rf $O/prof_fetch5 $O/prof_write5
  rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch5 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-config5 --images 4 --latent 96 --dtype $dt > $O/prof_fetch5.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/prof_write5 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-config5 --images 4 --latent 96 --dtype $dt > $O/prof_write5.log 2>&1
  python3 $R/tools/pmc_summary.py $O/prof_fetch5 $O/prof_write5 $O/pmc_traffic_images4_latent96_$dt.json
  rm -rf $O/prof_fetch5 $O/prof_write5
done
ls -la $O
