#!/bin/bash
# HBM traffic of the step (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes, per the guide's gfx950 correction: tools/pmc_summary.py),
# for the default workload and for config 5's per-GPU shape in both dtypes; results in gpurun_out/pmc_traffic*.json (stamped with the kernel-source hash)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() {   # name, bench arguments
  rm -rf $O/prof_fetch $O/prof_write
  rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-config5 $2 > $O/prof_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/prof_write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-config5 $2 > $O/prof_write.log 2>&1
  python3 $R/tools/pmc_summary.py $O/prof_fetch $O/prof_write $O/pmc_traffic$1.json > /dev/null
  rm -rf $O/prof_fetch $O/prof_write
}
one "" ""
one _images4_latent96_fp16 "--images 4 --latent 96 --dtype fp16"
one _images4_latent96_fp8 "--images 4 --latent 96 --dtype fp8"
