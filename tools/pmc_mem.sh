#!/bin/bash
# Memory-side counters of the step per kernel family (L2 hit rate, L1->L2 read latency, TA stalls): three small rocprofv3 --pmc passes
# (a pass that asks for more than the TCC block can collect aborts), summarised by tools/pmc_sq_summary.py into gpurun_out/pmc_mem_*.json
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
pass() {   # name, counters...
  n=$1; shift
  rm -rf $O/prof_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/prof_$n --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e > $O/prof_$n.log 2>&1
  python3 $R/tools/pmc_sq_summary.py $O/prof_$n $O/pmc_mem_$n.json > /dev/null
  rm -rf $O/prof_$n
  echo "pass $n done"
}
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
# (a TA_* + TCP_TCR pass aborts in the profiler on this image: left out)
