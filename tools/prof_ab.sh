#!/bin/bash
# tools/prof_ab.sh VAR=val1 VAR=val2 ...: rocprofv3 kernel statistics of a short bench run under each setting (GPU box); per-step
# summaries land in gpurun_out/prof_ab_<n>.txt.  12 steps executed per run: 2 in compile() + 2 warm-up + 5 timed + 3 instrumented... counted below.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  rm -rf $O/prof_ab
  ( export $v; rocprofv3 --kernel-trace --stats -d $O/prof_ab --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-roofline --no-config5 > $O/prof_ab.log 2>&1 )
  echo "== $v" > $O/prof_ab_$i.txt
  python3 $R/tools/prof_summary.py $O/prof_ab 14 >> $O/prof_ab_$i.txt 2>&1
  rm -rf $O/prof_ab
done
