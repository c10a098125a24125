#!/bin/bash
# Per-kernel micro-benchmarks of the round (GPU box, from the repo root): every output is stamped with the hash of the kernel sources it ran on.
#   results land in gpurun_out/micro_*.txt; copy them to profiles/r0N_*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
H=$(python3 -c "import sys; sys.path.insert(0, '$R'); from bench import csrc_hash; print(csrc_hash())")
run() {   # name, command...
  local name=$1; shift
  { echo "# csrc $H  $*"; timeout -k 10 600 "$@" 2>&1 | grep -v "amdgpu.ids"; } > $O/micro_$name.txt
}
run c4_bench python3 $R/tools/c4_bench.py
run pp_bench python3 $R/tools/pp_bench.py all
run pp_bench8 python3 $R/tools/pp_bench8.py
run sdpa_bench python3 $R/tools/sdpa_bench.py
run sdpa_bench_4wave env TF_SDPA_NW=4 python3 $R/tools/sdpa_bench.py
run sdpa_dbg python3 $R/tools/sdpa_dbg.py
run pp_dbg python3 $R/tools/pp_dbg.py
ls -la $O/micro_*.txt
