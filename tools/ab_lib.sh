#!/bin/bash
# tools/ab_lib.sh ROUNDS OLD_LIB [bench args...]: same-box A/B of the built library against an earlier one (e.g. _r03lib/libtinyfusers_hip.so),
# alternating, one bench process per measurement (200 graph-replayed steps each); prints ms per step per round.
# BOTH arms run with TF_HOIST_STEP_INVARIANTS=0: a library from before round 4 has no tf_set_step_params_copy, and switching the hoisting off for the OLD arm only
# (as this script did until round 5) handicaps it by about 1 % -- that is how round 5 first read a -0.65 % "gain" that a fair comparison showed to be +0.16 %
# (profiles/r05_ab.txt blocks 1-2).  Both arms also read the CURRENT tuning table: an old library drops rows of kernel variants it does not know and runs those shapes
# on a fallback tile.  For a library that has its own table and the hoisting entry (round 4 on), use tools/ab_round.sh instead.
R=$1; OLD=$2; shift 2
mkdir -p gpurun_out
one() {   # label, env assignment
  ( export $2 TF_LIB_ALLOW_MISSING=1 TF_GEMM_AUTOTUNE=1; exec python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-roofline --no-e2e --no-config5 "${@:3}" > gpurun_out/ab.json 2> gpurun_out/ab.err )
  python - "$1" <<'PY'
import json, sys
try:
    print("%-10s %.4f ms/step" % (sys.argv[1], json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])["ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "failed:", e, open("gpurun_out/ab.err").read()[-300:])
PY
}
for i in $(seq 1 $R); do
  one "old[$i]" "TF_LIB_PATH=$OLD TF_HOIST_STEP_INVARIANTS=0" "$@"     # (an earlier round's library has no tf_set_step_params_copy)
  one "new[$i]" "TF_HOIST_STEP_INVARIANTS=0" "$@"
done
