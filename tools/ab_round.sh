#!/bin/bash
# tools/ab_round.sh ROUNDS OLD_LIB OLD_TABLE [bench args...]: same-box A/B of this round's library + shipped table against an earlier round's library WITH ITS OWN TABLE
# (an old library drops table rows of kernel variants it does not know and would run those shapes on a fallback tile), both arms with identical settings otherwise;
# alternating, one bench process per measurement.  GPU box only.  e.g. tools/ab_round.sh 3 _r04lib/libtinyfusers_hip.so _r04lib/gemm_tune_r04.txt --images 4 --latent 96
# The old library and its table are not in the history (_rNNlib/ is git-ignored: binaries).  To recreate round 4's (commit d7b0887, "round 4: VERDICT + ADVICE + BENCH"):
#   git worktree add /tmp/r04 d7b0887 && (cd /tmp/r04 && python -m tinyfusers_amd.build) && mkdir -p _r04lib && cp /tmp/r04/tinyfusers_amd/lib/libtinyfusers_hip.so _r04lib/
#   git show d7b0887:tinyfusers_amd/gemm_tune_gfx950.txt > _r04lib/gemm_tune_r04.txt
# (the directory travels to the GPU box with gpurun: it is not in .gpurunignore)
R=$1; OLD=$2; OLDT=$3; shift 3
mkdir -p gpurun_out
one() {   # label, env assignments (space separated)
  ( export $2 TF_LIB_ALLOW_MISSING=1 TF_GEMM_AUTOTUNE=1; exec python bench.py --steps ${AB_STEPS:-100} --warmup 10 --no-cpu-baseline --no-roofline --no-e2e --no-config5 "${@:3}" > gpurun_out/ab.json 2> gpurun_out/ab.err )
  python - "$1" <<'PY'
import json, sys
try:
    print("%-10s %.4f ms/step" % (sys.argv[1], json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])["ms_per_step"]), flush=True)
except Exception as e:
    print(sys.argv[1], "failed:", e, open("gpurun_out/ab.err").read()[-400:], flush=True)
PY
}
for i in $(seq 1 $R); do
  one "old[$i]" "TF_LIB_PATH=$PWD/$OLD TF_GEMM_TUNE_TABLE=$PWD/$OLDT" "$@"
  one "new[$i]" "TF_AB_DUMMY=1" "$@"
done
