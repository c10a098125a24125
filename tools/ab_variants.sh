#!/bin/bash
# tools/ab_variants.sh ROUNDS "CMD" TAG...: same-box A/B of tagged experimental builds (python -m tinyfusers_amd.build --tag T -D...) against the
# shipped library: runs CMD once per library and round, alternating; CMD's stdout is prefixed with the tag ("base" = the shipped library).
R=$1; CMD=$2; shift 2
for i in $(seq 1 $R); do
  for t in base "$@"; do
    if [ "$t" = base ]; then L=""; else L="$(pwd)/tinyfusers_amd/lib/libtinyfusers_hip_$t.so"; fi
    ( export TF_LIB_PATH=$L; $CMD 2>&1 | sed "s/^/[$t r$i] /" )
  done
done
