#!/bin/bash
# k_gemm_ar timing experiments (tagged builds -DTF_AR_EXP=mask: results are WRONG by design): what a K step waits for.  GPU box only.
# mask bits: 1 no stores by the loaders, 2 no MFMAs, 4 no fragment reads, 8 no LDS-DMA
for e in ${AR_EXPS:-base 9 10 12 13 14}; do
  if [ $e = base ]; then lib=tinyfusers_amd/lib/libtinyfusers_hip.so; else lib=tinyfusers_amd/lib/libtinyfusers_hip_arexp$e.so; fi
  echo "== TF_AR_EXP=$e"
  TF_LIB_PATH=$PWD/$lib TF_LIB_ALLOW_MISSING=1 C4_K320=1 C4_ONLY=c5 C4_AR_ONLY=1 timeout -k 10 120 python tools/c4_bench.py 2>&1 | cut -c150-200
done
