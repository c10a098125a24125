// k_igemm instances of the 160-wide tiles (csrc/gemm.hip is the host side; gemm_igemm.h the kernel)
#include "gemm_k_igemm.inc"
int TFK(tfk_launch_igemm_160)(const GemmP& p, hipStream_t st, int bm, bool wide, bool all8) {
  if (bm == 128) return launch_cfg<128, 160, false>(p, st, wide, all8);     // scratch 86 KB: one block per CU only
  if (bm == 64) return launch_cfg<64, 160, true>(p, st, wide, all8);
  tf_set_error("run_gemm: no kernel for tile %dx160", bm);
  return TF_E_UNSUPPORTED;
}
