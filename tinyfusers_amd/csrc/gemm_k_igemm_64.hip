// k_igemm instances of the 64-wide tiles (csrc/gemm.hip is the host side; gemm_igemm.h the kernel)
#include "gemm_k_igemm.inc"
int TFK(tfk_launch_igemm_64)(const GemmP& p, hipStream_t st, int bm, bool wide, bool all8) {
  if (bm == 128) return launch_cfg<128, 64, true>(p, st, wide, all8);
  if (bm == 64) return launch_cfg<64, 64, true>(p, st, wide, all8);
  tf_set_error("run_gemm: no kernel for tile %dx64", bm);
  return TF_E_UNSUPPORTED;
}
