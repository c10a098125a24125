// k_igemm instances of the 128-wide tiles (csrc/gemm.hip is the host side; gemm_igemm.h the kernel)
#include "gemm_k_igemm.inc"
int TFK(tfk_launch_igemm_128)(const GemmP& p, hipStream_t st, int bm, bool wide, bool all8) {
  if (bm == 128) return launch_cfg<128, 128, true>(p, st, wide, all8);
  if (bm == 64) return launch_cfg<64, 128, true>(p, st, wide, all8);
  tf_set_error("run_gemm: no kernel for tile %dx128", bm);
  return TF_E_UNSUPPORTED;
}
int TFK(tfk_launch_igemm_256x128)(const GemmP& p, hipStream_t st) {
  if (gemm_generic(p) || p.gi_part) { tf_set_error("run_gemm: the 256x128 tile needs channel counts on the 64 grid and no input GroupNorm"); return TF_E_UNSUPPORTED; }
  return launch_cfg3<256, 128, false, false>(p, st);
}
