// k_igemm instances of the 64-wide tiles and the bfloat16 instances (csrc/gemm.hip is the host side; gemm_igemm.h the kernel)
#include "gemm_k_igemm.inc"
int tfk_launch_igemm_64(const GemmP& p, hipStream_t st, int bm, bool wide, bool all8) {
  if (bm == 128) return launch_cfg<128, 64, true>(p, st, wide, all8);
  if (bm == 64) return launch_cfg<64, 64, true>(p, st, wide, all8);
  tf_set_error("run_gemm: no kernel for tile %dx64", bm);
  return TF_E_UNSUPPORTED;
}
// bfloat16 instances: plain deep ring, one launch (no split-K: the reduce kernels are fp16), no statistics, no input GroupNorm
template <int BM, int BN, bool GENERIC>
static int launch_bf(const GemmP& p, hipStream_t st) {
  const int smem = igemm_lds_bytes(BM, BN, false);
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm<BM, BN, GENERIC, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_igemm<BM, BN, GENERIC, false, false, false, true>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tfk_launch_igemm_bf16(const GemmP& p, hipStream_t st, int bm, int bn) {
  const bool g = gemm_generic(p);
  if (bm == 128 && bn == 128) return g ? launch_bf<128, 128, true>(p, st) : launch_bf<128, 128, false>(p, st);
  if (bm == 64 && bn == 64) return g ? launch_bf<64, 64, true>(p, st) : launch_bf<64, 64, false>(p, st);
  tf_set_error("run_gemm: no bfloat16 kernel for tile %dx%d", bm, bn);
  return TF_E_UNSUPPORTED;
}
