// k_igemm_pp3 instances (csrc/gemm.hip is the host side: pp3_setup; gemm_pp3.h the kernel)
#include "gemm_pp3.h"

template <int BN>
static int launch_pp3(const GemmP& p, hipStream_t st) {
  constexpr int scratch = 4 * 48 * (BN / 2 + 4) * 4, tail = 96 * 8 + 4 * BN * 8 + 3 * BN * 4;      // the epilogue's share (as launch_pp2 with BM = 192)
  const int ring = 2 * p.pt_stage + 3 * BN * 128;
  const int smem = ring > scratch + tail ? ring : scratch + tail;
  if (smem > 163840) { tf_set_error("k_igemm_pp3: %d bytes of LDS", smem); return TF_E_UNSUPPORTED; }
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm_pp3<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_igemm_pp3<BN>), dim3(p.ntm * p.ntn), dim3(512), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tfk_launch_pp3(const GemmP& p, hipStream_t st, int bn) {
  if (bn == 160) return launch_pp3<160>(p, st);
  if (bn == 128) return launch_pp3<128>(p, st);
  tf_set_error("k_igemm_pp3: no instance for a %d-wide tile", bn);
  return TF_E_UNSUPPORTED;
}
