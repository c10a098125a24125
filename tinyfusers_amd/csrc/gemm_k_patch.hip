// k_igemm_patch instances (csrc/gemm.hip is the host side: patch_setup fills the geometry; gemm_patch.h the kernel)
#include "gemm_patch.h"
template <int BM, int BN>
static int launch_patch(const GemmP& p, hipStream_t st) {
  constexpr int TM = BM / 2, TN = BN / 2;
  constexpr int scratch = 4 * TM * (TN + 4) * 4, tail = BM * 8 + 4 * BN * 8;
  const int ring = 2 * p.pt_ppc * 1024 + p.pt_ns * p.pt_stage + gi_table_bytes(p);
  const int smem = ring > scratch + tail ? ring : scratch + tail;
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm_patch<BM, BN, false, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm_patch<BM, BN, true, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  if (p.gi_part) hipLaunchKernelGGL((k_igemm_patch<BM, BN, true, kBF>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), smem, st, p);
  else hipLaunchKernelGGL((k_igemm_patch<BM, BN, false, kBF>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int TFK(tfk_launch_patch)(const GemmP& p, hipStream_t st, int bm, int bn) {
  if (bm == 128 && bn == 160) return launch_patch<128, 160>(p, st);
  if (bm == 64 && bn == 160) return launch_patch<64, 160>(p, st);
  if (bm == 128 && bn == 128) return launch_patch<128, 128>(p, st);
  if (bm == 64 && bn == 128) return launch_patch<64, 128>(p, st);
  tf_set_error("run_gemm: no patch kernel for tile %dx%d", bm, bn);
  return TF_E_UNSUPPORTED;
}
