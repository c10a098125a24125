// k_igemm8 instances: e4m3 operands on the loader / consumer ring (csrc/gemm.hip is the host side; gemm_igemm8.h the kernel)
#include "gemm_igemm8.h"
template <int BM, int BN>
static int launch8(const GemmP& p, hipStream_t st) {
  constexpr int TM = BM / 2, TN = BN / 2;
  constexpr int ring = ring_slots8(BM, BN) * (BM + BN) * 64;
  constexpr int scratch = 4 * TM * (TN + 4) * 4, tail = BM * 8 + 4 * BN * 8;
  constexpr int smem = ring > scratch + tail ? ring : scratch + tail;
  static_assert(smem <= 163840, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm8<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_igemm8<BM, BN>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
// (a 256x128 tile spills: the compiler keeps two copies of the accumulator set to issue the two k halves independently)
int tfk_launch_igemm8(const GemmP& p, hipStream_t st, int bm, int bn) {
  if (bm == 128 && bn == 128) return launch8<128, 128>(p, st);
  if (bm == 64 && bn == 128) return launch8<64, 128>(p, st);
  if (bm == 128 && bn == 64) return launch8<128, 64>(p, st);
  if (bm == 256 && bn == 64) return launch8<256, 64>(p, st);
  if (bm == 64 && bn == 64) return launch8<64, 64>(p, st);
  tf_set_error("run_gemm: no fp8 kernel for tile %dx%d", bm, bn);
  return TF_E_UNSUPPORTED;
}
