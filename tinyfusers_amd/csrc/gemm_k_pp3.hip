// k_igemm_pp3 instances (csrc/gemm.hip is the host side: pp3_setup; gemm_pp3.h the kernel)
#include "gemm_pp3.h"

template <int BN, int W, bool F8 = false, bool H2 = false, bool BFK = false>
static int launch_pp3(const GemmP& p, hipStream_t st) {
  constexpr int PROWS = (192 / W + 2) * (W + 4), PB = ((PROWS + 7) / 8) * 1024;
  constexpr int NS = 3;
  constexpr int scratch = 4 * 48 * (BN / 2 + 4) * 4, tail = 96 * 8 + 4 * BN * 8 + 3 * BN * 4;      // the epilogue's share (as launch_pp2 with BM = 192)
  constexpr int ring = 2 * PB + NS * BN * 128 + (F8 ? (H2 ? 4 : 2) * ((PROWS + 63) / 64) * 256 : (2 * 192 * 128 <= PB ? 0 : 192 * 128));   // (+ the scale patches / slot 1 of the extra segment)
  constexpr int smem = ring > scratch + tail ? ring : scratch + tail;
  static_assert(smem <= 163840, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm_pp3<BN, W, F8, H2, BFK>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_igemm_pp3<BN, W, F8, H2, BFK>), dim3(p.ntm * p.ntn), dim3(512), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
// instances: the tile width that wins at each level of BASELINE config 5 (tools/pp3_bench.py: 160 for the 320-channel convs on 96-pixel rows, 128 for
// 640 / 1280 channels on 48 / 24-pixel rows); pp3_setup (csrc/gemm.hip) admits exactly these
int TFK(tfk_launch_pp3)(const GemmP& p, hipStream_t st, int bn) {
#if !TF_TU_BF
  if (p.fp8) {                                             // block-scaled e4m3: 128-wide tiles; H2 = the form for channel counts on the 64 grid (it takes the 128 grid as well)
    const bool h2 = (p.C1 % 128) != 0;
    if (p.mx && bn == 128 && p.Wo == 96) return launch_pp3<128, 96, true, true>(p, st);
    if (p.mx && bn == 128 && p.Wo == 48) return h2 ? launch_pp3<128, 48, true, true>(p, st) : launch_pp3<128, 48, true>(p, st);
    if (p.mx && bn == 128 && p.Wo == 24 && !h2) return launch_pp3<128, 24, true>(p, st);
    tf_set_error("k_igemm_pp3: no e4m3 instance for a %d-wide tile on %d-pixel rows (mx=%d)", bn, p.Wo, p.mx);
    return TF_E_UNSUPPORTED;
  }
#endif
  if (bn == 160 && p.Wo == 96) return launch_pp3<160, 96, false, false, kBF>(p, st);
  if (bn == 128 && p.Wo == 48) return launch_pp3<128, 48, false, false, kBF>(p, st);
  if (bn == 128 && p.Wo == 24) return launch_pp3<128, 24, false, false, kBF>(p, st);
  tf_set_error("k_igemm_pp3: no instance for a %d-wide tile on %d-pixel rows", bn, p.Wo);
  return TF_E_UNSUPPORTED;
}
