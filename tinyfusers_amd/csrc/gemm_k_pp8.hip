// k_igemm_pp, e4m3 instances on the block-scaled MFMA (csrc/gemm.hip is the host side: pp_ok; gemm_pp.h the kernel)
#include "gemm_k_pp.inc"
template <int BN, int BM>
static int launch_pp8(const GemmP& p, hipStream_t st) {
  const bool h2 = (p.C1 % 128) || (p.C2 % 128) || (p.C3 % 128) || (p.C4 % 128);
  if (!pp_fast(p) || !p.mx) { tf_set_error("k_igemm_pp: the e4m3 instances take block-scaled activations and the lean addressing (stride 1, no up-sampling) only"); return TF_E_UNSUPPORTED; }
  if constexpr (BM == 256 && BN == 160) {                  // (no room for the half-tile scale table behind a 3 x 52 KiB ring)
    if (h2) { tf_set_error("k_igemm_pp: no 256x160 e4m3 instance for channel counts off the 128 grid"); return TF_E_UNSUPPORTED; }
    return launch_pp2<BN, 1, true, true, false, BM>(p, st);
  } else {
    return h2 ? launch_pp2<BN, 1, true, true, true, BM>(p, st) : launch_pp2<BN, 1, true, true, false, BM>(p, st);
  }
}
int tfk_launch_pp8(const GemmP& p, hipStream_t st, int bm, int bn) {
  if (bm == 192 && bn == 128) return launch_pp8<128, 192>(p, st);
  if (bm == 192 && bn == 160) return launch_pp8<160, 192>(p, st);
  if (bm == 256 && bn == 128) return launch_pp8<128, 256>(p, st);
  if (bm == 256 && bn == 160) return launch_pp8<160, 256>(p, st);
  tf_set_error("k_igemm_pp: no e4m3 instance for tile %dx%d", bm, bn);   // (256-wide: the e4m3 form holds a whole K tile's fragments and needs the three-slot ring)
  return TF_E_UNSUPPORTED;
}
