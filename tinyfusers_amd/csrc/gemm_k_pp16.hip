// k_igemm_pp, fp16 instances (csrc/gemm.hip is the host side: pp_ok; gemm_pp.h the kernel)
#include "gemm_k_pp.inc"
template <int BN, int BM>
static int launch_pp16(const GemmP& p, hipStream_t st, int np_force) {
  const bool fast = pp_fast(p);
  constexpr bool CAN1 = (163840 / ((BM + BN) * 128)) >= 3;
  const bool np1 = CAN1 && np_force != 2;
  if constexpr (CAN1) {
    if (p.ln_colsum) return launch_pp2<BN, 1, true, false, false, BM, true>(p, st);    // (pp_ok admits linears only: the lean addressing)
    if (np1) return fast ? launch_pp2<BN, 1, true, false, false, BM>(p, st) : launch_pp2<BN, 1, false, false, false, BM>(p, st);
  }
  if constexpr (BM == 256) {
    if (p.ln_colsum) return launch_pp2<BN, 2, true, false, false, 256, true>(p, st);
    return fast ? launch_pp2<BN, 2, true>(p, st) : launch_pp2<BN, 2, false>(p, st);
  }
  else { tf_set_error("k_igemm_pp: the 192-row tile has the one-phase form only"); return TF_E_UNSUPPORTED; }
}
int TFK(tfk_launch_pp16)(const GemmP& p, hipStream_t st, int bm, int bn, int np_force) {
  if (bm == 192 && bn == 128) return launch_pp16<128, 192>(p, st, np_force);
  if (bm == 192 && bn == 160) return launch_pp16<160, 192>(p, st, np_force);
  if (bm == 256 && bn == 128) return launch_pp16<128, 256>(p, st, np_force);
  if (bm == 256 && bn == 160) return launch_pp16<160, 256>(p, st, np_force);
  if (bm == 256 && bn == 256) return launch_pp16<256, 256>(p, st, np_force);
  tf_set_error("k_igemm_pp: no fp16 instance for tile %dx%d", bm, bn);
  return TF_E_UNSUPPORTED;
}
