// k_gemm_ar instances: the activation-resident short-K kernel (csrc/gemm.hip is the host side: ar_ok; gemm_ar.h the kernel)
#include "gemm_ar.h"
#include <stdlib.h>
static int ar_num_cus() {
  static int n = 0;
  if (!n) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256; }
  return n;
}
int TFK(tfk_launch_ar)(const GemmP& p, hipStream_t st) {
  constexpr int smem = (5 + 4) * 128 * 128 + 4 * 2 * 16 * 128;   // the resident panel (up to 5 K tiles) + the 4-slot weight ring + two 16-row patches per consumer wave: all 160 KiB
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_gemm_ar<false, false, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    TF_HIP(hipFuncSetAttribute((const void*)k_gemm_ar<true, false, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    TF_HIP(hipFuncSetAttribute((const void*)k_gemm_ar<false, true, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    TF_HIP(hipFuncSetAttribute((const void*)k_gemm_ar<true, true, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  if (p.ktiles < 4 || p.ktiles > 5 || p.ktiles * 64 != p.K || p.residual) { tf_set_error("k_gemm_ar: K = %d is not 256 or 320, or a residual", p.K); return TF_E_UNSUPPORTED; }
  const int tiles = p.ntm * p.ntn;
  GemmP q = p;
  // every block the same run length of the n-fastest tile list: one resident 8-wave block per CU
  const int cus = ar_num_cus();
  const int blocks = tiles < cus ? tiles : cus;
  q.c4_chunk = (tiles + blocks - 1) / blocks;
  const int grid = (tiles + q.c4_chunk - 1) / q.c4_chunk;
  const bool gg = p.act == 1;
  if (p.ln_colsum && gg) hipLaunchKernelGGL((k_gemm_ar<true, true, kBF>), dim3(grid), dim3(512), smem, st, q);
  else if (p.ln_colsum) hipLaunchKernelGGL((k_gemm_ar<true, false, kBF>), dim3(grid), dim3(512), smem, st, q);
  else if (gg) hipLaunchKernelGGL((k_gemm_ar<false, true, kBF>), dim3(grid), dim3(512), smem, st, q);
  else hipLaunchKernelGGL((k_gemm_ar<false, false, kBF>), dim3(grid), dim3(512), smem, st, q);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
