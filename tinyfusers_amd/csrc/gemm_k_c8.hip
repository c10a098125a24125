// k_gemm_c8 instances: the 256-row persistent short-K kernel (csrc/gemm.hip is the host side: c8_ok; gemm_c8.h the kernel)
#include "gemm_c8.h"
#include <stdlib.h>
static int c8_num_cus() {
  static int n = 0;
  if (!n) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256; }
  return n;
}
int TFK(tfk_launch_c8)(const GemmP& p, hipStream_t st) {
  constexpr int smem = 3 * (256 + 128) * 128 + 8 * 64 * 8;   // the three-slot ring (the epilogue's patches live in the slot the next two K tiles do not use) + the LayerNorm row-sum exchange
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_gemm_c8<false, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    TF_HIP(hipFuncSetAttribute((const void*)k_gemm_c8<true, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  const int tiles = p.ntm * p.ntn;
  GemmP q = p;
  // consecutive tiles per block: with the LayerNorm fold (n-fastest order) 2 or 4 while every block still gets several chunks -- a row block's statistics are computed once per chunk
  int chunk = 1;
  if (p.ln_colsum && p.order == 0) chunk = tiles / 4 >= 4 * c8_num_cus() ? 4 : tiles / 2 >= 4 * c8_num_cus() ? 2 : 1;
  q.c4_chunk = chunk;
  const int chunks = (tiles + chunk - 1) / chunk;
  const int grid = chunks < c8_num_cus() ? chunks : c8_num_cus();   // one resident 8-wave block per CU walks the tile list
  if (p.ln_colsum) hipLaunchKernelGGL((k_gemm_c8<true, kBF>), dim3(grid), dim3(512), smem, st, q);
  else hipLaunchKernelGGL((k_gemm_c8<false, kBF>), dim3(grid), dim3(512), smem, st, q);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
