// k_gemm_c4 instances: the persistent short-K kernel (csrc/gemm.hip is the host side: c4_ok; gemm_c4.h the kernel)
#include "gemm_c4.h"
#include <stdlib.h>
static int c4_num_cus() {
  static int n = 0;
  if (!n) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256; }
  return n;
}
static int g_c4_chunk = getenv("TF_C4_CHUNK") ? atoi(getenv("TF_C4_CHUNK")) : 0;   // A/B: tiles per chunk of k_gemm_c4's walk (0 = per-shape choice)
int TFK(tfk_launch_c4)(const GemmP& p, hipStream_t st) {
  constexpr int smem = 2 * (128 + 128) * 128 + 4 * 64 * 8;   // the two-slot ring (the epilogue's patches live in slot 1) + the LayerNorm row-sum exchange
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_gemm_c4<false, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    TF_HIP(hipFuncSetAttribute((const void*)k_gemm_c4<true, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  const int tiles = p.ntm * p.ntn;
  GemmP q = p;
  // consecutive tiles per block.  Without the LayerNorm fold: one (chunks of 2-8 were 2-8 % faster on three narrow-N shapes and up to 6x slower
  // wherever they left fewer chunks than blocks).  With it (n-fastest order): 4 or 2 while every block still gets >= 4 chunks -- the statistics
  // of a row block are computed once per chunk
  int chunk = 1;
  if (p.ln_colsum && p.order == 0) chunk = tiles / 4 >= 8 * c4_num_cus() ? 4 : tiles / 2 >= 8 * c4_num_cus() ? 2 : 1;
  q.c4_chunk = g_c4_chunk > 0 ? g_c4_chunk : chunk;
  const int chunks = (tiles + q.c4_chunk - 1) / q.c4_chunk;
  const int grid = chunks < 2 * c4_num_cus() ? chunks : 2 * c4_num_cus();   // two resident blocks per CU walk the tile list
  if (p.ln_colsum) hipLaunchKernelGGL((k_gemm_c4<true, kBF>), dim3(grid), dim3(256), smem, st, q);
  else hipLaunchKernelGGL((k_gemm_c4<false, kBF>), dim3(grid), dim3(256), smem, st, q);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
