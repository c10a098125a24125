// Does LDS-DMA transfer overlap with a ds_read + MFMA chain of the same waves?  (design probe for k_igemm)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE bit0: do loads; bit1: do ds_read+MFMA chain (NMFMA per iteration per wave); LOADERS: 0 = all waves load, 1 = only waves >= 4 load (producer waves)
template <int ROUNDS, int DEPTH, int THREADS, int NMFMA, int NREAD>
__global__ void __launch_bounds__(THREADS) k(const char* __restrict__ src, size_t footprint, int iters, int mode, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = ROUNDS * THREADS * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)footprint, 0x00020000);
  size_t base_tile = (size_t)(blockIdx.x / 32) * iters;
  auto stage = [&](int buf, int t) {
    if (!(mode & 1)) return;
    unsigned off = (unsigned)(((base_tile + t) * (size_t)TILE) % footprint);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      char* l = smem + buf * TILE + (r * (THREADS / 64) + wid) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)l, 16, off + r * THREADS * 16 + tid * 16, 0, 0, 0);
    }
  };
  for (int s_ = 0; s_ < DEPTH - 1; ++s_) if (s_ < iters) stage(s_, s_);
  f4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  for (int t = 0; t < iters; ++t) {
    int newer = iters - 1 - t;
    if (newer >= DEPTH - 2) wait_vm<(DEPTH - 2) * ROUNDS>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + DEPTH - 1 < iters) stage((t + DEPTH - 1) % DEPTH, t + DEPTH - 1);
    if (mode & 2) {
      const char* b = smem + (t % DEPTH) * TILE;
      h8 f[NREAD];
#pragma unroll
      for (int i = 0; i < NREAD; ++i) f[i] = *reinterpret_cast<const h8*>(b + ((i * 64 + lane) * 16) % TILE);
#pragma unroll
      for (int i = 0; i < NMFMA; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[i % NREAD], f[(i + 1) % NREAD], acc[i & 3], 0, 0, 0);
    }
  }
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (s == 12345.678f) sink[0] = s;
}

template <int ROUNDS, int DEPTH, int THREADS, int NMFMA, int NREAD>
void run(const char* buf, size_t footprint, int blocks, int iters, float* sink, const char* tag) {
  constexpr int TILE = ROUNDS * THREADS * 16;
  int smem = TILE * DEPTH;
  auto kern = k<ROUNDS, DEPTH, THREADS, NMFMA, NREAD>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  float t[4];
  for (int mode = 1; mode <= 3; ++mode) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) kern<<<blocks, THREADS, smem>>>(buf, footprint, iters, mode, sink);
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) kern<<<blocks, THREADS, smem>>>(buf, footprint, iters, mode, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); t[mode] = ms / 5 * 1e3;
  }
  double gbs = (double)iters * TILE / (t[1] * 1e-6) / 1e9;
  printf("%-40s tile %2d KB x%d thr %3d mfma/iter/wave %2d reads %2d: loads %6.1f us (%5.1f GB/s/CU) | chain %6.1f us | both %6.1f us (sum %6.1f, max %6.1f)\n", tag, TILE / 1024, DEPTH,
         THREADS, NMFMA, NREAD, t[1], gbs, t[2], t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
}

int main() {
  size_t cap = (size_t)64 << 20;
  char* buf; CK(hipMalloc(&buf, cap)); CK(hipMemset(buf, 0, cap));
  float* sink; CK(hipMalloc(&sink, 4));
  run<7, 4, 256, 20, 7>(buf, cap, 256, 256, sink, "4 waves, 28KB tile (64x160), 20 mfma");
  run<9, 4, 256, 40, 9>(buf, cap, 256, 256, sink, "4 waves, 36KB tile (128x160), 40 mfma");
  run<4, 4, 512, 10, 7>(buf, cap, 256, 256, sink, "8 waves, 32KB tile, 10 mfma");
  run<4, 4, 512, 20, 9>(buf, cap, 256, 256, sink, "8 waves, 32KB tile, 20 mfma");
  run<4, 4, 512, 40, 13>(buf, cap, 256, 256, sink, "8 waves, 32KB tile, 40 mfma");
  run<8, 2, 512, 80, 13>(buf, cap, 256, 128, sink, "8 waves, 64KB tile x2, 80 mfma");
  run<2, 4, 1024, 10, 7>(buf, cap, 256, 256, sink, "16 waves, 32KB tile, 10 mfma");
  return 0;
}
