// Microbenchmark: per-CU LDS ingest rate with global_load_lds_dwordx4 (the staging primitive of k_igemm),
// as a function of the source footprint (L2 / Infinity Cache / HBM), ring depth and waves per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// each block streams `iters` tiles of TILE_KB; tile t of block b comes from offset ((b*stride_blocks + t) * TILE) % footprint
template <int ROUNDS, int DEPTH, int THREADS>
__global__ void __launch_bounds__(THREADS) k_stream(const char* __restrict__ src, size_t footprint, int iters, int share, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = ROUNDS * THREADS * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // `share` consecutive blocks read the same tiles (models operand sharing inside an XCD)
  size_t base_tile = (size_t)(blockIdx.x / share) * iters;
  auto stage = [&](int buf, int t) {
    size_t off = ((base_tile + t) * (size_t)TILE) % footprint;
    const char* s = src + off;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const char* g = s + (size_t)r * THREADS * 16 + tid * 16;
      char* l = smem + buf * TILE + (r * (THREADS / 64) + wid) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
  };
  for (int s_ = 0; s_ < DEPTH - 1; ++s_) if (s_ < iters) stage(s_, s_);
  float acc = 0.f;
  for (int t = 0; t < iters; ++t) {
    int newer = iters - 1 - t;
    if (newer >= DEPTH - 2) wait_vm<(DEPTH - 2) * ROUNDS>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + DEPTH - 1 < iters) stage((t + DEPTH - 1) % DEPTH, t + DEPTH - 1);
    acc += *(const float*)(smem + (t % DEPTH) * TILE + lane * 4);   // touch LDS so nothing is optimised away
  }
  if (acc == 12345.678f) sink[0] = acc;
}

template <int ROUNDS, int DEPTH, int THREADS>
void run(const char* buf, size_t footprint, int blocks, int iters, int share, float* sink, const char* tag) {
  constexpr int TILE = ROUNDS * THREADS * 16;
  int smem = TILE * DEPTH;
  CK(hipFuncSetAttribute((const void*)k_stream<ROUNDS, DEPTH, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 2; ++w) k_stream<ROUNDS, DEPTH, THREADS><<<blocks, THREADS, smem>>>(buf, footprint, iters, share, sink);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) k_stream<ROUNDS, DEPTH, THREADS><<<blocks, THREADS, smem>>>(buf, footprint, iters, share, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  double bytes = (double)blocks * iters * TILE;
  printf("%-34s tile %3d KB depth %d thr %4d blocks %4d share %3d footprint %7.1f MB : %7.1f us  %7.2f TB/s  %6.1f GB/s per block\n", tag, TILE / 1024, DEPTH,
         THREADS, blocks, share, footprint / 1048576.0, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / blocks / (ms * 1e-3) / 1e9);
}

int main() {
  size_t cap = (size_t)4 << 30;
  char* buf; CK(hipMalloc(&buf, cap)); CK(hipMemset(buf, 1, cap));
  float* sink; CK(hipMalloc(&sink, 4));
  const int iters = 256;
  size_t fps[] = {(size_t)1 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)2 << 30};
  for (size_t fp : fps) {
    for (int share : {1, 32, 256}) {
      run<8, 4, 256>(buf, fp, 256, iters, share, sink, "32KB tiles x4 ring, 4 waves");
    }
  }
  size_t fp = (size_t)16 << 20;
  run<8, 2, 256>(buf, fp, 256, iters, 32, sink, "depth 2");
  run<8, 3, 256>(buf, fp, 256, iters, 32, sink, "depth 3");
  run<4, 4, 256>(buf, fp, 256, iters * 2, 32, sink, "16KB tiles");
  run<4, 4, 512>(buf, fp, 256, iters, 32, sink, "8 waves");
  run<2, 4, 1024>(buf, fp, 256, iters, 32, sink, "16 waves");
  run<8, 4, 256>(buf, fp, 512, iters, 32, sink, "2 blocks/CU");   // 128 KB LDS each -> only 1 resident; see below
  run<4, 4, 256>(buf, fp, 512, iters, 32, sink, "2 blocks/CU 16KB tiles");
  run<2, 4, 256>(buf, fp, 1024, iters, 32, sink, "4 blocks/CU 8KB tiles");
  run<8, 4, 256>(buf, fp, 128, iters, 32, sink, "128 blocks");
  return 0;
}
