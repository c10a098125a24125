// Ingest micro-benchmark (hipcc --offload-arch=gfx950 -O3 tools/ingest_waves.hip -o tools/ingest_waves): how fast can ONE CU pull L2-resident data, as a function of the landing place and of the bytes in flight?
//   mode 0: LDS-DMA (buffer_load ... lds), W waves, D 1-KiB pieces in flight per wave (ring in LDS)
//   mode 1: global_load_dwordx4 into VGPRs, W waves, D loads (1 KiB per wave each) in flight per wave, data xor-ed away
//   mode 2: as mode 1 but every landed piece is also written to LDS (ds_write_b128), the cost a register-staged loader pays
// Every block streams `iters` pieces per wave out of a `span`-byte window (L2 resident when span is small), 256 blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int D>
__global__ void __launch_bounds__(512) k_dma(const char* src, unsigned span, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, span, 0x00020000);
  char* my = smem + wid * D * 1024;
  unsigned off = ((blockIdx.x * nw + wid) * 1024u * 37u) % span;
  for (int it = 0; it < iters; it += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(my + d * 1024), 16, off + lane * 16, 0, 0, 0);
      off += 1024u * nw * 3u; if (off >= span - 1024u) off -= span - 1024u;
    }
    if (it + D < iters) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && lane == 0 && my[0] == 123) sink[0] = 1;
}

template <int D, bool LDSW>
__global__ void __launch_bounds__(512) k_reg(const char* src, unsigned span, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, span, 0x00020000);
  unsigned off = ((blockIdx.x * nw + wid) * 1024u * 37u) % span;
  i4 r[D];
  i4 acc = {0, 0, 0, 0};
  char* my = smem + wid * 4096 + lane * 16;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    r[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + lane * 16, 0, 0);
    off += 1024u * nw * 3u; if (off >= span - 1024u) off -= span - 1024u;
  }
  for (int it = D; it < iters; it += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      // consume the oldest, refill its slot
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D - 1) : "memory");
      if (LDSW) *reinterpret_cast<i4*>(my + (d & 3) * 1024) = r[d]; else acc ^= r[d];
      r[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + lane * 16, 0, 0);
      off += 1024u * nw * 3u; if (off >= span - 1024u) off -= span - 1024u;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int d = 0; d < D; ++d) acc ^= r[d];
  if (sink && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) sink[0] = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static float timeit(F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}

int main() {
  const unsigned span = 48u << 20;       // bigger than the L2s, inside the 256 MB Infinity Cache; second run: 2 MB (L2 resident)
  char* buf; int* sink; CK(hipMalloc(&buf, span)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, span));
  const int iters = 4096;
  for (unsigned sp : {2u << 20, 48u << 20}) {
    printf("window %u MB\n", sp >> 20);
    for (int waves : {4, 8}) {
      double bytes = 256.0 * waves * iters * 1024.0;
#define RUN(name, kern, smem) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), smem, 0, buf, sp, iters, sink); }); \
        printf("  %-28s waves %d: %7.1f GB/s per CU  (%.2f TB/s chip)\n", name, waves, bytes / ms / 1e6 / 256, bytes / ms / 1e9); }
      RUN("lds-dma  D=8  (8 KB/wave)", (k_dma<8>), waves * 8 * 1024)
      RUN("lds-dma  D=16 (16 KB/wave)", (k_dma<16>), waves * 16 * 1024)
      if (waves == 4) RUN("lds-dma  D=32 (32 KB/wave)", (k_dma<32>), waves * 32 * 1024)
      RUN("vgpr     D=8", (k_reg<8, false>), 64 * 1024)
      RUN("vgpr     D=16", (k_reg<16, false>), 64 * 1024)
      RUN("vgpr     D=32", (k_reg<32, false>), 64 * 1024)
      RUN("vgpr+dsw D=16", (k_reg<16, true>), 64 * 1024)
      RUN("vgpr+dsw D=32", (k_reg<32, true>), 64 * 1024)
    }
  }
  return 0;
}
