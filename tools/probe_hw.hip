// Hardware-assumption probe for gfx950 (run once on the GPU box; output kept under profiles/).
// Verifies with exact integer data: MFMA f16 16x16x32 / 16x16x16 operand+accumulator lane maps,
// ds_read_b64_tr_b16 gather semantics, global_load_lds lane-linear destination, and prints device props.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// A[16][32], B[32][16] row-major f16 in global; C[16][16] f32 out.
__global__ void k_mfma32(const _Float16* A, const _Float16* B, float* C) {
  int l = threadIdx.x, r = l & 15, g = l >> 4;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 32 + 8 * g + j]; b[j] = B[(8 * g + j) * 16 + r]; }
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = acc[i];
}
// A[16][16], B[16][16]
__global__ void k_mfma16(const _Float16* A, const _Float16* B, float* C) {
  int l = threadIdx.x, r = l & 15, g = l >> 4;
  h4 a, b;
  for (int j = 0; j < 4; ++j) { a[j] = A[r * 16 + 4 * g + j]; b[j] = B[(4 * g + j) * 16 + r]; }
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = acc[i];
}
// LDS image [R=16 rows][16 cols] of shorts, value = row*16+col. Each 16-lane group g reads rows 4g..4g+3.
__global__ void k_tr(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[16 * 16];
  int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (short)i;
  __syncthreads();
  int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  const short* addr = lds + (4 * g + q) * 16 + 4 * p;
  s4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)addr);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = t[j];
}
// global_load_lds: each lane gives its own source (reversed order), dest should be base + lane*16.
__global__ void k_glds(const int* src, int* out) {
  __shared__ __attribute__((aligned(16))) int lds[64 * 4 * 2];
  int l = threadIdx.x;
  const int* s = src + (63 - l) * 4;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                   (__attribute__((address_space(3))) void*)(lds + 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = lds[256 + l * 4 + j];
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s arch=%s CUs=%d clock=%d kHz memclk=%d kHz bus=%d bits L2=%d B smem/block=%zu maxSmemPerCU=%zu regs/block=%d warp=%d totalMem=%.1f GiB\n",
         p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.memoryBusWidth, p.l2CacheSize,
         p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.warpSize, p.totalGlobalMem / 1073741824.0);
  int fails = 0;
  { // mfma 16x16x32
    std::vector<_Float16> A(16 * 32), B(32 * 16); std::vector<float> C(256), R(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (_Float16)((i * 3 + k * 5) % 7 - 3);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (_Float16)((k * 2 + j * 7) % 5 - 2);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += (float)A[i * 32 + k] * (float)B[k * 16 + j];
    _Float16 *dA, *dB; float* dC; CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += C[i] != R[i];
    printf("mfma_f32_16x16x32_f16 lane map: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad); fails += bad != 0;
  }
  { // mfma 16x16x16
    std::vector<_Float16> A(256), B(256); std::vector<float> C(256), R(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (_Float16)((i * 3 + k * 5) % 7 - 3);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (_Float16)((k * 2 + j * 7) % 5 - 2);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 16; ++k) R[i * 16 + j] += (float)A[i * 16 + k] * (float)B[k * 16 + j];
    _Float16 *dA, *dB; float* dC; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += C[i] != R[i];
    printf("mfma_f32_16x16x16f16 lane map: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad); fails += bad != 0;
  }
  { // tr read
    short* d; CK(hipMalloc(&d, 512)); std::vector<short> o(256);
    k_tr<<<1, 64>>>(d); CK(hipDeviceSynchronize()); CK(hipMemcpy(o.data(), d, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int g = l >> 4, i = l & 15; bad += o[l * 4 + j] != (short)((4 * g + j) * 16 + i); }
    printf("ds_read_b64_tr_b16 (lane i <- column i, element q <- row q): %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad); fails += bad != 0;
    if (bad) for (int l = 0; l < 64; ++l) printf("  lane %2d: %d %d %d %d\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  }
  { // glds
    int *s, *d; CK(hipMalloc(&s, 1024)); CK(hipMalloc(&d, 1024)); std::vector<int> h(256), o(256);
    for (int i = 0; i < 256; ++i) h[i] = i; CK(hipMemcpy(s, h.data(), 1024, hipMemcpyHostToDevice));
    k_glds<<<1, 64>>>(s, d); CK(hipDeviceSynchronize()); CK(hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) bad += o[l * 4 + j] != (63 - l) * 4 + j;
    printf("global_load_lds dwordx4 (dest = base + lane*16, per-lane source): %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad); fails += bad != 0;
  }
  printf("probe: %s\n", fails ? "FAILED" : "ALL OK");
  return fails != 0;
}
