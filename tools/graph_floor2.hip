#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct Big { void* p[12]; int v[30]; };
__global__ void k_triv(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_lds(float* p) { extern __shared__ float s[]; if (threadIdx.x == 0) s[0] = p[0]; __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = s[0] + 1.f; }
__global__ void k_big(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) ((float*)b.p[0])[0] += b.v[3]; }
__global__ void k_write(float4* d, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = float4{1, 2, 3, 4}; }
template <typename F> void bench(const char* tag, hipStream_t st, F launch) {
  const int N = 400;
  hipGraph_t g; hipGraphExec_t x; hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) launch(i);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(x, st));
  CK(hipEventRecord(a, st));
  for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(x, st));
  CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("%-50s %.2f us per node\n", tag, ms * 1e3 / (5 * N));
}
int main() {
  float* p; CK(hipMalloc(&p, 256 << 20)); CK(hipMemset(p, 0, 256 << 20));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 147456));
  Big big = {}; big.p[0] = p;
  bench("trivial 256x256", st, [&](int) { k_triv<<<256, 256, 0, st>>>(p); });
  bench("trivial 256x512 threads", st, [&](int) { k_triv<<<256, 512, 0, st>>>(p); });
  bench("144KB dyn LDS 256x256", st, [&](int) { k_lds<<<256, 256, 147456, st>>>(p); });
  bench("144KB dyn LDS 256x512", st, [&](int) { k_lds<<<256, 512, 147456, st>>>(p); });
  bench("64KB dyn LDS 256x256", st, [&](int) { k_lds<<<256, 256, 65536, st>>>(p); });
  bench("big kernarg (216 B)", st, [&](int) { k_big<<<256, 256, 0, st>>>(big); });
  bench("write 5 MB", st, [&](int) { k_write<<<1280, 256, 0, st>>>((float4*)p, 327680); });
  bench("write 5 MB, rotating 32 buffers", st, [&](int i) { k_write<<<1280, 256, 0, st>>>((float4*)p + (size_t)(i % 32) * 327680, 327680); });
  bench("alternate trivial / 144KB LDS", st, [&](int i) { if (i & 1) k_lds<<<256, 512, 147456, st>>>(p); else k_triv<<<256, 256, 0, st>>>(p); });
  return 0;
}
