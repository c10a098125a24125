// How long does one node of a HIP graph (or an eager back-to-back launch) take for a trivial kernel?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_triv(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_copy(float4* d, const float4* s, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = s[i]; }
int main() {
  float* p; CK(hipMalloc(&p, 64 << 20)); CK(hipMemset(p, 0, 64 << 20));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int variant = 0; variant < 4; ++variant) {
    int grid = variant == 0 ? 1 : variant == 1 ? 256 : variant == 2 ? 2048 : 1280;
    const int N = 400;
    hipGraph_t g; hipGraphExec_t x;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
      if (variant < 3) k_triv<<<grid, 256, 0, st>>>(p);
      else k_copy<<<grid, 256, 0, st>>>((float4*)p + (4 << 20) / 16 * 0 + 327680, (const float4*)p, 327680);   // 5.2 MB copy
    }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(x, st));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(x, st));
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("graph: %s grid %4d : %.2f us per node\n", variant < 3 ? "trivial" : "copy 5.2MB", grid, ms * 1e3 / (5 * N));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < N; ++i) { if (variant < 3) k_triv<<<grid, 256, 0, st>>>(p); else k_copy<<<grid, 256, 0, st>>>((float4*)p + 327680, (const float4*)p, 327680); }
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("eager: grid %4d : %.2f us per launch\n", grid, ms * 1e3 / N);
  }
  return 0;
}
