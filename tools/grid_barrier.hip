// What does one grid-wide barrier cost inside a kernel on MI355X (8 XCDs, L2s not coherent with each other), when only a small table has to
// cross it?  256 / 128 / 64 blocks: each writes 64 floats of "statistics" with write-through stores, arrives on an agent-scope counter
// (sense-reversing, bounded spin), then reads the 64 floats of every block of its "image" past the L2.  Compared with the same kernel without
// the barrier and with an empty kernel, as nodes of a HIP graph.  No release fence anywhere: a release at agent scope is an L2 write-back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty(float* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.f; }

template <bool BARRIER>
__global__ void __launch_bounds__(256) k_bar(float* table, unsigned* ctl, float* out, int* fail) {
  const int t = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
  __shared__ unsigned s0;
  if (t == 0) s0 = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sense at entry
  __syncthreads();
  if (t < 64) __hip_atomic_store(table + b * 64 + t, (float)(b + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
  if (BARRIER) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores have been acknowledged
    __syncthreads();
    if (t == 0) {
      const unsigned sense = s0;
      unsigned old = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)nb - 1) {
        __hip_atomic_store(ctl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(ctl + 1, sense ^ 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        int spins = 0;
        while (__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sense) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1 << 20)) { *fail = 1; break; }   // bounded: never hangs
        }
      }
    }
    __syncthreads();
  }
  // every thread sums one column over the blocks of its half ("image") past the L2
  const int half = b < nb / 2 ? 0 : 1, first = half * (nb / 2);
  float s = 0.f;
  if (t < 64) for (int j = 0; j < nb / 2; ++j) s += __hip_atomic_load(table + (first + j) * 64 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t < 64) out[b * 64 + t] = s;
}

int main() {
  float *table, *out; unsigned* ctl; int* fail;
  CK(hipMalloc(&table, 1 << 20)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&ctl, 64)); CK(hipMalloc(&fail, 4));
  CK(hipMemset(ctl, 0, 64)); CK(hipMemset(fail, 0, 4)); CK(hipMemset(table, 0, 1 << 20));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int nb : {256, 128, 64}) {
    for (int variant = 0; variant < 3; ++variant) {
      const int N = 200;
      hipGraph_t g; hipGraphExec_t x;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < N; ++i) {
        if (variant == 0) k_empty<<<nb, 256, 0, st>>>(out);
        else if (variant == 1) k_bar<false><<<nb, 256, 0, st>>>(table, ctl, out, fail);
        else k_bar<true><<<nb, 256, 0, st>>>(table, ctl, out, fail);
      }
      CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
      for (int w = 0; w < 2; ++w) CK(hipGraphLaunch(x, st));
      CK(hipEventRecord(a, st));
      for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(x, st));
      CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      int f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
      float o[64]; CK(hipMemcpy(o, out, 256, hipMemcpyDeviceToHost));
      // expected column 0 of half 0: sum_{j < nb/2} j
      printf("blocks %3d  %-28s %.2f us per node   (fail %d, check %s)\n", nb,
             variant == 0 ? "empty" : variant == 1 ? "table write + read, no barrier" : "table write, BARRIER, read", ms * 1e3 / (5 * N), f,
             variant == 2 ? (o[0] == (float)((nb / 2) * (nb / 2 - 1) / 2) ? "ok" : "WRONG") : "-");
      CK(hipGraphExecDestroy(x)); CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
