// fp32 GEMM with cuBLAS semantics -- the replacement of cublasSgemm_v2 / cublasSgemmBatched as the reference's
// linear_cublas / gemm_batch call them (ff/linear.py:8-61, :82-110; native/cublas/ops.py:22-53).  These are the
// reference's test-only fp32 paths (tests/linear.py:64-110), not the UNet's hot path: a plain LDS-tiled FMA kernel,
// exact fp32 accumulation in ascending k (no MFMA: the xf32 / bf16 matrix paths would not be an SGEMM).
//   C (m x n, column-major, ldc) = alpha * op(A) (m x k) * op(B) (k x n) + beta * C
//   op = 0 (N): the matrix is stored column-major as given;  op = 1 / 2 (T / C): its transpose is stored.
#include "common.h"
#include "../../include/tinyfusers_hip.h"

#define SG_T 64   // block tile (m and n)
#define SG_K 16   // k step

struct SgemmP {
  const float* A; const float* B; float* C;
  const float* const* Ab; const float* const* Bb; float* const* Cb;   // batched: arrays of device pointers (or NULL)
  int m, n, k, lda, ldb, ldc, ta, tb;
  float alpha, beta;
};

// element (i, j) of op(X): column-major storage with leading dimension ld
__device__ __forceinline__ float sg_at(const float* X, int ld, int t, int i, int j) { return t ? X[(long long)i * ld + j] : X[(long long)j * ld + i]; }

__global__ void __launch_bounds__(256) k_sgemm(const SgemmP p) {
  __shared__ float As[SG_K][SG_T + 1], Bs[SG_K][SG_T + 1];
  const float* A = p.Ab ? p.Ab[blockIdx.z] : p.A;
  const float* B = p.Bb ? p.Bb[blockIdx.z] : p.B;
  float* C = p.Cb ? p.Cb[blockIdx.z] : p.C;
  const int m0 = blockIdx.x * SG_T, n0 = blockIdx.y * SG_T;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 x 16 threads, 4 x 4 outputs each
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < p.k; k0 += SG_K) {
    for (int e = threadIdx.x; e < SG_K * SG_T; e += 256) {
      // consecutive threads walk the storage-contiguous index of each operand
      int kk, ii;
      if (p.ta) { kk = e % SG_K; ii = e / SG_K; } else { ii = e % SG_T; kk = e / SG_T; }
      int gm = m0 + ii, gk = k0 + kk;
      As[kk][ii] = (gm < p.m && gk < p.k) ? sg_at(A, p.lda, p.ta, gm, gk) : 0.f;
      int jj;
      if (p.tb) { jj = e % SG_T; kk = e / SG_T; } else { kk = e % SG_K; jj = e / SG_K; }
      int gn = n0 + jj; gk = k0 + kk;
      Bs[kk][jj] = (gn < p.n && gk < p.k) ? sg_at(B, p.ldb, p.tb, gk, gn) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < SG_K; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][tx + 16 * i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][ty + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int gm = m0 + tx + 16 * i, gn = n0 + ty + 16 * j;
      if (gm < p.m && gn < p.n) {
        float* c = C + (long long)gn * p.ldc + gm;
        *c = p.beta == 0.f ? p.alpha * acc[i][j] : p.alpha * acc[i][j] + p.beta * *c;
      }
    }
}

static int sgemm_check(const char* fn, int ta, int tb, int m, int n, int k, int lda, int ldb, int ldc) {
  TF_REQUIRE(ta >= 0 && ta <= 2 && tb >= 0 && tb <= 2, "%s: transa/transb must be 0 (N), 1 (T) or 2 (C)", fn);
  TF_REQUIRE(m >= 0 && n >= 0 && k >= 0, "%s: negative dimension", fn);
  TF_REQUIRE(lda >= (ta ? k : m) && lda >= 1, "%s: lda=%d too small", fn, lda);      // same rules as cuBLAS (status 7 there)
  TF_REQUIRE(ldb >= (tb ? n : k) && ldb >= 1, "%s: ldb=%d too small", fn, ldb);
  TF_REQUIRE(ldc >= m && ldc >= 1, "%s: ldc=%d too small", fn, ldc);
  return TF_OK;
}

extern "C" {

int tf_sgemm_f32(int transa, int transb, int m, int n, int k, float alpha, const void* A, int lda, const void* B, int ldb, float beta,
                 void* C, int ldc, tfStream_t s) {
  int rc = sgemm_check("tf_sgemm_f32", transa, transb, m, n, k, lda, ldb, ldc);
  if (rc) return rc;
  if (m == 0 || n == 0) return TF_OK;
  TF_REQUIRE(A && B && C, "tf_sgemm_f32: null matrix");
  SgemmP p = {(const float*)A, (const float*)B, (float*)C, nullptr, nullptr, nullptr, m, n, k, lda, ldb, ldc, transa != 0, transb != 0, alpha, beta};
  hipLaunchKernelGGL(k_sgemm, dim3((m + SG_T - 1) / SG_T, (n + SG_T - 1) / SG_T, 1), dim3(256), 0, tf_hs(s), p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

int tf_sgemm_batched_f32(int transa, int transb, int m, int n, int k, float alpha, const void* const* Aarray, int lda,
                         const void* const* Barray, int ldb, float beta, void* const* Carray, int ldc, int batch, tfStream_t s) {
  int rc = sgemm_check("tf_sgemm_batched_f32", transa, transb, m, n, k, lda, ldb, ldc);
  if (rc) return rc;
  TF_REQUIRE(batch >= 0 && batch <= 65535, "tf_sgemm_batched_f32: batch=%d out of range", batch);
  if (m == 0 || n == 0 || batch == 0) return TF_OK;
  TF_REQUIRE(Aarray && Barray && Carray, "tf_sgemm_batched_f32: null pointer array");
  SgemmP p = {nullptr, nullptr, nullptr, (const float* const*)Aarray, (const float* const*)Barray, (float* const*)Carray, m, n, k, lda, ldb, ldc,
              transa != 0, transb != 0, alpha, beta};
  hipLaunchKernelGGL(k_sgemm, dim3((m + SG_T - 1) / SG_T, (n + SG_T - 1) / SG_T, batch), dim3(256), 0, tf_hs(s), p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

}  // extern "C"
