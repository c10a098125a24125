// Implicit-GEMM convolution / linear on MFMA, gfx950: HOST side (tile choice, autotuner, split-K reduce, the C-ABI entries) and the small
// kernels around the GEMMs (split-K reduce + GroupNorm, e4m3 packing, GEMV, LayerNorm weight fold).  The GEMM kernels live in
// gemm_{igemm,patch,igemm8,pp,c4}.h and are instantiated in gemm_k_*.hip, one translation unit per family so that they compile in
// parallel; gemm_common.h holds GemmP, the shared epilogue and the launcher declarations (its head comment describes the computation).
#include "gemm_common.h"
#include <stdlib.h>

// Split-K partial slabs are fp32 or -- round 4, GemmP::part16 -- fp16 (half the bytes of the seam: a slab is written once and read once, both
// through HBM / L2; the reducer accumulates in fp32 in split order either way).  PT = the slab's element type.
template <typename PT> __device__ __forceinline__ f4 part_load4(const PT* p);
template <> __device__ __forceinline__ f4 part_load4<float>(const float* p) { return *reinterpret_cast<const f4*>(p); }
template <> __device__ __forceinline__ f4 part_load4<half_t>(const half_t* p) {
  const h4 h = *reinterpret_cast<const h4*>(p);
  return (f4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
// The split partials of one element quad, summed in split order with NB independent loads in flight.  The loads are UNCONDITIONAL -- a slab index
// past the last one re-reads the last slab and its value is dropped -- because a load under `z < splitk` compiles to load / wait / branch one slab at a
// time (round 4: the reducers spent 8-16 serial L2 latencies per element that way, 14 us for a launch that moves 5 MB).  NB is the smallest of 2 / 4 /
// 8 / 16 that covers splitk in one batch where it can: no redundant loads for the common split counts.
template <typename PT, int NB, bool VS = true>
__device__ __forceinline__ f4 sum_partials_nb(const PT* __restrict__ partial, long long total, long long e0, int splitk) {
  f4 v = {0.f, 0.f, 0.f, 0.f};
  // (the slab stride as a per-lane value: with a uniform stride the compiler keeps NB 64-bit slab bases in SGPRs -- 26-41 of them spilled in the
  // fp16-slab instances, VERDICT r4 -- where one 64-bit VALU multiply-add per load does)
  // VS = false (k_splitk_reduce_gn_apply: 128-VGPR budget at 1024 threads, where the per-lane addresses spill VECTOR registers instead): uniform stride
  int zv_ = 0;
  if constexpr (VS) asm volatile("v_mov_b32 %0, 0" : "=v"(zv_));
  const long long stride_v = total + (long long)zv_;
  for (int z0 = 0; z0 < splitk; z0 += NB) {
    f4 u[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) u[i] = part_load4<PT>(partial + (long long)min(z0 + i, splitk - 1) * stride_v + e0);
#pragma unroll
    for (int i = 0; i < NB; ++i) v += z0 + i < splitk ? u[i] : (f4){0.f, 0.f, 0.f, 0.f};
  }
  return v;
}
template <typename PT, bool VS = true>
__device__ __forceinline__ f4 sum_partials(const PT* __restrict__ partial, long long total, long long e0, int splitk) {
  if (splitk <= 2) return sum_partials_nb<PT, 2, VS>(partial, total, e0, splitk);
  if (splitk <= 4) return sum_partials_nb<PT, 4, VS>(partial, total, e0, splitk);
  if (sizeof(PT) == 2 && splitk > 8) return sum_partials_nb<PT, 16, VS>(partial, total, e0, splitk);
  return sum_partials_nb<PT, 8, VS>(partial, total, e0, splitk);
}
// split-K reduce + epilogue: y[m,n] = sum_z partial[z,m,n] + bias + bias_nc + residual   (N % 4 == 0 fast path)
// BF: bias / bias_nc / residual / gamma / beta / y / z hold bfloat16 (containers as in gemm_common.h: e2f / f2e)
template <typename PT, bool BF = false>
__global__ void __launch_bounds__(256) k_splitk_reduce(half_t* __restrict__ y, const PT* __restrict__ partial, const half_t* __restrict__ bias,
                                                       const half_t* __restrict__ bias_nc, const half_t* __restrict__ residual, int M, int N,
                                                       int HoWo, int splitk, long long bnc_stride) {
  long long total = (long long)M * N;
  long long gs = (long long)gridDim.x * 256;
  if ((N & 3) == 0) {
    long long nv = total >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += gs) {
      long long e0 = i << 2;
      int m = (int)(e0 / N), n = (int)(e0 - (long long)m * N);
      f4 v = sum_partials<PT>(partial, total, e0, splitk);
      if (bias) { h4 b = *reinterpret_cast<const h4*>(bias + n); for (int e = 0; e < 4; ++e) v[e] += e2f<BF>(b[e]); }
      if (bias_nc) { h4 b = *reinterpret_cast<const h4*>(bias_nc + (long long)(m / HoWo) * bnc_stride + n); for (int e = 0; e < 4; ++e) v[e] += e2f<BF>(b[e]); }
      if (residual) { h4 b = *reinterpret_cast<const h4*>(residual + e0); for (int e = 0; e < 4; ++e) v[e] += e2f<BF>(b[e]); }
      h4 o;
      for (int e = 0; e < 4; ++e) o[e] = f2e<BF>(v[e]);
      *reinterpret_cast<h4*>(y + e0) = o;
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += gs) {
      int m = (int)(i / N), n = (int)(i - (long long)m * N);
      float v = 0.f;
      for (int z = 0; z < splitk; ++z) v += (float)partial[(long long)z * total + i];
      if (bias) v += e2f<BF>(bias[n]);
      if (bias_nc) v += e2f<BF>(bias_nc[(long long)(m / HoWo) * bnc_stride + n]);
      if (residual) v += e2f<BF>(residual[i]);
      y[i] = f2e<BF>(v);
    }
  }
}

// split-K reduce + epilogue + GroupNorm statistics of the output (tf_conv2d_fused_f16 on a split-K shape): a block owns R
// whole output rows (R = HoWo / chunks); thread t owns column quad t % nq and the rows r = t / nq (mod RL), so the
// per-channel sums stay in its registers; row lanes and channels -> groups meet through LDS in a fixed order.
// The split partials of an element are fetched 8 at a time (independent loads) and added in split order.
template <typename PT, bool BF = false>
__global__ void __launch_bounds__(1024) k_splitk_reduce_gn(half_t* __restrict__ y, const PT* __restrict__ partial, const half_t* __restrict__ bias,
                                                           const half_t* __restrict__ bias_nc, const half_t* __restrict__ residual, int M, int N,
                                                           int HoWo, int splitk, long long bnc_stride, float* __restrict__ gn_part, int G, int cpg,
                                                           int chunks, int R, int RL) {
  extern __shared__ float chan[];                        // [RL][N][2]
  const long long total = (long long)M * N;
  const int m_first = blockIdx.x * R;
  const int nq = N >> 2;
  const int rl = threadIdx.x / nq, q0 = threadIdx.x - rl * nq;
  if (rl < RL) {
    const int n = q0 << 2;
    f4 cs = {0.f, 0.f, 0.f, 0.f}, cq = {0.f, 0.f, 0.f, 0.f};
    f4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) { h4 b = *reinterpret_cast<const h4*>(bias + n); for (int e = 0; e < 4; ++e) bv[e] = e2f<BF>(b[e]); }
    for (int r = rl; r < R; r += RL) {
      const int m = m_first + r;
      const long long e0 = (long long)m * N + n;
      h4 bnc = {0, 0, 0, 0}, res = {0, 0, 0, 0};
      if (bias_nc) bnc = *reinterpret_cast<const h4*>(bias_nc + (long long)(m / HoWo) * bnc_stride + n);
      if (residual) res = *reinterpret_cast<const h4*>(residual + e0);
      f4 v = sum_partials<PT>(partial, total, e0, splitk);
      v += bv;
      for (int e = 0; e < 4; ++e) v[e] += e2f<BF>(bnc[e]);
      for (int e = 0; e < 4; ++e) v[e] += e2f<BF>(res[e]);
      h4 o;
      for (int e = 0; e < 4; ++e) { o[e] = f2e<BF>(v[e]); float f = e2f<BF>(o[e]); cs[e] += f; cq[e] += f * f; }
      *reinterpret_cast<h4*>(y + e0) = o;
    }
    float* ch = chan + (long long)rl * N * 2;
    for (int e = 0; e < 4; ++e) { ch[2 * (n + e)] = cs[e]; ch[2 * (n + e) + 1] = cq[e]; }
  }
  __syncthreads();
  // one wave per group (round-robin): lane = channel of the group (cpg <= 64), RL row-lane reads, shuffle tree
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int img = m_first / HoWo, slot = (m_first - img * HoWo) / R;
  for (int g = wv; g < G; g += nw) {
    float S = 0.f, Q = 0.f;
    if (lane < cpg) {
      const int c = g * cpg + lane;
      for (int l = 0; l < RL; ++l) { S += chan[((long long)l * N + c) * 2]; Q += chan[((long long)l * N + c) * 2 + 1]; }
    }
    S = wave_sum(S); Q = wave_sum(Q);
    if (lane == 0) {
      float* dst = gn_part + ((long long)(img * chunks + slot) * G + g) * 2;
      dst[0] = S; dst[1] = Q;
    }
  }
}

// split-K reduce + epilogue + GroupNorm of the output, statistics AND apply, in one launch (conv -> GroupNorm -> SiLU of
// vision/resnet.py:17-22 behind a split-K conv): a block owns ALL rows of one image for `gpb` whole groups (CW = gpb * cpg channels),
// so the statistics are complete inside the block and no second launch has to wait for them.  Thread t holds the column quad
// t % CV of rows t / CV + k * RPS (k < RGA_MAXR) in registers: partials summed in split order (8 loads in flight), + bias + bias_nc +
// residual, rounded to fp16 (y, optional), per-channel sums in registers -> LDS -> fixed-order fold -> (mean, rstd) -> z = silu?(y a + b).
// Also leaves the (sum, sum of squares) of every group as a one-chunk partial table, so y.gn stays available to later consumers.
#define RGA_MAXR 8
template <typename PT, bool BF = false>
__global__ void __launch_bounds__(1024) k_splitk_reduce_gn_apply(half_t* __restrict__ y, half_t* __restrict__ z, const PT* __restrict__ partial,
                                                                 const half_t* __restrict__ bias, const half_t* __restrict__ bias_nc,
                                                                 const half_t* __restrict__ residual, int M, int N, int HoWo, int splitk, long long bnc_stride,
                                                                 float* __restrict__ gn_part, int G, int cpg, int gpb, const half_t* __restrict__ gamma,
                                                                 const half_t* __restrict__ beta, float eps, int do_silu, int RPS, int CV) {
  extern __shared__ float sm[];                          // [RPS][CW][2], then [parts][CW][2] behind it, then [gpb][2]
  const int nb = G / gpb;
  const int img = blockIdx.x / nb, gs = blockIdx.x - img * nb;
  const int CW = gpb * cpg, c0 = gs * CW;
  const int t = threadIdx.x;
  const int rl = t / CV, v = t - rl * CV;
  const bool act = rl < RPS;
  const int n = c0 + v * 4;
  const long long total = (long long)M * N;
  f4 bv = {0.f, 0.f, 0.f, 0.f};
  h4 bnc = {0, 0, 0, 0};
  h4 gm = {f2e<BF>(1.f), f2e<BF>(1.f), f2e<BF>(1.f), f2e<BF>(1.f)}, bt = {0, 0, 0, 0};               // (fetched here, under the partials' latency: behind the block barriers they would be one more serial L2 round trip)
  if (act) {
    if (bias) { h4 b = *reinterpret_cast<const h4*>(bias + n); for (int e = 0; e < 4; ++e) bv[e] = e2f<BF>(b[e]); }
    if (bias_nc) bnc = *reinterpret_cast<const h4*>(bias_nc + (long long)img * bnc_stride + n);
    if (gamma) { gm = *reinterpret_cast<const h4*>(gamma + n); bt = *reinterpret_cast<const h4*>(beta + n); }
  }
  h4 out[RGA_MAXR];
  f4 cs = {0.f, 0.f, 0.f, 0.f}, cq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < RGA_MAXR; ++k) {
    const int r = rl + k * RPS;
    out[k] = (h4){0, 0, 0, 0};
    if (act && r < HoWo) {
      const long long e0 = ((long long)img * HoWo + r) * N + n;
      h4 res = {0, 0, 0, 0};
      if (residual) res = *reinterpret_cast<const h4*>(residual + e0);
      f4 acc = sum_partials<PT, false>(partial, total, e0, splitk);
      acc += bv;
      for (int e = 0; e < 4; ++e) acc[e] += e2f<BF>(bnc[e]);
      for (int e = 0; e < 4; ++e) acc[e] += e2f<BF>(res[e]);
      h4 o;
      for (int e = 0; e < 4; ++e) { o[e] = f2e<BF>(acc[e]); float f = e2f<BF>(o[e]); cs[e] += f; cq[e] += f * f; }
      out[k] = o;
      if (y) *reinterpret_cast<h4*>(y + e0) = o;
    }
  }
  f2* col = reinterpret_cast<f2*>(sm);                   // [RPS][CW]
  if (act) for (int e = 0; e < 4; ++e) col[rl * CW + v * 4 + e] = (f2){cs[e], cq[e]};
  __syncthreads();
  // fold 1: thread (part, c) sums the row lanes part, part + parts, ... of channel c, in order
  const int parts = 1024 / CW;
  f2* p1 = col + RPS * CW;                               // [parts][CW]
  {
    const int part = t / CW, c = t - part * CW;
    if (part < parts) {
      float S = 0.f, Q = 0.f;
      for (int l = part; l < RPS; l += parts) { f2 q = col[l * CW + c]; S += q[0]; Q += q[1]; }
      p1[part * CW + c] = (f2){S, Q};
    }
  }
  __syncthreads();
  // fold 2: one wave per group: lanes stride over the (part, channel of the group) pairs in a fixed order, fp64, shuffle tree
  float* st = reinterpret_cast<float*>(p1 + parts * CW);  // [gpb][2]: mean, rstd
  {
    const int wv = t >> 6, lane = t & 63;
    if (wv < gpb) {
      double S = 0.0, Q = 0.0;
      const int npairs = parts * cpg;
      for (int q = lane; q < npairs; q += 64) {
        int part = q / cpg, c = wv * cpg + (q - part * cpg);
        f2 u = p1[part * CW + c];
        S += (double)u[0]; Q += (double)u[1];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { S += __shfl_xor(S, o, 64); Q += __shfl_xor(Q, o, 64); }
      if (lane == 0) {
        const double cnt = (double)HoWo * cpg;
        double mean = S / cnt, var = Q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        st[2 * wv] = (float)mean; st[2 * wv + 1] = (float)(1.0 / sqrt(var + (double)eps));
        if (gn_part) { float* d = gn_part + ((long long)img * G + gs * gpb + wv) * 2; d[0] = (float)S; d[1] = (float)Q; }
      }
    }
  }
  __syncthreads();
  if (!act) return;
  float a[4], b[4];
  {
    for (int e = 0; e < 4; ++e) {
      const int g = (v * 4 + e) / cpg;
      a[e] = st[2 * g + 1] * e2f<BF>(gm[e]);
      b[e] = e2f<BF>(bt[e]) - st[2 * g] * a[e];
    }
  }
#pragma unroll
  for (int k = 0; k < RGA_MAXR; ++k) {
    const int r = rl + k * RPS;
    if (r < HoWo) {
      h4 o;
      for (int e = 0; e < 4; ++e) { float f = e2f<BF>(out[k][e]) * a[e] + b[e]; o[e] = f2e<BF>(do_silu ? silu_f(f) : f); }
      *reinterpret_cast<h4*>(z + ((long long)img * HoWo + r) * N + n) = o;
    }
  }
}
// geometry of that launch for (HoWo, N, G): groups per block, quads per row, rows per sweep, LDS bytes; false = not eligible
static bool rga_geometry(int HoWo, int N, int G, int* gpb, int* CV, int* RPS, size_t* lds) {
  if (G < 1 || N % G || N % 4) return false;
  const int cpg = N / G;
  int g = 1;
  while (g <= G && ((g * cpg) % 4 != 0 || G % g != 0)) ++g;
  if (g > G || g > 16) return false;                     // one wave per group in fold 2
  const int CW = g * cpg;
  if (CW > 256) return false;
  int cv = CW / 4, rps = 1024 / cv;
  if (rps > HoWo) rps = HoWo;
  if ((long long)rps * RGA_MAXR < HoWo) return false;
  const int parts = 1024 / CW;
  size_t bytes = ((size_t)rps * CW + (size_t)parts * CW) * 8 + (size_t)g * 8;
  if (bytes > 160 * 1024) return false;
  *gpb = g; *CV = cv; *RPS = rps; *lds = bytes;
  return true;
}

// ---- fp8 (OCP e4m3) packing for the config-5 path ---------------------------------------------------------------------------
// activations: y8 = e4m3(x * scale), saturating (8 elements per thread, 16-byte loads / 8-byte stores)
__global__ void __launch_bounds__(256) k_quantize_fp8(unsigned char* __restrict__ y, const half_t* __restrict__ x, float scale, long long n8) {
  long long gs = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += gs) {
    h8 v = *reinterpret_cast<const h8*>(x + i * 8);
    f4 a, b;
    for (int e = 0; e < 4; ++e) { a[e] = (float)v[e] * scale; b[e] = (float)v[4 + e] * scale; }
    *reinterpret_cast<uint2*>(y + i * 8) = pack8_fp8(a, b);
  }
}
// block-scaled form (common.h: mx_quant8): rows x C fp16 -> rows x C codes + rows x C/32 E8M0 bytes behind them; a thread = 8 channels, 4 lanes = a block
__global__ void __launch_bounds__(256) k_quantize_mx8(unsigned char* __restrict__ y, const half_t* __restrict__ x, long long rows, int C) {
  const long long n8 = rows * (C >> 3), gs = (long long)gridDim.x * 256;
  const int cv8 = C >> 3;
  for (long long i0 = (long long)blockIdx.x * 256; i0 < n8; i0 += gs) {       // (block-uniform loop bound: every lane reaches the shuffles)
    const long long i = i0 + threadIdx.x;
    const bool live = i < n8;
    f4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      h8 v = *reinterpret_cast<const h8*>(x + i * 8);
      for (int e = 0; e < 4; ++e) { a[e] = (float)v[e]; b[e] = (float)v[4 + e]; }
    }
    unsigned sb;
    const uint2 code = mx_quant8(a, b, sb);
    if (live) {
      *reinterpret_cast<uint2*>(y + i * 8) = code;
      if ((i & 3) == 0) { const long long row = i / cv8; y[rows * C + row * (C >> 5) + ((i - row * cv8) >> 2)] = (unsigned char)sb; }
    }
  }
}
// weights: one wave per output row n: scale[n] = max|w[n, :]| / 448 (1 for an all-zero row), w8[n, k] = e4m3(w[n, k] / scale[n])
__global__ void __launch_bounds__(256) k_pack_weight_fp8(unsigned char* __restrict__ w8, float* __restrict__ scale, const half_t* __restrict__ w, int N, int K) {
  int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  int n = blockIdx.x * 4 + wv;
  if (n >= N) return;
  const half_t* wr = w + (long long)n * K;
  float m = 0.f;
  for (int k = l * 8; k < K; k += 512) { h8 v = *reinterpret_cast<const h8*>(wr + k); for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)v[j])); }
  m = wave_max(m);
  const float sc = m > 0.f ? __fdiv_rn(m, 448.0f) : 1.0f;
  if (l == 0) scale[n] = sc;
  for (int k = l * 8; k < K; k += 512) {
    h8 v = *reinterpret_cast<const h8*>(wr + k);
    f4 a, b;
    // a correctly rounded quotient (not w * (1 / scale)): a value on an e4m3 code boundary must round the way the definition says
    for (int e = 0; e < 4; ++e) { a[e] = __fdiv_rn((float)v[e], sc); b[e] = __fdiv_rn((float)v[4 + e], sc); }
    *reinterpret_cast<uint2*>(w8 + (long long)n * K + k) = pack8_fp8(a, b);
  }
}

// ---- weight-streaming GEMV for M <= 8 (time-embedding MLP, ResBlock emb_layers): one wave per output row
template <bool BF = false>
__global__ void __launch_bounds__(256) k_gemv(half_t* __restrict__ y, const half_t* __restrict__ x, const half_t* __restrict__ w,
                                              const half_t* __restrict__ bias, int M, int N, int K, int silu_in) {
  int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  int n = blockIdx.x * 4 + wv;
  if (n >= N) return;
  const half_t* wr = w + (long long)n * K;
  float acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = 0.f;
  for (int k = l * 8; k < K; k += 512) {
    h8 wv8 = *reinterpret_cast<const h8*>(wr + k);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < M) {
        h8 xv = *reinterpret_cast<const h8*>(x + (long long)m * K + k);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float xf = e2f<BF>(xv[j]);
          if (silu_in) xf = silu_f(xf);
          acc[m] += xf * e2f<BF>(wv8[j]);
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    if (m < M) {
      float v = wave_sum(acc[m]);
      if (l == 0) y[(long long)m * N + n] = f2e<BF>(v + (bias ? e2f<BF>(bias[n]) : 0.f));
    }
  }
}

// LayerNorm fold of a Linear weight (one wave per output row n):
//   w'[n,k] = fp16(w[n,k] * gamma[k]);  colsum[n] = sum_k float(w'[n,k]);  bias'[n] = sum_k beta[k] * w[n,k] + bias[n]
template <bool BF = false>
__global__ void __launch_bounds__(256) k_ln_fold(half_t* __restrict__ wo, half_t* __restrict__ bo, float* __restrict__ colsum, const half_t* __restrict__ w,
                                                 const half_t* __restrict__ bias, const half_t* __restrict__ gamma, const half_t* __restrict__ beta, int N, int K) {
  int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
  int n = blockIdx.x * 4 + wv;
  if (n >= N) return;
  float cs = 0.f, bs = 0.f;
  for (int k = l * 8; k < K; k += 512) {
    h8 v = *reinterpret_cast<const h8*>(w + (long long)n * K + k), g = *reinterpret_cast<const h8*>(gamma + k), b = *reinterpret_cast<const h8*>(beta + k), o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = f2e<BF>(e2f<BF>(v[j]) * e2f<BF>(g[j]));
      cs += e2f<BF>(o[j]);
      bs += e2f<BF>(b[j]) * e2f<BF>(v[j]);
    }
    *reinterpret_cast<h8*>(wo + (long long)n * K + k) = o;
  }
  cs = wave_sum(cs); bs = wave_sum(bs);
  if (l == 0) { colsum[n] = cs; bo[n] = f2e<BF>(bs + (bias ? e2f<BF>(bias[n]) : 0.f)); }
}

// ------------------------------------------------------------------------------------------------
// per-launch event profiling of this kernel family (bench.py roofline leg)
#define g_prof g_tf_prof      /* the one switch of every family (common.h) */
static double g_prof_ms = 0.0, g_prof_ms_full = 0.0, g_prof_flops = 0.0;
static long long g_prof_launches = 0;
struct ProfRec { hipEvent_t a, b, c; bool has_reduce; double flops, reduce_bytes; int M, N, K, taps, bm, bn, splitk, variant; };   // a .. b: the GEMM kernel alone; a .. c: with the split-K reduce that finishes it
#include <map>
#include <array>
static std::map<std::array<int, 8>, std::pair<long long, double>> g_prof_shapes;
static std::vector<ProfRec> g_prof_pending;

static int g_dbg = 0, g_force_wide = -1, g_force_order = -1;
static int g_part16 = 1;                                  // split-K partial slabs in fp16 (tf_gemm_splitk_partials: 16 / 32)
struct TileCfg { int bm, bn, splitk; };

// Cost model (microseconds) calibrated on MI355X with tools/gemm_bench.py: a K tile costs the larger of its LDS-DMA
// ingest time ((bm+bn)*128 B at ~90 GB/s per CU) and its MFMA time (~7 TFLOP/s per CU sustained), one block per CU per
// wave of blocks; split-K adds a reduce launch and an fp32 round trip of the output.
static TileCfg choose_tiles(int M, int N, int K, int act, bool allow_split) {
  static const int cand[][2] = {{128, 160}, {64, 160}, {128, 128}, {64, 128}, {128, 64}, {64, 64}};
  const int ncand = 6;
  int ktiles = (K + 63) / 64;
  TileCfg best = {64, 64, 1};
  double best_t = 1e30;
  for (int ci = 0; ci < ncand; ++ci) {
    int bm = cand[ci][0], bn = cand[ci][1];
    if (act == 1 && (bn % 64) != 0) continue;           // GEGLU pairs 16-row blocks inside a wave tile
    int ntm = (M + bm - 1) / bm, ntn = (N + bn - 1) / bn;
    double tiles = (double)ntm * ntn;
    double t_ing = (bm + bn) * 128.0 / 90e3, t_mfma = (double)bm * bn * 128.0 / 7.0e6;
    double t_tile = (t_ing > t_mfma ? t_ing : t_mfma) + 0.05;
    int max_split = (allow_split && act == 0) ? 32 : 1;
    for (int sk = 1; sk <= max_split; sk *= 2) {
      if (sk > 1 && ktiles / sk < 8) break;
      double blocks = tiles * sk;
      double waves = ceil(blocks / 256.0);
      double t = 3.0 + waves * ((ktiles + sk - 1) / sk) * t_tile;
      if (sk > 1) t += 4.0 + (double)M * N * 4.0 * (sk + 1) / 3.0e6;
      if (t < best_t) { best_t = t; best = {bm, bn, sk}; }
    }
  }
  return best;
}

// GroupNorm statistics from the producing conv: limits shared by the host entry, the tuner and the launches
#define TF_GN_MAX_CHUNKS 192    // (192: a 128-wide tile on 96 x 96 outputs emits 2 pieces x 96 half-tile chunks per image: the block-scaled patch kernel at config 5's first level)
static int gn_reduce_chunks(int HoWo) { int R = (HoWo + TF_GN_MAX_CHUNKS - 1) / TF_GN_MAX_CHUNKS; while (HoWo % R) ++R; return HoWo / R; }
static int gn_pieces(const GemmP& p, int bn) { return bn % p.gn_cpg == 0 ? 1 : 2; }   // chunks per m-tile (igemm_gn_stats)
static int gn_chunks_for(const GemmP& p, TileCfg c, int splitk) {
  return splitk > 1 ? gn_reduce_chunks(p.HoWo) : gn_pieces(p, c.bn) * (p.HoWo / c.bm);
}
static bool gn_tile_ok(const GemmP& p, int bm, int bn) { return p.HoWo % bm == 0 && gn_pieces(p, bn) * (p.HoWo / bm) <= TF_GN_MAX_CHUNKS; }

// k_igemm_patch: eligibility + geometry for a (bm, bn) tile.  3x3 / stride 1 / pad 1, no up-sampling, every channel count a
// multiple of 64, W a power of two that divides bm, m-tiles inside one image, and an LDS budget that leaves >= 3 ring slots.
static bool patch_setup(GemmP& p, int bm, int bn) {
  if (p.act || p.ln_colsum || p.S != 3 || p.Kc != 9 * p.C || p.stride != 1 || p.pad != 1 || p.ups) return false;
  if ((p.C1 % 64) || (p.C2 % 64) || (p.C3 % 64) || (p.C4 % 64) || p.H != p.Ho || p.W != p.Wo) return false;
  if (!((bm == 64 || bm == 128) && (bn == 128 || bn == 160))) return false;
  if ((p.W & (p.W - 1)) || p.W < 8 || p.W > bm || p.HoWo % bm) return false;
  int l2 = 0;
  while ((1 << l2) < p.W) ++l2;
  const int ppix = (bm / p.W + 2) * (p.W + 2), ppc = (ppix + 7) / 8;
  if ((ppc + 3) / 4 > TF_PATCH_PPW) return false;
  const int stage = bn * 128 + ((p.C3 + p.C4) ? bm * 128 : 0);
  int ns = (163840 - 2 * ppc * 1024 - gi_table_bytes(p)) / stage;
  if (ns > 5) ns = 5;                                    // patch pieces ride from tap 4 on: needs ns - 1 <= 4 (k_igemm_patch TAP0)
  if (ns < 3) return false;
  p.pt_ppc = ppc; p.pt_ppix = ppix; p.pt_stage = stage; p.pt_ns = ns; p.pt_log2w = l2;
  p.gi_off = 2 * ppc * 1024 + ns * stage;
  return true;
}
// k_igemm_pp (variant 4): 256 x BN tiles, every channel count on the 64 grid, no LayerNorm fold, no input GroupNorm, fp16 only
static bool pp_ok(const GemmP& p, int bn, int bm = 256) {
  if (bn != 128 && bn != 160 && bn != 256) return false;
  if (bm != 256 && !(bm == 192 && bn != 256)) return false;
  if ((p.bf16 && p.fp8) || p.gi_part || gemm_generic(p)) return false;
  if (p.ln_colsum && (p.fp8 || p.S != 1 || p.stride != 1 || p.ups)) return false;   // the LayerNorm fold: linears, fp16
  if (p.fp8) {
    // the e4m3 form: block-scaled activations only, the lean addressing only (stride 1, no up-sampling), a whole K tile's fragments in
    // registers (three-slot ring: no 256-wide tile), and room for the scale table behind the ring (not 256 x 160 with half-tile slabs)
    if (!p.mx || bn == 256 || p.stride != 1 || p.ups || p.S * p.S > 31 || p.C3 || p.C4) return false;
    const bool h2 = (p.C1 % 128) || (p.C2 % 128);
    if (bm == 256 && bn == 160 && h2) return false;
    if (p.out8 && !(p.act == 1 && bn == 128)) return false;   // a block-scaled output: the GEGLU epilogue of the 128-wide tile
  }
  if (p.bias_nc && p.HoWo < bm) return false;            // the epilogue's time-embedding table holds two images per tile
  return p.act != 1 || bn % 64 == 0;                     // GEGLU pairs 16-row value | gate blocks inside a wave tile
}
static int g_pp_np = 0;                                    // test / tuning hook: 0 = default phases per K tile, 2 = one phase per k-step where the tile has both forms
// rows of a tile as the GroupNorm-statistics code sees them: the ping-pong kernel's epilogue works in 128-row sub-blocks
static int stats_bm(int bm, int variant) { return (variant == 4 || variant == 6) ? bm / 2 : bm; }
// k_igemm_pp3 (variant 6): the PATCH form of the ping-pong kernel -- 3x3 / stride 1 / pad 1 convolutions of fp16 operands, every channel count on
// the 64 grid, a 192-row tile that is a whole number of image rows inside one image (W | 192, 192 | H W: the 96 / 48 / 24-pixel levels of
// BASELINE config 5, the OUTPUT row length a template parameter; nearest-2x up-sampling folds into the patch gather), no split-K; fp16, or block-scaled
// e4m3 on the 128-channel grid; two patch buffers + three weight slots in LDS
static bool pp3_setup(GemmP& p, int bm, int bn) {
  if (bm != 192 || bn != ((p.Wo == 96 && !p.fp8) ? 160 : 128)) return false;                          // the instantiated (output row length, tile width) pairs
  if ((p.bf16 && p.fp8) || p.gi_part || p.ln_colsum || p.act || p.out8 || p.out32 || gemm_generic(p)) return false;
  // e4m3: block-scaled, 128-channel slabs; a channel count on the 64 grid (one source tensor, its last slab half full) has instances for 96 / 48-pixel rows
  if (p.fp8 && (!p.mx || ((p.C1 % 128) && (p.C2 || p.Wo == 24)) || (p.C2 % 128))) return false;
  if (p.S != 3 || p.Kc != 9 * p.C || p.K != p.Kc + p.C3 + p.C4 || p.stride != 1 || p.pad != 1) return false;
  if ((p.C3 || p.C4) && (p.fp8 || p.ups || (p.C3 % 64) || (p.C4 % 64))) return false;               // the folded 1x1 skip projection: fp16, its sources at output resolution
  if ((p.C1 % 64) || (p.C2 % 64) || (p.H << p.ups) != p.Ho || (p.W << p.ups) != p.Wo) return false;   // (nearest-2x up-sampling folds into the patch gather)
  if ((p.Wo != 96 && p.Wo != 48 && p.Wo != 24) || (p.HoWo % 192) || (p.M % p.HoWo)) return false;   // the instantiated row lengths; a tile = whole rows of one image
  return true;
}
// k_gemm_c4 (variant 5): the persistent short-K kernel -- linears / 1x1 stride-1 convolutions of fp16 operands whose channel counts sit on
// the 64 grid, one launch (no split-K), no statistics, no time-embedding bias; bias, residual, GEGLU and the LayerNorm fold ride along
static bool c4_ok(const GemmP& p) {
  if (p.fp8 || p.gi_part || p.gn_part || p.bias_nc || p.out32 || p.out8 || p.on_z) return false;
  if (p.S != 1 || p.stride != 1 || p.pad != 0 || p.ups || p.C3 || p.C4 || p.K != p.Kc) return false;
  if ((p.C1 % 64) || (p.C2 % 64) || (p.N % 8) || p.M < 1) return false;
  return p.act == 0 || (p.act == 1 && p.N % 64 == 0);
}
// k_gemm_ar (variant 8): the activation-resident short-K kernel -- k_gemm_c4's launches whose K is 4 or 5 whole K tiles (256 / 320: the 128-row panel stays in LDS), no residual
static bool ar_ok(const GemmP& p) { return c4_ok(p) && (p.K == 256 || p.K == 320) && !p.residual; }   // (its loader waves store the outputs: a residual would be a second load stream in their instruction budget -- those launches stay on k_gemm_c4)
static const int kTiles8[][2] = {{128, 128}, {64, 128}, {128, 64}, {256, 64}, {64, 64}};
static const int kNumTiles8 = 5;

// GroupNorm of the input inside the launch (gi): which (tile, variant) can carry it.  3x3 / stride 1 / pad 1: the PATCH kernel only
// (a piece is normalised once for its nine taps); 1x1: the tap-by-tap kernel (k = channel), any ring variant; every channel count on
// the 64 grid, m-tiles inside one image (one statistics table per block), and room in LDS for the table.
static bool gi_tile_ok(const GemmP& p, int bm, int bn, int variant) {
  if (!p.gi_part) return true;
  if (gemm_generic(p) || p.act || p.ln_colsum || p.HoWo % bm) return false;
  if (p.S == 3) { GemmP probe = p; return variant == 2 && patch_setup(probe, bm, bn); }
  if (p.S != 1 || p.Kc != p.C || p.stride != 1 || p.pad != 0 || p.ups) return false;
  if (variant == 2 || (bm == 128 && bn == 160)) return false;
  return ((igemm_lds_bytes(bm, bn, variant == 1) + 15) & ~15) + gi_table_bytes(p) <= 163840;
}
static bool gi_any_ok(const GemmP& p) {
  static const int cand[][2] = {{128, 160}, {64, 160}, {128, 128}, {64, 128}, {128, 64}, {64, 64}};
  for (int ci = 0; ci < 6; ++ci)
    for (int v = 0; v < 4; ++v)
      if (gi_tile_ok(p, cand[ci][0], cand[ci][1], v)) return true;
  return false;
}

// K tiles of a launch: 64 elements, except the e4m3 ping-pong kernel's 128 (128 BYTES of a row either way).  Everything that reasons about
// split-K -- the effective split count, whether a reduce launch follows, the tuner's "at least 4 K tiles per split" -- goes through this
static int ktiles_for(const GemmP& p, int variant) { return ((variant == 4 || variant == 6) && p.fp8) ? (p.K + 127) / 128 : (p.K + 63) / 64; }
// the split count a launch really runs with (launch_one rounds the requested one to whole K tiles)
static int eff_splitk(const GemmP& p, int variant, int splitk) {
  const int kt = ktiles_for(p, variant), kps = (kt + splitk - 1) / splitk;
  return (kt + kps - 1) / kps;
}
// one fully specified launch (tile, split-K, ring variant) of the kernel family (+ the split-K reduce)
// variant: 0 deep ring, 1 WIDE (two blocks per CU), 2 PATCH (k_igemm_patch; falls back to 0 when the shape is not eligible),
// 3 ALL8 (deep ring, the consumer waves issue part of the weight pieces; falls back to 0 for channel counts off the 64 grid)
static hipEvent_t g_prof_end = nullptr;   // profiling pass only: recorded right behind the GEMM kernel, in front of its split-K reduce
static int launch_one(GemmP p, TileCfg c, int variant, int order, void* workspace, hipStream_t st) {
  int rc = 0;
  const bool wide = variant == 1, all8 = variant == 3;
  if (p.gi_part && !gi_tile_ok(p, c.bm, c.bn, variant)) {
    tf_set_error("run_gemm: tile %dx%d variant %d cannot carry the input GroupNorm", c.bm, c.bn, variant);
    return TF_E_UNSUPPORTED;
  }
  p.order = order;
  p.ktiles = ktiles_for(p, variant);
  p.ktiles_per_split = (p.ktiles + c.splitk - 1) / c.splitk;
  p.splitk = (p.ktiles + p.ktiles_per_split - 1) / p.ktiles_per_split;
  p.partial = (float*)workspace;
  p.part16 = (g_part16 && !p.bf16 && p.splitk > 1 && (p.N & 7) == 0) ? 1 : 0;     // 16-byte rows segments of halves; other widths -- and the bfloat16 launches, whose partials may leave fp16's range -- keep fp32 slabs
  p.ntm = (p.M + c.bm - 1) / c.bm;
  p.ntn = (p.N + c.bn - 1) / c.bn;
  float* gn_part = p.gn_part;
  if (gn_part) {
    // chunk geometry of the statistics partials: in-kernel (2 pieces per m-tile) or in the split-K reduce (row stripes)
    if (p.splitk > 1) { p.gn_chunks = gn_reduce_chunks(p.HoWo); p.gn_part = nullptr; }
    else p.gn_chunks = gn_pieces(p, c.bn) * (p.HoWo / stats_bm(c.bm, variant));
  }
  const bool bf = p.bf16 != 0;
  if (bf && p.fp8) { tf_set_error("run_gemm: e4m3 operands with bfloat16 outputs: no such kernel"); return TF_E_UNSUPPORTED; }
  if (p.mx && variant != 4 && variant != 6) { tf_set_error("run_gemm: block-scaled e4m3 operands run on the ping-pong kernel only (variant %d, tile %dx%d)", variant, c.bm, c.bn); return TF_E_UNSUPPORTED; }
  else if (p.fp8 && variant != 4 && variant != 6) rc = tfk_launch_igemm8(p, st, c.bm, c.bn);
  else if (variant == 4) {
    if (!pp_ok(p, c.bn, c.bm)) { tf_set_error("run_gemm: the ping-pong kernel cannot run tile %dx%d of this launch", c.bm, c.bn); return TF_E_UNSUPPORTED; }
    rc = p.fp8 ? tfk_launch_pp8(p, st, c.bm, c.bn) : bf ? tfk_launch_pp16_bf16(p, st, c.bm, c.bn, g_pp_np) : tfk_launch_pp16(p, st, c.bm, c.bn, g_pp_np);
  }
  else if (variant == 5) {
    if (!c4_ok(p) || c.bm != 128 || c.bn != 128 || p.splitk != 1) { tf_set_error("run_gemm: the persistent short-K kernel cannot run this launch (tile %dx%d, split %d)", c.bm, c.bn, p.splitk); return TF_E_UNSUPPORTED; }
    rc = bf ? tfk_launch_c4_bf16(p, st) : tfk_launch_c4(p, st);
  }
  else if (variant == 7) {
    if (!c4_ok(p) || c.bm != 256 || c.bn != 128 || p.splitk != 1) { tf_set_error("run_gemm: the 256-row persistent short-K kernel cannot run this launch (tile %dx%d, split %d)", c.bm, c.bn, p.splitk); return TF_E_UNSUPPORTED; }
    rc = bf ? tfk_launch_c8_bf16(p, st) : tfk_launch_c8(p, st);
  }
  else if (variant == 8) {
    if (!ar_ok(p) || c.bm != 128 || c.bn != 128 || p.splitk != 1) { tf_set_error("run_gemm: the activation-resident short-K kernel cannot run this launch (tile %dx%d, split %d, K %d)", c.bm, c.bn, p.splitk, p.K); return TF_E_UNSUPPORTED; }
    rc = bf ? tfk_launch_ar_bf16(p, st) : tfk_launch_ar(p, st);
  }
  else if (variant == 6) {
    if (p.splitk != 1 || !pp3_setup(p, c.bm, c.bn)) { tf_set_error("run_gemm: the patch form of the ping-pong kernel cannot run this launch (tile %dx%d, split %d)", c.bm, c.bn, p.splitk); return TF_E_UNSUPPORTED; }
    rc = bf ? tfk_launch_pp3_bf16(p, st, c.bn) : tfk_launch_pp3(p, st, c.bn);
  }
  else if (variant == 2 && patch_setup(p, c.bm, c.bn)) rc = bf ? tfk_launch_patch_bf16(p, st, c.bm, c.bn) : tfk_launch_patch(p, st, c.bm, c.bn);
  else if (c.bm == 256 && c.bn == 128) rc = bf ? tfk_launch_igemm_256x128_bf16(p, st) : tfk_launch_igemm_256x128(p, st);
  else if (c.bn == 160) rc = bf ? tfk_launch_igemm_160_bf16(p, st, c.bm, wide, all8) : tfk_launch_igemm_160(p, st, c.bm, wide, all8);
  else if (c.bn == 128) rc = bf ? tfk_launch_igemm_128_bf16(p, st, c.bm, wide, all8) : tfk_launch_igemm_128(p, st, c.bm, wide, all8);
  else if (c.bn == 64) rc = bf ? tfk_launch_igemm_64_bf16(p, st, c.bm, wide, all8) : tfk_launch_igemm_64(p, st, c.bm, wide, all8);
  else { tf_set_error("run_gemm: no kernel for tile %dx%d", c.bm, c.bn); return TF_E_UNSUPPORTED; }
  if (rc) return rc;
  if (g_prof_end) { TF_HIP(hipEventRecord(g_prof_end, st)); g_prof_end = nullptr; }   // the bracket holds k_igemm* alone (what rocprofv3 lists under that name)
  p.gn_part = gn_part;
  if (p.on_applied) *p.on_applied = 0;
  int rg_gpb = 0, rg_cv = 0, rg_rps = 0;
  size_t rg_lds = 0;
  if (p.splitk > 1 && p.on_z && p.gn_part && rga_geometry(p.HoWo, p.N, p.gn_G, &rg_gpb, &rg_cv, &rg_rps, &rg_lds)) {

    static bool attr_set = false;
    if (!attr_set) {
      TF_HIP(hipFuncSetAttribute((const void*)k_splitk_reduce_gn_apply<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      TF_HIP(hipFuncSetAttribute((const void*)k_splitk_reduce_gn_apply<half_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      TF_HIP(hipFuncSetAttribute((const void*)k_splitk_reduce_gn_apply<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    const int nimg = p.M / p.HoWo;
#define TF_RGA_LAUNCH(PT, BFV) hipLaunchKernelGGL((k_splitk_reduce_gn_apply<PT, BFV>), dim3(nimg * (p.gn_G / rg_gpb)), dim3(1024), rg_lds, st, p.y, p.on_z, (const PT*)p.partial, p.bias, p.bias_nc, \
                       p.residual, p.M, p.N, p.HoWo, p.splitk, p.bias_nc_stride, p.gn_part, p.gn_G, p.gn_cpg, rg_gpb, p.on_gamma, p.on_beta, p.on_eps, p.on_silu, rg_rps, rg_cv)
    if (bf) TF_RGA_LAUNCH(float, true);
    else if (p.part16) TF_RGA_LAUNCH(half_t, false);
    else TF_RGA_LAUNCH(float, false);
#undef TF_RGA_LAUNCH
    TF_LAUNCH_CHECK();
    if (p.on_applied) *p.on_applied = 1;
  } else if (p.splitk > 1 && p.gn_part) {
    const int R = p.HoWo / p.gn_chunks, nq = p.N >> 2;
    int RL = 1024 / nq;
    if (RL > R) RL = R;
    int threads = (RL * nq + 63) & ~63;
#define TF_RG_LAUNCH(PT, BFV) hipLaunchKernelGGL((k_splitk_reduce_gn<PT, BFV>), dim3(p.M / R), dim3(threads), (size_t)RL * p.N * 2 * sizeof(float), st, p.y, (const PT*)p.partial, \
                       p.bias, p.bias_nc, p.residual, p.M, p.N, p.HoWo, p.splitk, p.bias_nc_stride, p.gn_part, p.gn_G, p.gn_cpg, p.gn_chunks, R, RL)
    if (bf) TF_RG_LAUNCH(float, true);
    else if (p.part16) TF_RG_LAUNCH(half_t, false);
    else TF_RG_LAUNCH(float, false);
#undef TF_RG_LAUNCH
    TF_LAUNCH_CHECK();
  } else if (p.splitk > 1) {
    long long nv = ((long long)p.M * p.N) >> 2;
    int grid = (int)((nv + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
#define TF_R_LAUNCH(PT, BFV) hipLaunchKernelGGL((k_splitk_reduce<PT, BFV>), dim3(grid), dim3(256), 0, st, p.y, (const PT*)p.partial, p.bias, p.bias_nc, p.residual, p.M, p.N, p.HoWo, p.splitk, p.bias_nc_stride)
    if (bf) TF_R_LAUNCH(float, true);
    else if (p.part16) TF_R_LAUNCH(half_t, false);
    else TF_R_LAUNCH(float, false);
#undef TF_R_LAUNCH
    TF_LAUNCH_CHECK();
  }
  return TF_OK;
}

// ---- per-shape autotuner ("measure, don't guess"): the first eager call of a shape times every admissible
// (tile, split-K, ring variant) on the caller's own buffers with HIP events and caches the winner.  Never runs
// inside a stream capture (a captured shape that was never seen eagerly falls back to the cost model).
#define TF_SPLITK_WS_CAP ((size_t)64 << 20)
static int g_autotune = 1;     // 0: cost model only; 1: a shape missing from the table is tuned on its first eager use; 2: table only -- a missing shape is an error (every rank of a multi-GPU run must pick the same kernels)
struct TunedCfg { TileCfg c; int variant; int order; };   // variant: see launch_one
static std::map<std::array<int, 10>, TunedCfg> g_tuned;
static bool g_trace_keys = false;                       // tf_gemm_tune_trace: remember every shape key a launch looks up (tools/gemm_keys.py)
static std::map<std::array<int, 10>, bool> g_traced;

// untuned default for a launch that carries the input GroupNorm: the first admissible (tile, variant), split-K of the cost model
static TunedCfg gi_default(const GemmP& p) {
  static const int cand[][2] = {{64, 160}, {128, 160}, {64, 128}, {128, 128}, {64, 64}, {128, 64}};
  TileCfg m = choose_tiles(p.M, p.N, p.K, p.act, true);
  for (int ci = 0; ci < 6; ++ci)
    for (int v = (p.S == 3 ? 2 : 0); v < 4; ++v)
      if (gi_tile_ok(p, cand[ci][0], cand[ci][1], v)) {
        int sk = m.splitk;
        long long blocks = (long long)((p.M + cand[ci][0] - 1) / cand[ci][0]) * ((p.N + cand[ci][1] - 1) / cand[ci][1]);
        while (sk > 1 && (blocks * sk > 1024 || p.ktiles / sk < 4)) sk >>= 1;
        return {{cand[ci][0], cand[ci][1], sk}, v, 0};
      }
  return {m, 0, 0};
}

// block-scaled e4m3 launches: which (tile, split) of the ping-pong kernel a launch gets without a table row; tile.bm = 0 when none can take it
static TunedCfg mx_default(const GemmP& p) {
  static const int cand[][2] = {{192, 160}, {192, 128}, {256, 128}, {256, 160}};
  TunedCfg best = {{0, 0, 1}, 4, 0};
  {                                                        // the patch form where it applies: 2.0-2.4 against 1.2-1.45 PFLOP/s on config 5's 3x3 convs (tools/mx_bench.py)
    GemmP probe = p;
    if (pp3_setup(probe, 192, 128) && (long long)(p.M / 192) * ((p.N + 127) / 128) >= 128) return {{192, 128, 1}, 6, 1};
  }
  long long best_blocks = -1;
  for (int ci = 0; ci < 4; ++ci) {
    const int bm = cand[ci][0], bn = cand[ci][1];
    if (!pp_ok(p, bn, bm)) continue;
    if (bn == 160 && p.N % 160 != 0 && p.N % 128 == 0) continue;
    const long long blocks = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
    const long long waste = (long long)((p.M + bm - 1) / bm) * bm * (long long)((p.N + bn - 1) / bn) * bn - (long long)p.M * p.N;
    if (best_blocks < 0 || waste < best_blocks) { best_blocks = waste; best = {{bm, bn, 1}, 4, 0}; }
    (void)blocks;
  }
  return best;
}
#define TF_FLUSH_BYTES ((size_t)384 << 20)
static void* g_flush = nullptr;
static int autotune(const GemmP& p, void* workspace, size_t workspace_bytes, hipStream_t st, TunedCfg* out) {
  if (!g_flush) TF_HIP(hipMalloc(&g_flush, TF_FLUSH_BYTES));
  static const int cand[][2] = {{128, 160}, {64, 160}, {128, 128}, {64, 128}, {128, 64}, {64, 64}, {256, 128}};
  hipEvent_t a, b;
  TF_HIP(hipEventCreate(&a)); TF_HIP(hipEventCreate(&b));
  float best = 1e30f;
  TunedCfg bc = {choose_tiles(p.M, p.N, p.K, p.act, true), 0, 0};
  if (p.gi_part) bc = gi_default(p);
  if (p.mx) bc = mx_default(p);
  for (int ci = 0; ci < (p.mx ? 0 : p.fp8 ? kNumTiles8 : 7); ++ci) {        // (block-scaled launches have the ping-pong kernel only: below)
    int bm = p.fp8 ? kTiles8[ci][0] : cand[ci][0], bn = p.fp8 ? kTiles8[ci][1] : cand[ci][1];
    if (p.act == 1 && (bn % 64) != 0) continue;
    if (bm >= 128 && p.M <= 64) continue;
    if (bm == 256 && p.M <= 128) continue;
    // the 256x128 fp16 tile: plain deep ring, channel counts on the 64 grid, and only where it still leaves every CU a tile
    if (!p.fp8 && bm == 256 && (gemm_generic(p) || p.gi_part || p.ln_colsum || (long long)((p.M + 255) / 256) * ((p.N + 127) / 128) < 256)) continue;
    if (bn >= 128 && p.N <= 64) continue;
    for (int sk = 1; sk <= 32; sk *= 2) {
      if (sk > 1 && (p.act == 1 || p.ln_colsum || p.out32 || p.ktiles / sk < 4 || !workspace || (size_t)sk * p.M * p.N * 4 > workspace_bytes)) break;
      long long blocks = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * sk;
      if (sk > 1 && blocks > 1024) break;
      for (int wide = 0; wide < 4; ++wide) {                // the launch_one variants
        if ((p.fp8 || bm == 256) && wide != 0) continue;   // k_igemm8 and the 256-row tile have the deep ring only
        if (wide == 3 && gemm_generic(p)) continue;
        if (wide == 1 && (bm == 128 && bn == 160)) continue;
        if (wide == 1 && blocks <= 256) continue;          // two blocks per CU need more blocks than CUs
        if (wide == 2) { GemmP probe = p; if (!patch_setup(probe, bm, bn)) continue; }
        if (!gi_tile_ok(p, bm, bn, wide)) continue;
        TileCfg c = {bm, bn, sk};
        for (int order = 0; order < 2; ++order) {
          if (order == 1 && (p.M + bm - 1) / bm == 1) continue;   // a single m tile: both orders coincide
          GemmP q = p;
          if (q.gn_part && sk == 1 && !gn_tile_ok(q, bm, bn)) q.gn_part = nullptr;
          int rc = launch_one(q, c, wide, order, workspace, st);   // warm-up
          if (rc) return rc;
          // In the real step every layer's weights come from HBM (1.7 GB of weights per step never stay cached), so each
          // timed launch is preceded by a cache flush (a 384 MiB memset, outside the timed interval): median of 5.
          float tv[5];
          for (int r = 0; r < 5; ++r) {
            TF_HIP(hipMemsetAsync(g_flush, r, TF_FLUSH_BYTES, st));
            TF_HIP(hipEventRecord(a, st));
            rc = launch_one(q, c, wide, order, workspace, st);
            if (rc) return rc;
            TF_HIP(hipEventRecord(b, st));
            TF_HIP(hipEventSynchronize(b));
            TF_HIP(hipEventElapsedTime(&tv[r], a, b));
          }
          for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) if (tv[j] < tv[i]) { float t = tv[i]; tv[i] = tv[j]; tv[j] = t; }
          float ms = tv[2];
          if (ms < best) { best = ms; bc = {c, wide, order}; }
        }
      }
    }
  }
  // the ping-pong kernel (variant 4): 256 x {128, 160, 256} tiles for launches that still give most CUs a tile with them
  static const int ppbn[3] = {160, 128, 256};
  static const int ppbm[2] = {256, 192};
  for (int bi = 0; bi < 2; ++bi)
  for (int ci = 0; ci < 3; ++ci) {
    const int bm = ppbm[bi], bn = ppbn[ci];
    if (!pp_ok(p, bn, bm) || p.M <= 256) continue;
    for (int sk = 1; sk <= 32; sk *= 2) {        // (round 4: up to 32 -- at the 16 x 16 level a 256-row tile halves the weight re-reads per CU, and only a deep split fills the chip with it)
      if (sk > 1 && (p.act == 1 || p.out32 || p.ln_colsum || ktiles_for(p, 4) / sk < 4 || !workspace || (size_t)sk * p.M * p.N * 4 > workspace_bytes)) break;
      long long blocks = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * sk;
      if (blocks < 128) continue;
      if (sk > 1 && blocks > 1024) break;
      TileCfg c = {bm, bn, sk};
      for (int order = 0; order < 2; ++order) {
        GemmP q = p;
        if (q.gn_part && sk == 1 && !gn_tile_ok(q, bm / 2, bn)) q.gn_part = nullptr;
        int rc = launch_one(q, c, 4, order, workspace, st);   // warm-up
        if (rc) return rc;
        float tv[5];
        for (int r = 0; r < 5; ++r) {
          TF_HIP(hipMemsetAsync(g_flush, r, TF_FLUSH_BYTES, st));
          TF_HIP(hipEventRecord(a, st));
          rc = launch_one(q, c, 4, order, workspace, st);
          if (rc) return rc;
          TF_HIP(hipEventRecord(b, st));
          TF_HIP(hipEventSynchronize(b));
          TF_HIP(hipEventElapsedTime(&tv[r], a, b));
        }
        for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) if (tv[j] < tv[i]) { float t = tv[i]; tv[i] = tv[j]; tv[j] = t; }
        if (tv[2] < best) { best = tv[2]; bc = {c, 4, order}; }
      }
    }
  }
  // the patch form of the ping-pong kernel (variant 6): 192 x {160, 128} tiles, one launch
  for (int ci = 0; ci < 2; ++ci) {
    const int bn = ppbn[ci];
    GemmP probe = p;
    if (!pp3_setup(probe, 192, bn)) continue;
    if ((long long)(p.M / 192) * ((p.N + bn - 1) / bn) < 128) continue;
    TileCfg c = {192, bn, 1};
    for (int order = 0; order < 2; ++order) {
      GemmP q = p;
      if (q.gn_part && !gn_tile_ok(q, 96, bn)) q.gn_part = nullptr;
      int rc = launch_one(q, c, 6, order, workspace, st);   // warm-up
      if (rc) return rc;
      float tv[5];
      for (int r = 0; r < 5; ++r) {
        TF_HIP(hipMemsetAsync(g_flush, r, TF_FLUSH_BYTES, st));
        TF_HIP(hipEventRecord(a, st));
        rc = launch_one(q, c, 6, order, workspace, st);
        if (rc) return rc;
        TF_HIP(hipEventRecord(b, st));
        TF_HIP(hipEventSynchronize(b));
        TF_HIP(hipEventElapsedTime(&tv[r], a, b));
      }
      for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) if (tv[j] < tv[i]) { float t = tv[i]; tv[i] = tv[j]; tv[j] = t; }
      if (tv[2] < best) { best = tv[2]; bc = {c, 6, order}; }
    }
  }
  // the persistent short-K kernel (variant 5): a candidate once its 128 x 128 tiles occupy a good part of the CUs (with fewer tiles than
  // blocks it is simply a 4-wave kernel with a register epilogue: 8192 x 320 x 320 8.2 vs 9.0 us, 2048 x 1920 x 640 12.0 vs 13.6)
  if (c4_ok(p) && (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) >= 96) {
    TileCfg c = {128, 128, 1};
    for (int order = 0; order < 2; ++order) {
      int rc = launch_one(p, c, 5, order, workspace, st);   // warm-up
      if (rc) return rc;
      float tv[5];
      for (int r = 0; r < 5; ++r) {
        TF_HIP(hipMemsetAsync(g_flush, r, TF_FLUSH_BYTES, st));
        TF_HIP(hipEventRecord(a, st));
        rc = launch_one(p, c, 5, order, workspace, st);
        if (rc) return rc;
        TF_HIP(hipEventRecord(b, st));
        TF_HIP(hipEventSynchronize(b));
        TF_HIP(hipEventElapsedTime(&tv[r], a, b));
      }
      for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) if (tv[j] < tv[i]) { float t = tv[i]; tv[i] = tv[j]; tv[j] = t; }
      if (tv[2] < best) { best = tv[2]; bc = {c, 5, order}; }
    }
  }
  // the activation-resident short-K kernel (variant 8; K = 256 / 320): the tile order is its own (n fastest inside a block's run)
  if (ar_ok(p) && (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) >= 96) {
    TileCfg c = {128, 128, 1};
    int rc = launch_one(p, c, 8, 0, workspace, st);   // warm-up
    if (rc) return rc;
    float tv[5];
    for (int r = 0; r < 5; ++r) {
      TF_HIP(hipMemsetAsync(g_flush, r, TF_FLUSH_BYTES, st));
      TF_HIP(hipEventRecord(a, st));
      rc = launch_one(p, c, 8, 0, workspace, st);
      if (rc) return rc;
      TF_HIP(hipEventRecord(b, st));
      TF_HIP(hipEventSynchronize(b));
      TF_HIP(hipEventElapsedTime(&tv[r], a, b));
    }
    for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) if (tv[j] < tv[i]) { float t = tv[i]; tv[i] = tv[j]; tv[j] = t; }
    if (tv[2] < best) { best = tv[2]; bc = {c, 8, 0}; }
  }
  // the 256-row persistent short-K kernel (variant 7): one 8-wave block per CU walks 256 x 128 tiles.  MEASURED SLOWER than k_gemm_c4 on every shape it was
  // built for (profiles/r05_c8_bench.txt: 5-20 %: eight waves in lockstep idle the matrix pipe during every epilogue, where k_gemm_c4's two independent blocks
  // overlap one's epilogue with the other's K loop), so the tuner tries it only when asked (TF_TUNE_C8=1); table rows and tf_gemm_debug(16384) still select it
  static const bool tune_c8 = getenv("TF_TUNE_C8") != nullptr;
  if (tune_c8 && c4_ok(p) && (long long)((p.M + 255) / 256) * ((p.N + 127) / 128) >= 192) {
    TileCfg c = {256, 128, 1};
    for (int order = 0; order < 2; ++order) {
      int rc = launch_one(p, c, 7, order, workspace, st);   // warm-up
      if (rc) return rc;
      float tv[5];
      for (int r = 0; r < 5; ++r) {
        TF_HIP(hipMemsetAsync(g_flush, r, TF_FLUSH_BYTES, st));
        TF_HIP(hipEventRecord(a, st));
        rc = launch_one(p, c, 7, order, workspace, st);
        if (rc) return rc;
        TF_HIP(hipEventRecord(b, st));
        TF_HIP(hipEventSynchronize(b));
        TF_HIP(hipEventElapsedTime(&tv[r], a, b));
      }
      for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) if (tv[j] < tv[i]) { float t = tv[i]; tv[i] = tv[j]; tv[j] = t; }
      if (tv[2] < best) { best = tv[2]; bc = {c, 7, order}; }
    }
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  *out = bc;
  return TF_OK;
}

#if TF_IGEMM_STAMP
// diagnostic build (tools/igemm_stamp.py): every k_igemm launch writes its blocks' phase stamps here; tf_debug_stamps copies them out
static unsigned long long* g_stamp_buf = nullptr;
#define TF_STAMP_BLOCKS 8192
extern "C" int tf_debug_stamps(void* host_out, int blocks) {
  TF_REQUIRE(host_out && blocks >= 1 && blocks <= TF_STAMP_BLOCKS && g_stamp_buf, "tf_debug_stamps: no stamps");
  TF_HIP(hipMemcpy(host_out, g_stamp_buf, (size_t)blocks * 64, hipMemcpyDeviceToHost));
  return TF_OK;
}
extern "C" int tf_debug_loop_stamps(void* host_out) {      // TF_IGEMM_STAMP == 2: 256 K tiles x 8 u32 stamps of block 0 (behind the per-block table)
  TF_REQUIRE(host_out && g_stamp_buf, "tf_debug_loop_stamps: no stamps");
  TF_HIP(hipMemcpy(host_out, g_stamp_buf + (size_t)TF_STAMP_BLOCKS * 8, 256 * 8 * 4, hipMemcpyDeviceToHost));
  return TF_OK;
}
#endif
static int run_gemm(GemmP p, void* workspace, size_t workspace_bytes, int force_bm, int force_bn, int force_split, hipStream_t st, int* gn_chunks = nullptr) {
#if TF_IGEMM_STAMP
  if (!g_stamp_buf) { TF_HIP(hipMalloc((void**)&g_stamp_buf, (size_t)TF_STAMP_BLOCKS * 64 + 256 * 8 * 4)); TF_HIP(hipMemset(g_stamp_buf, 0, (size_t)TF_STAMP_BLOCKS * 64 + 256 * 8 * 4)); }
  p.stamp = g_stamp_buf;
#endif
  p.ktiles = (p.K + 63) / 64;
  p.dbg = g_dbg;
  fast_div_magic((unsigned)p.HoWo, &p.dv_howo_mul, &p.dv_howo_shr);
  fast_div_magic((unsigned)p.Wo, &p.dv_wo_mul, &p.dv_wo_shr);
  TunedCfg t = {choose_tiles(p.M, p.N, p.K, p.act, true), 0, 0};
  bool tuned = false;
  if (p.gi_part) { t = gi_default(p); tuned = true; }    // (tuned: keep gi_default's variant unless the tuner knows better)
  TunedCfg fp8_default = t;
  if (p.fp8 && !p.mx) {                                   // untuned fp8 default: the widest tile that still gives every CU a block
    long long b128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    t.c = {p.M >= 128 ? 128 : 64, p.act == 1 || p.N >= 128 ? 128 : 64, b128 >= 128 ? 1 : t.c.splitk};
    t.variant = 0; tuned = true;
    fp8_default = t;
  }
  if (p.mx) {                                             // block-scaled e4m3: the ping-pong kernel or nothing (tf_mx8_gemm_supported tells a caller beforehand)
    t = mx_default(p);
    if (!t.c.bm) { tf_set_error("run_gemm: no block-scaled e4m3 kernel for M=%d N=%d K=%d (S=%d stride=%d ups=%d act=%d): ask tf_mx8_gemm_supported first", p.M, p.N, p.K, p.S, p.stride, p.ups, p.act); return TF_E_UNSUPPORTED; }
    tuned = true;
    fp8_default = t;
  }
  if (false) {}
  else if (force_bm) {
    t.c = {force_bm, force_bn, force_split > 0 ? force_split : 1};
    t.order = g_force_order > 0 ? 1 : 0;
    if (p.gi_part) t.variant = p.S == 3 ? 2 : 0;
  } else if (g_autotune && !g_dbg) {
    std::array<int, 10> key = {p.M, p.N, p.K, p.C1, p.C2, p.S, p.stride, p.ups, p.act,
                               (p.bias ? 1 : 0) | (p.residual ? 2 : 0) | (p.bias_nc ? 4 : 0) | (p.ln_colsum ? 8 : 0) | (p.gi_part ? 16 : 0) | (p.fp8 ? 64 : 0) | (p.out8 ? 128 : 0) | (p.out32 ? 256 : 0) | (p.mx ? 512 : 0) | (p.bf16 ? 1024 : 0)};   // (on_z shares the plain key: same tile, another reduce kernel)
    auto it = g_tuned.find(key);
    if (it == g_tuned.end() && p.bf16) {                    // a bfloat16 launch without a row of its own takes the fp16 row of the shape: the same kernels, the same bytes and FLOPs
      std::array<int, 10> k16 = key;
      k16[9] &= ~1024;
      it = g_tuned.find(k16);
    }
    if (g_trace_keys) g_traced[key] = it != g_tuned.end();
    if (it != g_tuned.end()) {
      t = it->second; tuned = true;
      // a table row (shipped, or loaded from a user's file) whose tile cannot carry this launch's input GroupNorm -- the key holds
      // only a gi flag, not HoWo / H / W -- falls back to the first admissible tile instead of failing the forward
      if (p.gi_part && !gi_tile_ok(p, t.c.bm, t.c.bn, t.variant)) t = gi_default(p);
      // e4m3 rows: a fixed-scale launch has k_igemm8 only, a block-scaled one the ping-pong kernel only -- a row that says otherwise (an
      // older table, a user's file) falls back to the default instead of failing the forward
      if (p.fp8) {
        GemmP probe = p;
        const bool blk = t.variant == 4 || t.variant == 6;
        if (blk != (p.mx != 0) || (t.variant == 4 && !pp_ok(p, t.c.bn, t.c.bm)) || (t.variant == 6 && !pp3_setup(probe, t.c.bm, t.c.bn))) t = fp8_default;
      }
    }
    else if (g_autotune == 2) {
      tf_set_error("run_gemm: shape M=%d N=%d K=%d C1=%d C2=%d S=%d stride=%d ups=%d act=%d flags=%d is not in the tuning table and tuning is off "
                   "(tf_gemm_autotune(2): with WORLD_SIZE > 1 every rank must run the same kernels); tune it on one GPU (tools/tune_best.sh) and ship the row",
                   key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7], key[8], key[9]);
      return TF_E_STATE;
    }
    else {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(st, &cs);
      if (cs == hipStreamCaptureStatusNone) {
        int rc = autotune(p, workspace, workspace_bytes, st, &t);
        if (rc) return rc;
        g_tuned[key] = t;
        tuned = true;
      }
    }
  }
  if (p.ln_colsum || p.out32) t.c.splitk = 1;            // row statistics need the whole K range in one block; so does the raw fp32 output
  if (t.c.splitk > 1) {
    size_t need = (size_t)t.c.splitk * p.M * p.N * sizeof(float);
    if (!workspace || workspace_bytes < need) t.c.splitk = 1;   // degrade gracefully: correctness does not depend on split-K
  }
  int wide = t.variant;
  if (!tuned) {
    // cost-model fallback: WIDE (two blocks per CU) pays when a CU gets several tiles with a short K loop each
    long long blocks = (long long)((p.M + t.c.bm - 1) / t.c.bm) * ((p.N + t.c.bn - 1) / t.c.bn) * t.c.splitk;
    wide = (blocks > 256 && p.ktiles / t.c.splitk <= 24) ? 1 : 0;
  }
  if (g_force_wide >= 0 && !(p.gi_part && p.S == 3)) wide = g_force_wide;
  if (wide == 4 && !pp_ok(p, t.c.bn, t.c.bm)) {   // a table configuration this launch cannot take falls back; an explicit request fails
    if (g_force_wide == 4) { tf_set_error("run_gemm: the ping-pong kernel cannot run this launch (tile %dx%d)", t.c.bm, t.c.bn); return TF_E_UNSUPPORTED; }
    wide = 0;
    if (t.c.bm >= 192) t.c = choose_tiles(p.M, p.N, p.K, p.act, workspace != nullptr);
  }
  if (wide == 6) {
    // the tile is the kernel's own: a forced tile keeps its width where the kernel has it; a table row this launch cannot take falls back
    GemmP q = p;
    const int bn6 = (p.Wo == 96 && !p.fp8) ? 160 : 128;
    if (pp3_setup(q, 192, bn6)) t.c = {192, bn6, 1};
    else if (g_force_wide == 6) { tf_set_error("run_gemm: the patch form of the ping-pong kernel cannot run this launch"); return TF_E_UNSUPPORTED; }
    else { wide = 0; if (t.c.bm >= 192) t.c = choose_tiles(p.M, p.N, p.K, p.act, workspace != nullptr); }
  }
  if (wide == 7) {
    GemmP q = p;
    if (!gn_chunks) q.gn_part = nullptr;
    if (c4_ok(q) && (!force_bm || (t.c.bm == 256 && t.c.bn == 128 && t.c.splitk == 1))) t.c = {256, 128, 1};
    else if (g_force_wide == 7) { tf_set_error("run_gemm: the 256-row persistent short-K kernel cannot run this launch"); return TF_E_UNSUPPORTED; }
    else { wide = 0; t.c = choose_tiles(p.M, p.N, p.K, p.act, workspace != nullptr); }
  }
  if (wide == 8) {
    GemmP q = p;
    if (!gn_chunks) q.gn_part = nullptr;
    if (ar_ok(q) && (!force_bm || (t.c.bm == 128 && t.c.bn == 128 && t.c.splitk == 1))) t.c = {128, 128, 1};
    else if (g_force_wide == 8) { tf_set_error("run_gemm: the activation-resident short-K kernel cannot run this launch"); return TF_E_UNSUPPORTED; }
    else wide = c4_ok(q) ? 5 : 0;                         // (a table row of another K: the persistent kernel it grew out of)
  }
  if (wide == 5) {
    GemmP q = p;
    if (!gn_chunks) q.gn_part = nullptr;                  // (statistics the caller did not ask to hear about are never requested)
    if (c4_ok(q) && (!force_bm || (t.c.bm == 128 && t.c.bn == 128 && t.c.splitk == 1))) t.c = {128, 128, 1};
    else if (g_force_wide == 5) { tf_set_error("run_gemm: the persistent short-K kernel cannot run this launch"); return TF_E_UNSUPPORTED; }
    else wide = 0;
  }
  ProfRec rec;
  if (g_prof) {
    TF_HIP(hipEventCreate(&rec.a)); TF_HIP(hipEventCreate(&rec.b)); TF_HIP(hipEventCreate(&rec.c));
    rec.flops = 2.0 * p.M * (double)p.N * p.K;
    rec.M = p.M; rec.N = p.N; rec.K = p.K; rec.taps = p.Kc / p.C; rec.bm = t.c.bm; rec.bn = t.c.bn; rec.splitk = t.c.splitk; rec.variant = wide;
    TF_HIP(hipEventRecord(rec.a, st));
  }
  if (g_force_order >= 0) t.order = g_force_order;
  if (p.gn_part) {
    // statistics ride along only when the chosen tiling maps m-tiles onto whole images; otherwise the caller is told
    // (chunks = 0) and runs the stand-alone statistics pass
    const int eff = eff_splitk(p, wide, t.c.splitk);
    TileCfg sc = t.c;
    sc.bm = stats_bm(t.c.bm, wide);
    if (eff == 1 && !gn_tile_ok(p, sc.bm, sc.bn)) p.gn_part = nullptr;
    if (gn_chunks) *gn_chunks = p.gn_part ? gn_chunks_for(p, sc, eff) : 0;
    int a_, b_, c_; size_t d_;
    if (gn_chunks && p.gn_part && eff > 1 && p.on_z && rga_geometry(p.HoWo, p.N, p.gn_G, &a_, &b_, &c_, &d_)) *gn_chunks = 1;   // the fused reduce leaves whole-image sums
  }
  if (g_prof) g_prof_end = rec.b;
  int rc = launch_one(p, t.c, wide, t.order, workspace, st);
  g_prof_end = nullptr;
  if (rc) return rc;
  if (g_prof) {
    // the second bracket only where a reduce launch followed the GEMM (an event pair of its own costs ~2 us of stream time)
    const int eff = eff_splitk(p, wide, t.c.splitk);
    rec.has_reduce = eff > 1;
    // algorithmic bytes of the reduce launch: the slabs in, y out (+ the normalised z of the fused form), bias rows negligible
    const double mn = (double)p.M * p.N;
    rec.reduce_bytes = mn * eff * ((g_part16 && !p.bf16 && (p.N & 7) == 0) ? 2.0 : 4.0) + mn * 2.0 * (1.0 + (p.residual ? 1.0 : 0.0) + ((p.on_z && p.gn_part) ? 1.0 : 0.0));
    if (rec.has_reduce) TF_HIP(hipEventRecord(rec.c, st));
    g_prof_pending.push_back(rec);
  }
  return TF_OK;
}

// workspace the caller must provide: enough for any split-K the tuner may pick (capped)
static size_t gemm_workspace(int M, int N, int K, int act) {
  if (act == 1 || (K + 63) / 64 < 8) return 0;
  size_t per = (size_t)M * N * sizeof(float);
  int sk = 32;
  while (sk > 1 && ((size_t)sk * per > TF_SPLITK_WS_CAP || (K + 63) / 64 / sk < 4)) sk >>= 1;
  return sk > 1 ? (size_t)sk * per : 0;
}

// test hook: force a tile configuration (0 = heuristic)
static int g_force_bm = 0, g_force_bn = 0, g_force_split = 0;

extern "C" {

int tf_gemm_debug(int flags) {
#ifdef TF_ABLATION
  g_dbg = (flags & 7) | ((flags & 4096) ? 8 : 0);         // 1 no stores, 2 no MFMA, 4 no staging, 4096 no fragment reads (k_igemm_pp): the ablation library only
#else
  TF_REQUIRE(!(flags & (7 | 4096)), "tf_gemm_debug: the ablation bits (1, 2, 4, 4096) exist only in the library built with -DTF_ABLATION (python -m tinyfusers_amd.build --ablation)");
#endif
  g_pp_np = (flags & 8192) ? 2 : 0;                       // 8192: k_igemm_pp with one phase per k-step even where the 3-slot ring allows one per K tile
  g_force_wide = (flags & 32768) ? 8 : (flags & 16384) ? 7 : (flags & 2048) ? 6 : (flags & 1024) ? 5 : (flags & 512) ? 4 : (flags & 256) ? 3 : (flags & 128) ? 2 : (flags & 16) ? 1 : (flags & 8) ? 0 : -1;   // 128 / 256 / 512 / 1024 / 2048: the PATCH / ALL8 / ping-pong / persistent short-K / ping-pong PATCH variants where eligible
  g_force_order = (flags & 64) ? 1 : (flags & 32) ? 0 : -1;
  return TF_OK;
}
int tf_gemm_autotune(int mode) {
  TF_REQUIRE(mode >= 0 && mode <= 2, "tf_gemm_autotune: mode=%d (0 cost model only, 1 tune missing shapes on first use, 2 table only: a missing shape is an error)", mode);
  g_autotune = mode;
  if (!mode) g_tuned.clear();
  return TF_OK;
}
// host-side view of the table (no device work): what the launch of a shape would pick.  key = {M, N, K, C1, C2, S, stride, upsample, act, flags}
// as tf_gemm_tune_save writes them, cfg = {bm, bn, splitk, variant, order}; TF_E_STATE when the shape has no row
int tf_gemm_tune_query(const int* key, int* cfg) {
  TF_REQUIRE(key && cfg, "tf_gemm_tune_query: null argument");
  std::array<int, 10> k;
  for (int i = 0; i < 10; ++i) k[i] = key[i];
  auto it = g_tuned.find(k);
  if (it == g_tuned.end()) { tf_set_error("tf_gemm_tune_query: shape M=%d N=%d K=%d has no row", key[0], key[1], key[2]); return TF_E_STATE; }
  cfg[0] = it->second.c.bm; cfg[1] = it->second.c.bn; cfg[2] = it->second.c.splitk; cfg[3] = it->second.variant; cfg[4] = it->second.order;
  return TF_OK;
}
// which shapes does a workload consult?  tf_gemm_tune_trace(1) starts remembering every key a launch looks up (and whether it had a row),
// tf_gemm_tune_trace_dump writes them, one per line: the ten key fields and 1 / 0 (tools/gemm_keys.py -> tests/golden/gemm_keys.json)
int tf_gemm_tune_trace(int on) { g_trace_keys = on != 0; if (on) g_traced.clear(); return TF_OK; }
int tf_gemm_tune_trace_dump(const char* path) {
  TF_REQUIRE(path, "tf_gemm_tune_trace_dump: null path");
  FILE* f = fopen(path, "w");
  TF_REQUIRE(f, "tf_gemm_tune_trace_dump: cannot open %s", path);
  for (auto& kv : g_traced) {
    for (int i = 0; i < 10; ++i) fprintf(f, "%d ", kv.first[i]);
    fprintf(f, "%d\n", kv.second ? 1 : 0);
  }
  fclose(f);
  return TF_OK;
}
int tf_gemm_tune_count(int* n) { TF_REQUIRE(n, "tf_gemm_tune_count: null argument"); *n = (int)g_tuned.size(); return TF_OK; }
int tf_gemm_tune_entry(int index, int* key, int* cfg) {
  TF_REQUIRE(key && cfg && index >= 0 && index < (int)g_tuned.size(), "tf_gemm_tune_entry: index %d out of range", index);
  auto it = g_tuned.begin();
  std::advance(it, index);
  for (int i = 0; i < 10; ++i) key[i] = it->first[i];
  cfg[0] = it->second.c.bm; cfg[1] = it->second.c.bn; cfg[2] = it->second.c.splitk; cfg[3] = it->second.variant; cfg[4] = it->second.order;
  return TF_OK;
}
// persist / restore the tuner's choices (one line per shape) so that profiled or repeated runs skip the tuning launches
int tf_gemm_tune_save(const char* path) {
  TF_REQUIRE(path, "tf_gemm_tune_save: null path");
  FILE* f = fopen(path, "w");
  TF_REQUIRE(f, "tf_gemm_tune_save: cannot open %s", path);
  for (auto& kv : g_tuned) {
    for (int i = 0; i < 10; ++i) fprintf(f, "%d ", kv.first[i]);
    fprintf(f, "%d %d %d %d %d\n", kv.second.c.bm, kv.second.c.bn, kv.second.c.splitk, kv.second.variant, kv.second.order);
  }
  fclose(f);
  return TF_OK;
}
int tf_gemm_tune_load(const char* path) {
  TF_REQUIRE(path, "tf_gemm_tune_load: null path");
  FILE* f = fopen(path, "r");
  if (!f) return TF_OK;                                  // no cache yet: tune on first use
  std::array<int, 10> k; int bm, bn, sk, wide, order;
  for (;;) {
    int n = 0;
    for (int i = 0; i < 10; ++i) n += fscanf(f, "%d", &k[i]);
    n += fscanf(f, "%d %d %d %d %d", &bm, &bn, &sk, &wide, &order);
    if (n != 15) break;
    const bool f8 = (k[9] & 64) != 0;
    bool ok = (bm == 64 || bm == 128 || (f8 && bm == 256 && bn == 64) || (!f8 && bm == 256 && bn == 128) || wide == 4) && (bn == 64 || bn == 128 || (!f8 && bn == 160) || wide == 4) &&
              sk >= 1 && sk <= 32;
    if (((f8 && wide != 4) || (bm == 256 && wide != 4 && wide != 7)) && wide != 0) ok = false;
    if (wide == 4) ok = ((bm == 256 && (bn == 128 || bn == 160 || (!f8 && bn == 256))) || (bm == 192 && (bn == 128 || bn == 160))) && sk >= 1 && sk <= 32;
    // rows the tuner itself never emits: GEGLU (act = 1) pairs 16-row value|gate blocks inside a wave tile (bn % 64 == 0), and
    // neither GEGLU nor the LayerNorm fold (flag bit 8) can be split along K
    const int act = k[8], ln = k[9] & 8;
    if (act == 1 && (bn % 64) != 0) ok = false;
    if ((act == 1 || ln || (k[9] & 256)) && sk > 1) ok = false;
    if (wide == 5) ok = bm == 128 && bn == 128 && sk == 1 && !f8;
    if (wide == 7) ok = bm == 256 && bn == 128 && sk == 1 && !f8;
    if (wide == 8) ok = bm == 128 && bn == 128 && sk == 1 && !f8;
    if (wide == 6) ok = bm == 192 && (bn == 128 || bn == 160) && sk == 1 && k[5] == 3 && k[6] == 1 && act == 0 && !ln && (!f8 || (k[9] & 512));   // the patch form: 3x3 / stride 1; e4m3 only block-scaled
    if (ok) g_tuned[k] = {{bm, bn, sk}, wide < 0 || wide > 8 ? 0 : wide, order != 0 ? 1 : 0};
  }
  fclose(f);
  return TF_OK;
}
// element type of the split-K partial slabs: 16 = fp16 (default; half the bytes of the split-K seam, fp32 accumulation in the reducer), 32 = fp32
int tf_gemm_splitk_partials(int bits) {
  TF_REQUIRE(bits == 16 || bits == 32, "tf_gemm_splitk_partials: bits=%d (16 or 32)", bits);
  g_part16 = bits == 16;
  return TF_OK;
}
int tf_gemm_force_config(int bm, int bn, int splitk) { g_force_bm = bm; g_force_bn = bn; g_force_split = splitk; return TF_OK; }

// What an event bracket [record a][kernel][record b] reads beyond the kernel's own begin-to-end duration (the figure rocprofv3 lists): the
// dispatch and completion latencies around it.  Rounds 1-2 subtracted the reading of an EMPTY pair, which over-corrects (the brackets then
// read ~2 us per launch shorter than rocprofv3: VERDICT r2, 2.98 vs 3.31 ms per step); no correction reads ~2.4 us per launch longer.  So
// the overhead is measured as what it is: brackets around a kernel that spins for T and for 2 T of the constant-rate clock read o + T and
// o + 2 T, hence o = 2 b(T) - b(2 T) (median of 9 pairs, T = 20 us).
__global__ void k_prof_spin(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
static float g_prof_overhead_ms = 0.f;
int tf_prof_enable(int on) {
  g_prof = on != 0;
  if (on) {
    g_prof_ms = 0.0; g_prof_ms_full = 0.0; g_prof_flops = 0.0; g_prof_launches = 0; g_prof_pending.clear(); g_prof_shapes.clear();
    tf_prof_fam_reset();
    hipEvent_t a, b;
    TF_HIP(hipEventCreate(&a)); TF_HIP(hipEventCreate(&b));
    float v[9];
    for (int i = 0; i < 9; ++i) {
      float t1 = 0.f, t2 = 0.f;
      TF_HIP(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_prof_spin, dim3(1), dim3(64), 0, 0, 2000LL); TF_HIP(hipEventRecord(b, 0));
      TF_HIP(hipEventSynchronize(b)); TF_HIP(hipEventElapsedTime(&t1, a, b));
      TF_HIP(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_prof_spin, dim3(1), dim3(64), 0, 0, 4000LL); TF_HIP(hipEventRecord(b, 0));
      TF_HIP(hipEventSynchronize(b)); TF_HIP(hipEventElapsedTime(&t2, a, b));
      v[i] = 2.f * t1 - t2;
    }
    for (int i = 0; i < 9; ++i) for (int j = i + 1; j < 9; ++j) if (v[j] < v[i]) { float t = v[i]; v[i] = v[j]; v[j] = t; }
    g_prof_overhead_ms = v[4] > 0.f ? v[4] : 0.f;
    g_tf_prof_overhead_ms = g_prof_overhead_ms;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  }
  return TF_OK;
}
float tf_prof_overhead_us() { return g_prof_overhead_ms * 1e3f; }
static int prof_collect() {
  for (auto& r : g_prof_pending) {
    float t = 0.f, tf = 0.f;
    TF_HIP(hipEventSynchronize(r.has_reduce ? r.c : r.b));
    TF_HIP(hipEventElapsedTime(&t, r.a, r.b));
    t -= g_prof_overhead_ms;
    if (t < 0.f) t = 0.f;
    tf = t;
    if (r.has_reduce) {                                    // GEMM bracket + the reduce's own bracket (b .. c): the event in between is not charged twice
      float tr = 0.f;
      TF_HIP(hipEventElapsedTime(&tr, r.b, r.c));
      tr -= g_prof_overhead_ms;
      if (tr > 0.f) tf += tr;
      tf_prof_fam_add(TF_PROF_FAM_SPLITK_REDUCE, r.reduce_bytes, tr > 0.f ? tr : 0.0);
    }
    g_prof_ms += t; g_prof_ms_full += tf; g_prof_flops += r.flops; g_prof_launches += 1;
    auto& e = g_prof_shapes[{r.M, r.N, r.K, r.taps, r.bm, r.bn, r.splitk, r.variant}];
    e.first += 1; e.second += tf;
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); (void)hipEventDestroy(r.c);
  }
  g_prof_pending.clear();
  return TF_OK;
}
int tf_prof_read(double* ms, double* flops, long long* launches) {
  int rc = prof_collect();
  if (rc) return rc;
  if (ms) *ms = g_prof_ms;
  if (flops) *flops = g_prof_flops;
  if (launches) *launches = g_prof_launches;
  return TF_OK;
}
int tf_prof_read_full(double* ms_with_reduce, double* ms_gemm_kernel_only, double* flops, long long* launches) {
  int rc = prof_collect();
  if (rc) return rc;
  if (ms_with_reduce) *ms_with_reduce = g_prof_ms_full;
  if (ms_gemm_kernel_only) *ms_gemm_kernel_only = g_prof_ms;
  if (flops) *flops = g_prof_flops;
  if (launches) *launches = g_prof_launches;
  return TF_OK;
}

int tf_prof_dump(const char* path) {
  TF_REQUIRE(path, "tf_prof_dump: null path");
  int rc = tf_prof_read(nullptr, nullptr, nullptr);
  if (rc) return rc;
  FILE* f = fopen(path, "w");
  TF_REQUIRE(f, "tf_prof_dump: cannot open %s", path);
  fprintf(f, "M,N,K,taps,bm,bn,splitk,variant,launches,total_ms,avg_us,tflops\n");   // variant: 0 deep ring, 1 wide, 2 patch, 3 all8, 4 ping-pong, 5 persistent short-K, 6 ping-pong patch, 7 persistent short-K 256-row, 8 activation-resident short-K; times include the split-K reduce
  for (auto& kv : g_prof_shapes) {
    const auto& k = kv.first;
    double ms = kv.second.second; long long n = kv.second.first;
    double tf = 2.0 * k[0] * (double)k[1] * k[2] * n / (ms * 1e-3) / 1e12;
    fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%d,%lld,%.4f,%.2f,%.1f\n", k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], n, ms, ms * 1e3 / n, tf);
  }
  fclose(f);
  return TF_OK;
}

static int conv_geometry(int H, int W, int R, int S, int stride, int pad, int ups, int* Ho, int* Wo) {
  int Hl = H << ups, Wl = W << ups;
  *Ho = (Hl + 2 * pad - R) / stride + 1;
  *Wo = (Wl + 2 * pad - S) / stride + 1;
  return (*Ho > 0 && *Wo > 0) ? 0 : 1;
}

size_t tf_conv2d_gn_partial_bytes(int N, int groups) { return (size_t)(N > 0 ? N : 0) * TF_GN_MAX_CHUNKS * (groups > 0 ? groups : 0) * 2 * sizeof(float); }

size_t tf_conv2d_workspace(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample) {
  int Ho, Wo;
  if (stride < 1 || conv_geometry(H, W, R, S, stride, pad, upsample ? 1 : 0, &Ho, &Wo)) return 0;
  return gemm_workspace(N * Ho * Wo, Cout, R * S * (C1 + C2), 0);
}

static int conv2d_impl(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                       const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                       void* workspace, size_t workspace_bytes, float* gn_partial, size_t gn_partial_bytes, int gn_groups, int* gn_chunks,
                       const void* x3, const void* x4, int C3, int C4, tfStream_t s, const GemmP* gi = nullptr) {
  if (gn_chunks) *gn_chunks = 0;
  TF_REQUIRE(C3 >= 0 && C4 >= 0 && (C3 == 0 || x3) && (C4 == 0 || (x4 && C3 > 0)) && C3 % 8 == 0 && C4 % 8 == 0,
             "tf_conv2d_fused_f16: extra sources C3=%d C4=%d must be multiples of 8 with their tensors given (x4 needs x3)", C3, C4);
  TF_REQUIRE(C3 == 0 || upsample == 0, "tf_conv2d_fused_f16: the extra 1x1 sources cannot be combined with upsample");
  TF_REQUIRE(y && x && w, "tf_conv2d_f16: null tensor");
  TF_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2), "tf_conv2d_f16: C1=%d C2=%d x2=%p", C1, C2, x2);
  TF_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0, "tf_conv2d_f16: channel counts must be multiples of 8 (C1=%d C2=%d); use tf_im2col_nhwc_f16 for tiny C", C1, C2);
  TF_REQUIRE(R >= 1 && S >= 1 && stride >= 1 && pad >= 0 && Cout >= 1 && N >= 0, "tf_conv2d_f16: bad geometry R=%d S=%d stride=%d pad=%d", R, S, stride, pad);
  int ups = upsample ? 1 : 0, Ho, Wo;
  TF_REQUIRE(!conv_geometry(H, W, R, S, stride, pad, ups, &Ho, &Wo), "tf_conv2d_f16: empty output for H=%d W=%d", H, W);
  if (N == 0) return TF_OK;
  TF_REQUIRE((long long)N * Ho * Wo < (1LL << 31) && (long long)R * S * (C1 + C2) < (1LL << 31), "tf_conv2d_f16: problem too large for 32-bit indexing");
  GemmP p = {};
  p.x = (const half_t*)x; p.x2 = (const half_t*)x2; p.w = (const half_t*)w; p.y = (half_t*)y;
  p.bias = (const half_t*)bias; p.bias_nc = (const half_t*)bias_nc; p.residual = (const half_t*)residual;
  p.bias_nc_stride = bias_nc_stride;
  TF_REQUIRE(bias_nc_stride % 4 == 0 || Cout % 4 != 0, "tf_conv2d_f16: bias_nc_stride must be a multiple of 4");
  p.M = N * Ho * Wo; p.N = Cout; p.C1 = C1; p.C2 = C2; p.C = C1 + C2; p.Kc = R * S * p.C; p.K = p.Kc + C3 + C4;
  p.x3 = (const half_t*)x3; p.x4 = (const half_t*)x4; p.C3 = C3; p.C4 = C4;
  p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.HoWo = Ho * Wo; p.S = S; p.stride = stride; p.pad = pad; p.ups = ups; p.act = 0;
  {
    long long xb = (long long)N * H * W * C1 * 2, x2b = (long long)N * H * W * C2 * 2, wb = (long long)Cout * p.K * 2;
    TF_REQUIRE(xb < (1LL << 31) && x2b < (1LL << 31) && wb < (1LL << 31), "tf_conv2d_f16: tensors must be < 2 GiB each");
    p.x_bytes = (unsigned)xb; p.x2_bytes = C2 ? (unsigned)x2b : (unsigned)xb; p.w_bytes = (unsigned)wb;
    long long x3b = (long long)N * H * W * C3 * 2, x4b = (long long)N * H * W * C4 * 2;
    TF_REQUIRE(x3b < (1LL << 31) && x4b < (1LL << 31), "tf_conv2d_fused_f16: tensors must be < 2 GiB each");
    p.x3_bytes = C3 ? (unsigned)x3b : (unsigned)xb; p.x4_bytes = C4 ? (unsigned)x4b : (unsigned)xb;
  }
  if (gn_partial) {
    TF_REQUIRE(gn_chunks, "tf_conv2d_fused_f16: gn_chunks must not be NULL");
    TF_REQUIRE(gn_groups >= 1 && Cout % gn_groups == 0, "tf_conv2d_fused_f16: Cout=%d not divisible by groups=%d", Cout, gn_groups);
    TF_REQUIRE(gn_partial_bytes >= tf_conv2d_gn_partial_bytes(N, gn_groups), "tf_conv2d_fused_f16: statistics buffer too small (%zu bytes)", gn_partial_bytes);
    int cpg = Cout / gn_groups;
    // group width the epilogue can fold (a group spans at most two n-tiles, one lane per group of a tile); anything else
    // simply reports chunks = 0 and the caller runs tf_group_norm_f16 as usual
    if (cpg >= 4 && cpg <= 64 && Cout % 8 == 0 && Cout <= 4096 && gn_groups <= 256) {
      p.gn_part = gn_partial; p.gn_G = gn_groups; p.gn_cpg = cpg;
    }
  }
  if (gi && gi->bf16) p.bf16 = 1;
  if (gi && gi->on_z && p.gn_part) {                       // (groups the epilogue cannot fold: no statistics, no apply -- *z_written stays 0)
    p.on_z = gi->on_z; p.on_gamma = gi->on_gamma; p.on_beta = gi->on_beta; p.on_eps = gi->on_eps; p.on_silu = gi->on_silu; p.on_applied = gi->on_applied;
  }
  if (gi && gi->gi_part) {
    p.gi_part = gi->gi_part; p.gi_part2 = gi->gi_part2; p.gi_gamma = gi->gi_gamma; p.gi_beta = gi->gi_beta;
    p.gi_chunks = gi->gi_chunks; p.gi_chunks2 = gi->gi_chunks2; p.gi_G = gi->gi_G; p.gi_G1 = gi->gi_G1; p.gi_G2 = gi->gi_G2; p.gi_mr = gi->gi_mr;
    p.gi_silu = gi->gi_silu; p.gi_eps = gi->gi_eps;
    p.ktiles = (p.K + 63) / 64;
    if (!gi_any_ok(p)) { tf_set_error("tf_conv2d_gn_f16: this geometry cannot carry the input GroupNorm (ask tf_conv2d_gn_supported first)"); return TF_E_UNSUPPORTED; }
  }
  return run_gemm(p, workspace, workspace_bytes, g_force_bm, g_force_bn, g_force_split, tf_hs(s), gn_chunks);
}

#define TF_REQUIRE_DTYPE(fn) TF_REQUIRE(dtype == TF_DTYPE_F16 || dtype == TF_DTYPE_BF16, fn ": dtype=%d (0 = float16, 1 = bfloat16)", dtype)
int tf_conv2d_fused_norm_16(int dtype, void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                            const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                            void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                            size_t gn_partial_bytes, int gn_groups, int* gn_chunks, void* z, const void* z_gamma, const void* z_beta, float z_eps,
                            int z_silu, int* z_written, tfStream_t s) {
  TF_REQUIRE_DTYPE("tf_conv2d_fused_norm_16");
  TF_REQUIRE(gn_partial && gn_chunks && z && z_written, "tf_conv2d_fused_norm_f16: gn_partial, gn_chunks, z and z_written must be given");
  TF_REQUIRE((z_gamma == nullptr) == (z_beta == nullptr), "tf_conv2d_fused_norm_f16: gamma and beta must both be given or both NULL");
  *z_written = 0;
  GemmP ex = {};
  ex.bf16 = dtype == TF_DTYPE_BF16;
  ex.on_z = (half_t*)z; ex.on_gamma = (const half_t*)z_gamma; ex.on_beta = (const half_t*)z_beta; ex.on_eps = z_eps; ex.on_silu = z_silu ? 1 : 0;
  ex.on_applied = z_written;
  return conv2d_impl(y, x, x2, w, bias, bias_nc, bias_nc_stride, residual, N, H, W, C1, C2, Cout, R, S, stride, pad, upsample, workspace,
                     workspace_bytes, (float*)gn_partial, gn_partial_bytes, gn_groups, gn_chunks, x3, x4, C3, C4, s, &ex);
}
int tf_conv2d_fused_norm_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                             const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                             void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                             size_t gn_partial_bytes, int gn_groups, int* gn_chunks, void* z, const void* z_gamma, const void* z_beta, float z_eps,
                             int z_silu, int* z_written, tfStream_t s) {
  return tf_conv2d_fused_norm_16(TF_DTYPE_F16, y, x, x2, w, bias, bias_nc, bias_nc_stride, residual, N, H, W, C1, C2, Cout, R, S, stride, pad, upsample, workspace, workspace_bytes,
                                 x3, x4, C3, C4, gn_partial, gn_partial_bytes, gn_groups, gn_chunks, z, z_gamma, z_beta, z_eps, z_silu, z_written, s);
}

int tf_conv2d_gn_supported(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample, int C3, int C4, int in_groups) {
  int Ho, Wo;
  if (N < 1 || C1 < 1 || C2 < 0 || Cout < 1 || R != S || stride < 1 || in_groups < 1 || (C1 + C2) % in_groups) return 0;
  if (conv_geometry(H, W, R, S, stride, pad, upsample ? 1 : 0, &Ho, &Wo)) return 0;
  static float dummy;
  GemmP p = {};
  p.M = N * Ho * Wo; p.N = Cout; p.C1 = C1; p.C2 = C2; p.C = C1 + C2; p.Kc = R * S * p.C; p.K = p.Kc + C3 + C4; p.C3 = C3; p.C4 = C4;
  p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.HoWo = Ho * Wo; p.S = S; p.stride = stride; p.pad = pad; p.ups = upsample ? 1 : 0;
  p.gi_part = &dummy; p.gi_G = in_groups;
  return gi_any_ok(p) ? 1 : 0;
}

int tf_conv2d_gn_16(int dtype, void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                    const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                    void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                    size_t gn_partial_bytes, int gn_groups, int* gn_chunks, const void* in_gamma, const void* in_beta, const void* in_partial,
                    int in_chunks, int in_groups1, const void* in_partial2, int in_chunks2, int in_groups2, int in_groups, float in_eps, int in_silu,
                    tfStream_t s) {
  TF_REQUIRE_DTYPE("tf_conv2d_gn_16");
  TF_REQUIRE(!gn_partial || gn_chunks, "tf_conv2d_gn_f16: gn_chunks must be given with gn_partial");
  TF_REQUIRE(in_partial && in_chunks >= 1 && in_chunks <= 4096 && in_groups >= 1 && (C1 + C2) % in_groups == 0, "tf_conv2d_gn_f16: input statistics missing (chunks=%d groups=%d)", in_chunks, in_groups);
  TF_REQUIRE((in_gamma == nullptr) == (in_beta == nullptr), "tf_conv2d_gn_f16: gamma and beta must both be given or both NULL");
  GemmP gi = {};
  gi.bf16 = dtype == TF_DTYPE_BF16;
  gi.gi_part = (const float*)in_partial; gi.gi_gamma = (const half_t*)in_gamma; gi.gi_beta = (const half_t*)in_beta;
  gi.gi_chunks = in_chunks; gi.gi_G = in_groups; gi.gi_G1 = in_groups; gi.gi_mr = 1; gi.gi_eps = in_eps; gi.gi_silu = in_silu ? 1 : 0;
  if (in_partial2) {
    // concat (x, x2) whose statistics came with its two sources: partials of G1 sub-groups of x and G2 of x2, all of one width,
    // mr adjacent sub-groups of the list [x's | x2's] form a group of the concat (tf_group_norm_apply_cat_f16's contract)
    TF_REQUIRE(C2 > 0 && in_groups1 >= 1 && in_groups2 >= 1 && C1 % in_groups1 == 0 && C2 % in_groups2 == 0 && in_chunks2 >= 1 && in_chunks2 <= 4096,
               "tf_conv2d_gn_f16: C1=%d C2=%d groups1=%d groups2=%d chunks2=%d", C1, C2, in_groups1, in_groups2, in_chunks2);
    const int sub = C1 / in_groups1, cpg = (C1 + C2) / in_groups;
    TF_REQUIRE(C2 / in_groups2 == sub && cpg % sub == 0 && cpg / sub <= 8, "tf_conv2d_gn_f16: the partials' sub-groups (%d and %d channels) do not tile the %d-channel groups", sub, C2 / in_groups2, cpg);
    gi.gi_part2 = (const float*)in_partial2; gi.gi_chunks2 = in_chunks2; gi.gi_G1 = in_groups1; gi.gi_G2 = in_groups2; gi.gi_mr = cpg / sub;
  }
  return conv2d_impl(y, x, x2, w, bias, bias_nc, bias_nc_stride, residual, N, H, W, C1, C2, Cout, R, S, stride, pad, upsample, workspace,
                     workspace_bytes, (float*)gn_partial, gn_partial_bytes, gn_groups, gn_chunks, x3, x4, C3, C4, s, &gi);
}
int tf_conv2d_gn_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                     const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                     void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                     size_t gn_partial_bytes, int gn_groups, int* gn_chunks, const void* in_gamma, const void* in_beta, const void* in_partial,
                     int in_chunks, int in_groups1, const void* in_partial2, int in_chunks2, int in_groups2, int in_groups, float in_eps, int in_silu,
                     tfStream_t s) {
  return tf_conv2d_gn_16(TF_DTYPE_F16, y, x, x2, w, bias, bias_nc, bias_nc_stride, residual, N, H, W, C1, C2, Cout, R, S, stride, pad, upsample, workspace, workspace_bytes, x3, x4, C3, C4,
                         gn_partial, gn_partial_bytes, gn_groups, gn_chunks, in_gamma, in_beta, in_partial, in_chunks, in_groups1, in_partial2, in_chunks2, in_groups2, in_groups, in_eps,
                         in_silu, s);
}

int tf_conv2d_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                  const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                  void* workspace, size_t workspace_bytes, tfStream_t s) {
  return conv2d_impl(y, x, x2, w, bias, bias_nc, bias_nc_stride, residual, N, H, W, C1, C2, Cout, R, S, stride, pad, upsample, workspace,
                     workspace_bytes, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, s);
}

int tf_conv2d_fused_16(int dtype, void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                       const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                       void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                       size_t gn_partial_bytes, int gn_groups, int* gn_chunks, tfStream_t s) {
  TF_REQUIRE_DTYPE("tf_conv2d_fused_16");
  TF_REQUIRE(!gn_partial || gn_chunks, "tf_conv2d_fused_f16: gn_chunks must be given with gn_partial");
  GemmP ex = {};
  ex.bf16 = dtype == TF_DTYPE_BF16;
  return conv2d_impl(y, x, x2, w, bias, bias_nc, bias_nc_stride, residual, N, H, W, C1, C2, Cout, R, S, stride, pad, upsample, workspace,
                     workspace_bytes, (float*)gn_partial, gn_partial_bytes, gn_groups, gn_chunks, x3, x4, C3, C4, s, &ex);
}
int tf_conv2d_fused_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                        const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                        void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                        size_t gn_partial_bytes, int gn_groups, int* gn_chunks, tfStream_t s) {
  return tf_conv2d_fused_16(TF_DTYPE_F16, y, x, x2, w, bias, bias_nc, bias_nc_stride, residual, N, H, W, C1, C2, Cout, R, S, stride, pad, upsample, workspace, workspace_bytes, x3, x4, C3, C4,
                            gn_partial, gn_partial_bytes, gn_groups, gn_chunks, s);
}

size_t tf_conv2d_fused_workspace(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample, int C3, int C4) {
  int Ho, Wo;
  if (stride < 1 || conv_geometry(H, W, R, S, stride, pad, upsample ? 1 : 0, &Ho, &Wo)) return 0;
  return gemm_workspace(N * Ho * Wo, Cout, R * S * (C1 + C2) + C3 + C4, 0);
}

size_t tf_linear_workspace(int M, int N, int K, int act) { return gemm_workspace(M, act == 1 ? 2 * N : N, K, act); }

int tf_linear_f16(void* y, const void* x, const void* w, const void* bias, const void* residual, int M, int N, int K, int act,
                  void* workspace, size_t workspace_bytes, tfStream_t s) {
  return tf_linear_16(TF_DTYPE_F16, y, x, w, bias, residual, M, N, K, act, workspace, workspace_bytes, s);
}
int tf_linear_16(int dtype, void* y, const void* x, const void* w, const void* bias, const void* residual, int M, int N, int K, int act,
                 void* workspace, size_t workspace_bytes, tfStream_t s) {
  TF_REQUIRE_DTYPE("tf_linear_16");
  TF_REQUIRE(y && x && w, "tf_linear_f16: null tensor");
  TF_REQUIRE(M >= 0 && N >= 1 && K >= 8 && K % 8 == 0, "tf_linear_f16: K=%d must be a positive multiple of 8", K);
  TF_REQUIRE(act == 0 || act == 1, "tf_linear_f16: act=%d", act);
  TF_REQUIRE(act == 0 || (bias && N % 16 == 0), "tf_linear_f16: GEGLU needs a bias and N %% 16 == 0 (N=%d)", N);
  if (M == 0) return TF_OK;
  GemmP p = {};
  p.x = (const half_t*)x; p.w = (const half_t*)w; p.y = (half_t*)y; p.bias = (const half_t*)bias; p.residual = (const half_t*)residual;
  p.M = M; p.N = act == 1 ? 2 * N : N; p.K = K; p.Kc = K; p.C1 = K; p.C2 = 0; p.C = K;
  p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.HoWo = M; p.S = 1; p.stride = 1; p.pad = 0; p.ups = 0; p.act = act;
  p.bf16 = dtype == TF_DTYPE_BF16;
  {
    long long xb = (long long)M * K * 2, wb = (long long)p.N * K * 2;
    TF_REQUIRE(xb < (1LL << 31) && wb < (1LL << 31), "tf_linear_f16: tensors must be < 2 GiB each");
    p.x_bytes = (unsigned)xb; p.x2_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
  }
  return run_gemm(p, workspace, workspace_bytes, g_force_bm, g_force_bn, g_force_split, tf_hs(s));
}

// scores of the unfused attention path (attention/sdpa.py:66 of the reference: cp.matmul(q, k^T) in fp32): y32[m, n] = sum_k x[m, k] w[n, k],
// fp16 operands, fp32 accumulators stored as they are
int tf_linear_f32out_f16(void* y_f32, const void* x, const void* w, int M, int N, int K, tfStream_t s) {
  TF_REQUIRE(y_f32 && x && w, "tf_linear_f32out_f16: null tensor");
  TF_REQUIRE(M >= 0 && N >= 1 && K >= 8 && K % 8 == 0, "tf_linear_f32out_f16: K=%d must be a positive multiple of 8", K);
  if (M == 0) return TF_OK;
  GemmP p = {};
  p.x = (const half_t*)x; p.w = (const half_t*)w; p.y = (half_t*)y_f32; p.out32 = (float*)y_f32;
  p.M = M; p.N = N; p.K = K; p.Kc = K; p.C1 = K; p.C2 = 0; p.C = K;
  p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.HoWo = M; p.S = 1; p.stride = 1; p.pad = 0; p.ups = 0; p.act = 0;
  {
    long long xb = (long long)M * K * 2, wb = (long long)N * K * 2;
    TF_REQUIRE(xb < (1LL << 31) && wb < (1LL << 31), "tf_linear_f32out_f16: tensors must be < 2 GiB each");
    p.x_bytes = (unsigned)xb; p.x2_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb; p.x3_bytes = p.x4_bytes = (unsigned)xb;
  }
  return run_gemm(p, nullptr, 0, g_force_bm, g_force_bn, g_force_split > 1 ? 1 : g_force_split, tf_hs(s));
}

// ---- bfloat16 entries: the reference's op tests parametrise bfloat16 next to float16 (tests/linear.py:13, tests/layer_norm.py:13,
// tests/group_norm.py:12) -- same tensors and semantics as the _f16 entries with every 16-bit tensor holding bfloat16 ----------------
int tf_linear_bf16(void* y, const void* x, const void* w, const void* bias, const void* residual, int M, int N, int K, tfStream_t s) {
  TF_REQUIRE(y && x && w, "tf_linear_bf16: null tensor");
  TF_REQUIRE(M >= 0 && N >= 1 && K >= 8 && K % 8 == 0, "tf_linear_bf16: K=%d must be a positive multiple of 8", K);
  if (M == 0) return TF_OK;
  GemmP p = {};
  p.x = (const half_t*)x; p.w = (const half_t*)w; p.y = (half_t*)y; p.bias = (const half_t*)bias; p.residual = (const half_t*)residual;
  p.M = M; p.N = N; p.K = K; p.Kc = K; p.C1 = K; p.C2 = 0; p.C = K;
  p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.HoWo = M; p.S = 1; p.stride = 1; p.pad = 0; p.ups = 0; p.act = 0; p.bf16 = 1;
  {
    long long xb = (long long)M * K * 2, wb = (long long)p.N * K * 2;
    TF_REQUIRE(xb < (1LL << 31) && wb < (1LL << 31), "tf_linear_bf16: tensors must be < 2 GiB each");
    p.x_bytes = (unsigned)xb; p.x2_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
  }
  return run_gemm(p, nullptr, 0, 0, 0, 0, tf_hs(s));
}
// ... with the GEGLU epilogue (act = 1: w / bias packed in 16-row value | gate blocks as for tf_linear_f16; N = the output width): ff/nn.py:5-12 on bfloat16
int tf_linear_act_bf16(void* y, const void* x, const void* w, const void* bias, const void* residual, int M, int N, int K, int act, tfStream_t s) {
  TF_REQUIRE(y && x && w, "tf_linear_act_bf16: null tensor");
  TF_REQUIRE(M >= 0 && N >= 1 && K >= 8 && K % 8 == 0, "tf_linear_act_bf16: K=%d must be a positive multiple of 8", K);
  TF_REQUIRE(act == 0 || (act == 1 && bias && N % 16 == 0), "tf_linear_act_bf16: act=%d (GEGLU needs a bias and N %% 16 == 0, N=%d)", act, N);
  if (M == 0) return TF_OK;
  GemmP p = {};
  p.x = (const half_t*)x; p.w = (const half_t*)w; p.y = (half_t*)y; p.bias = (const half_t*)bias; p.residual = (const half_t*)residual;
  p.M = M; p.N = act == 1 ? 2 * N : N; p.K = K; p.Kc = K; p.C1 = K; p.C2 = 0; p.C = K;
  p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.HoWo = M; p.S = 1; p.stride = 1; p.pad = 0; p.ups = 0; p.act = act; p.bf16 = 1;
  {
    long long xb = (long long)M * K * 2, wb = (long long)p.N * K * 2;
    TF_REQUIRE(xb < (1LL << 31) && wb < (1LL << 31), "tf_linear_act_bf16: tensors must be < 2 GiB each");
    p.x_bytes = (unsigned)xb; p.x2_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
  }
  return run_gemm(p, nullptr, 0, 0, 0, 0, tf_hs(s));
}
int tf_conv2d_bf16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                   const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample, tfStream_t s) {
  GemmP ex = {};
  ex.bf16 = 1;
  return conv2d_impl(y, x, x2, w, bias, bias_nc, bias_nc_stride, residual, N, H, W, C1, C2, Cout, R, S, stride, pad, upsample, nullptr, 0,
                     nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, s, &ex);
}

// ---- fp8 entries (config 5) ---------------------------------------------------------------------------------------------------
int tf_quantize_fp8_f16(void* y8, const void* x, long long n, float scale, tfStream_t s) {
  TF_REQUIRE(y8 && x && n >= 0 && n % 8 == 0, "tf_quantize_fp8_f16: n=%lld must be a multiple of 8", n);
  if (n == 0) return TF_OK;
  long long n8 = n / 8, grid = (n8 + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_quantize_fp8, dim3((unsigned)grid), dim3(256), 0, tf_hs(s), (unsigned char*)y8, (const half_t*)x, scale, n8);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_pack_weight_fp8(void* w8, void* scale_f32, const void* w, int N, int K, tfStream_t s) {
  TF_REQUIRE(w8 && scale_f32 && w && N >= 1 && K >= 8 && K % 8 == 0, "tf_pack_weight_fp8: N=%d K=%d (K must be a multiple of 8)", N, K);
  hipLaunchKernelGGL(k_pack_weight_fp8, dim3(ceil_div(N, 4)), dim3(256), 0, tf_hs(s), (unsigned char*)w8, (float*)scale_f32, (const half_t*)w, N, K);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
size_t tf_conv2d_fp8_workspace(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample) {
  return tf_conv2d_workspace(N, H, W, C1, C2, Cout, R, S, stride, pad, upsample);
}
int tf_conv2d_fp8(void* y, const void* x8, const void* x28, const void* w8, const void* wscale, const void* bias, const void* bias_nc,
                  long long bias_nc_stride, const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad,
                  int upsample, void* workspace, size_t workspace_bytes, void* gn_partial, size_t gn_partial_bytes, int gn_groups, int* gn_chunks,
                  tfStream_t s) {
  if (gn_chunks) *gn_chunks = 0;
  TF_REQUIRE(y && x8 && w8 && wscale, "tf_conv2d_fp8: null tensor");
  TF_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x28) && C1 % 64 == 0 && C2 % 64 == 0, "tf_conv2d_fp8: channel counts must be multiples of 64 (C1=%d C2=%d)", C1, C2);
  TF_REQUIRE(R >= 1 && R == S && stride >= 1 && pad >= 0 && Cout >= 1 && N >= 0, "tf_conv2d_fp8: bad geometry R=%d S=%d stride=%d pad=%d", R, S, stride, pad);
  TF_REQUIRE(!gn_partial || gn_chunks, "tf_conv2d_fp8: gn_chunks must be given with gn_partial");
  int ups = upsample ? 1 : 0, Ho, Wo;
  TF_REQUIRE(!conv_geometry(H, W, R, S, stride, pad, ups, &Ho, &Wo), "tf_conv2d_fp8: empty output for H=%d W=%d", H, W);
  if (N == 0) return TF_OK;
  TF_REQUIRE((long long)N * Ho * Wo < (1LL << 31), "tf_conv2d_fp8: problem too large for 32-bit indexing");
  GemmP p = {};
  p.fp8 = 1; p.wscale = (const float*)wscale;
  p.x = (const half_t*)x8; p.x2 = (const half_t*)x28; p.w = (const half_t*)w8; p.y = (half_t*)y;
  p.bias = (const half_t*)bias; p.bias_nc = (const half_t*)bias_nc; p.residual = (const half_t*)residual; p.bias_nc_stride = bias_nc_stride;
  TF_REQUIRE(bias_nc_stride % 4 == 0 || Cout % 4 != 0, "tf_conv2d_fp8: bias_nc_stride must be a multiple of 4");
  p.M = N * Ho * Wo; p.N = Cout; p.C1 = C1; p.C2 = C2; p.C = C1 + C2; p.Kc = R * S * p.C; p.K = p.Kc;
  p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.HoWo = Ho * Wo; p.S = S; p.stride = stride; p.pad = pad; p.ups = ups;
  {
    long long xb = (long long)N * H * W * C1, x2b = (long long)N * H * W * C2, wb = (long long)Cout * p.K;
    TF_REQUIRE(xb < (1LL << 31) && x2b < (1LL << 31) && wb < (1LL << 31), "tf_conv2d_fp8: tensors must be < 2 GiB each");
    p.x_bytes = (unsigned)xb; p.x2_bytes = C2 ? (unsigned)x2b : (unsigned)xb; p.w_bytes = (unsigned)wb;
    p.x3_bytes = p.x4_bytes = (unsigned)xb;
  }
  if (gn_partial) {
    TF_REQUIRE(gn_groups >= 1 && Cout % gn_groups == 0, "tf_conv2d_fp8: Cout=%d not divisible by groups=%d", Cout, gn_groups);
    TF_REQUIRE(gn_partial_bytes >= tf_conv2d_gn_partial_bytes(N, gn_groups), "tf_conv2d_fp8: statistics buffer too small (%zu bytes)", gn_partial_bytes);
    int cpg = Cout / gn_groups;
    if (cpg >= 4 && cpg <= 64 && Cout % 8 == 0 && Cout <= 4096 && gn_groups <= 256) { p.gn_part = (float*)gn_partial; p.gn_G = gn_groups; p.gn_cpg = cpg; }
  }
  return run_gemm(p, workspace, workspace_bytes, g_force_bm, g_force_bn, g_force_split, tf_hs(s), gn_chunks);
}
int tf_linear_fp8(void* y, const void* x8, const void* w8, const void* wscale, const void* bias, const void* residual, int M, int N, int K, int act,
                  int out_fp8, void* workspace, size_t workspace_bytes, tfStream_t s) {
  TF_REQUIRE(y && x8 && w8 && wscale, "tf_linear_fp8: null tensor");
  TF_REQUIRE(M >= 0 && N >= 1 && K >= 64 && K % 64 == 0, "tf_linear_fp8: K=%d must be a positive multiple of 64", K);
  TF_REQUIRE(act == 0 || act == 1, "tf_linear_fp8: act=%d", act);
  TF_REQUIRE(act == 0 || (bias && N % 16 == 0), "tf_linear_fp8: GEGLU needs a bias and N %% 16 == 0 (N=%d)", N);
  TF_REQUIRE(!out_fp8 || N % 8 == 0, "tf_linear_fp8: an e4m3 output needs N %% 8 == 0 (N=%d)", N);
  if (M == 0) return TF_OK;
  GemmP p = {};
  p.fp8 = 1; p.wscale = (const float*)wscale; p.out8 = out_fp8 ? 1 : 0;
  p.x = (const half_t*)x8; p.w = (const half_t*)w8; p.y = (half_t*)y; p.bias = (const half_t*)bias; p.residual = (const half_t*)residual;
  p.M = M; p.N = act == 1 ? 2 * N : N; p.K = K; p.Kc = K; p.C1 = K; p.C2 = 0; p.C = K;
  p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.HoWo = M; p.S = 1; p.stride = 1; p.pad = 0; p.ups = 0; p.act = act;
  {
    long long xb = (long long)M * K, wb = (long long)p.N * K;
    TF_REQUIRE(xb < (1LL << 31) && wb < (1LL << 31), "tf_linear_fp8: tensors must be < 2 GiB each");
    p.x_bytes = (unsigned)xb; p.x2_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb; p.x3_bytes = p.x4_bytes = (unsigned)xb;
  }
  if (p.out8) workspace = nullptr, workspace_bytes = 0;   // the split-K reduce writes fp16: an e4m3 output runs unsplit
  return run_gemm(p, workspace, workspace_bytes, g_force_bm, g_force_bn, g_force_split, tf_hs(s));
}

// ---- block-scaled e4m3 activations (round 4): one E8M0 scale per 32 consecutive channels of a pixel / token, fed to the scale operand of
// v_mfma_scale_f32_16x16x128_f8f6f4.  An "mx8" tensor is ONE buffer: rows x C codes, then rows x C/32 scale bytes (tf_mx8_bytes).
size_t tf_mx8_bytes(long long rows, int C) { return rows > 0 && C > 0 ? (size_t)rows * C + (size_t)rows * (C / 32) : 0; }
int tf_quantize_mx8_f16(void* y_mx, const void* x, long long rows, int C, tfStream_t s) {
  TF_REQUIRE(y_mx && x && rows >= 0 && C > 0 && C % 32 == 0, "tf_quantize_mx8_f16: C=%d must be a positive multiple of 32", C);
  if (rows == 0) return TF_OK;
  long long n8 = rows * (C / 8), grid = (n8 + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_quantize_mx8, dim3((unsigned)grid), dim3(256), 0, tf_hs(s), (unsigned char*)y_mx, (const half_t*)x, rows, C);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
// does the block-scaled kernel take a GEMM of M rows (pixels / tokens) x N outputs x K?  It is the ping-pong kernel: launches that fill the chip
// with its 192- / 256-row tiles (BASELINE config 5's regime); stride-1 convolutions without up-sampling and linears; callers keep fp16 otherwise
static bool mx_shape_ok(const GemmP& p) {
  if (p.M <= 256 || p.K % 64 || p.N % 8) return false;
  TunedCfg d = mx_default(p);
  if (!d.c.bm) return false;
  static const int cand[][2] = {{192, 160}, {192, 128}, {256, 128}, {256, 160}};
  for (int ci = 0; ci < 4; ++ci)
    if (pp_ok(p, cand[ci][1], cand[ci][0]) && (long long)((p.M + cand[ci][0] - 1) / cand[ci][0]) * ((p.N + cand[ci][1] - 1) / cand[ci][1]) >= 128) return true;
  return false;
}
int tf_mx8_gemm_supported(int M, int N, int K, int act, int out_mx) {
  if (M < 1 || N < 1 || K < 64) return 0;
  GemmP p = {};
  p.fp8 = 1; p.mx = 1; p.out8 = out_mx ? 1 : 0; p.act = act;
  p.M = M; p.N = act == 1 ? 2 * N : N; p.K = K; p.Kc = K; p.C1 = K; p.C = K; p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.HoWo = M; p.S = 1; p.stride = 1;
  return mx_shape_ok(p) ? 1 : 0;
}
int tf_mx8_conv_supported(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample) {
  int Ho, Wo;
  if (N < 1 || C1 < 64 || C2 < 0 || R != S || stride != 1 || upsample || (C1 % 64) || (C2 % 64) || conv_geometry(H, W, R, S, stride, pad, 0, &Ho, &Wo)) return 0;
  GemmP p = {};
  p.fp8 = 1; p.mx = 1;
  p.M = N * Ho * Wo; p.N = Cout; p.C1 = C1; p.C2 = C2; p.C = C1 + C2; p.Kc = R * S * p.C; p.K = p.Kc;
  p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.HoWo = Ho * Wo; p.S = S; p.stride = 1; p.pad = pad;
  p.bias_nc = (const half_t*)&p;                            // (the ResBlock convs carry a time-embedding bias: the stricter tile rule)
  return mx_shape_ok(p) ? 1 : 0;
}
int tf_conv2d_mx8(void* y, const void* x_mx, const void* x2_mx, const void* w8, const void* wscale, const void* bias, const void* bias_nc,
                  long long bias_nc_stride, const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad,
                  void* workspace, size_t workspace_bytes, void* gn_partial, size_t gn_partial_bytes, int gn_groups, int* gn_chunks, tfStream_t s) {
  if (gn_chunks) *gn_chunks = 0;
  TF_REQUIRE(y && x_mx && w8 && wscale, "tf_conv2d_mx8: null tensor");
  TF_REQUIRE(C1 > 0 && C2 >= 0 && (C2 == 0 || x2_mx) && C1 % 64 == 0 && C2 % 64 == 0, "tf_conv2d_mx8: channel counts must be multiples of 64 (C1=%d C2=%d)", C1, C2);
  TF_REQUIRE(R >= 1 && R == S && stride == 1 && pad >= 0 && Cout >= 1 && N >= 0, "tf_conv2d_mx8: stride-1 square filters only (R=%d S=%d stride=%d pad=%d)", R, S, stride, pad);
  TF_REQUIRE(!gn_partial || gn_chunks, "tf_conv2d_mx8: gn_chunks must be given with gn_partial");
  int Ho, Wo;
  TF_REQUIRE(!conv_geometry(H, W, R, S, stride, pad, 0, &Ho, &Wo), "tf_conv2d_mx8: empty output for H=%d W=%d", H, W);
  if (N == 0) return TF_OK;
  TF_REQUIRE((long long)N * Ho * Wo < (1LL << 31), "tf_conv2d_mx8: problem too large for 32-bit indexing");
  GemmP p = {};
  p.fp8 = 1; p.mx = 1; p.wscale = (const float*)wscale;
  p.x = (const half_t*)x_mx; p.x2 = (const half_t*)x2_mx; p.w = (const half_t*)w8; p.y = (half_t*)y;
  p.bias = (const half_t*)bias; p.bias_nc = (const half_t*)bias_nc; p.residual = (const half_t*)residual; p.bias_nc_stride = bias_nc_stride;
  TF_REQUIRE(bias_nc_stride % 4 == 0 || Cout % 4 != 0, "tf_conv2d_mx8: bias_nc_stride must be a multiple of 4");
  p.M = N * Ho * Wo; p.N = Cout; p.C1 = C1; p.C2 = C2; p.C = C1 + C2; p.Kc = R * S * p.C; p.K = p.Kc;
  p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.HoWo = Ho * Wo; p.S = S; p.stride = stride; p.pad = pad; p.ups = 0;
  {
    long long xb = (long long)N * H * W * C1, x2b = (long long)N * H * W * C2, wb = (long long)Cout * p.K;
    TF_REQUIRE(xb + xb / 32 < (1LL << 31) && x2b + x2b / 32 < (1LL << 31) && wb < (1LL << 31), "tf_conv2d_mx8: tensors must be < 2 GiB each");
    p.x_bytes = (unsigned)xb; p.x2_bytes = C2 ? (unsigned)x2b : (unsigned)xb; p.w_bytes = (unsigned)wb;     // (the codes; the scale bytes sit behind them)
    p.x3_bytes = p.x4_bytes = (unsigned)xb;
  }
  if (gn_partial) {
    TF_REQUIRE(gn_groups >= 1 && Cout % gn_groups == 0, "tf_conv2d_mx8: Cout=%d not divisible by groups=%d", Cout, gn_groups);
    TF_REQUIRE(gn_partial_bytes >= tf_conv2d_gn_partial_bytes(N, gn_groups), "tf_conv2d_mx8: statistics buffer too small (%zu bytes)", gn_partial_bytes);
    int cpg = Cout / gn_groups;
    if (cpg >= 4 && cpg <= 64 && Cout % 8 == 0 && Cout <= 4096 && gn_groups <= 256) { p.gn_part = (float*)gn_partial; p.gn_G = gn_groups; p.gn_cpg = cpg; }
  }
  return run_gemm(p, workspace, workspace_bytes, g_force_bm, g_force_bn, g_force_split, tf_hs(s), gn_chunks);
}
int tf_linear_mx8(void* y, const void* x_mx, const void* w8, const void* wscale, const void* bias, const void* residual, int M, int N, int K, int act,
                  int out_mx, void* workspace, size_t workspace_bytes, tfStream_t s) {
  TF_REQUIRE(y && x_mx && w8 && wscale, "tf_linear_mx8: null tensor");
  TF_REQUIRE(M >= 0 && N >= 1 && K >= 64 && K % 64 == 0, "tf_linear_mx8: K=%d must be a positive multiple of 64", K);
  TF_REQUIRE(act == 0 || act == 1, "tf_linear_mx8: act=%d", act);
  TF_REQUIRE(act == 0 || (bias && N % 16 == 0), "tf_linear_mx8: GEGLU needs a bias and N %% 16 == 0 (N=%d)", N);
  TF_REQUIRE(!out_mx || (act == 1 && N % 32 == 0 && !residual), "tf_linear_mx8: a block-scaled output is the GEGLU epilogue's (act = 1, N %% 32 == 0, no residual; N=%d)", N);
  if (M == 0) return TF_OK;
  GemmP p = {};
  p.fp8 = 1; p.mx = 1; p.wscale = (const float*)wscale; p.out8 = out_mx ? 1 : 0;
  p.x = (const half_t*)x_mx; p.w = (const half_t*)w8; p.y = (half_t*)y; p.bias = (const half_t*)bias; p.residual = (const half_t*)residual;
  p.M = M; p.N = act == 1 ? 2 * N : N; p.K = K; p.Kc = K; p.C1 = K; p.C2 = 0; p.C = K;
  p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.HoWo = M; p.S = 1; p.stride = 1; p.pad = 0; p.ups = 0; p.act = act;
  {
    long long xb = (long long)M * K, wb = (long long)p.N * K;
    TF_REQUIRE(xb + xb / 32 < (1LL << 31) && wb < (1LL << 31), "tf_linear_mx8: tensors must be < 2 GiB each");
    p.x_bytes = (unsigned)xb; p.x2_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb; p.x3_bytes = p.x4_bytes = (unsigned)xb;
  }
  if (p.out8) workspace = nullptr, workspace_bytes = 0;   // the split-K reduce writes fp16: an e4m3 output runs unsplit
  return run_gemm(p, workspace, workspace_bytes, g_force_bm, g_force_bn, g_force_split, tf_hs(s));
}

int tf_ln_fold_weights_16(int dtype, void* w_out, void* bias_out, void* colsum_out, const void* w, const void* bias, const void* gamma, const void* beta,
                          int N, int K, tfStream_t s) {
  TF_REQUIRE_DTYPE("tf_ln_fold_weights_16");
  TF_REQUIRE(w_out && bias_out && colsum_out && w && gamma && beta && N >= 1 && K % 8 == 0, "tf_ln_fold_weights_f16: bad arguments (K=%d)", K);
  if (dtype == TF_DTYPE_BF16) hipLaunchKernelGGL(k_ln_fold<true>, dim3(ceil_div(N, 4)), dim3(256), 0, tf_hs(s), (half_t*)w_out, (half_t*)bias_out, (float*)colsum_out, (const half_t*)w,
                                                 (const half_t*)bias, (const half_t*)gamma, (const half_t*)beta, N, K);
  else hipLaunchKernelGGL(k_ln_fold<false>, dim3(ceil_div(N, 4)), dim3(256), 0, tf_hs(s), (half_t*)w_out, (half_t*)bias_out, (float*)colsum_out, (const half_t*)w,
                          (const half_t*)bias, (const half_t*)gamma, (const half_t*)beta, N, K);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_ln_fold_weights_f16(void* w_out, void* bias_out, void* colsum_out, const void* w, const void* bias, const void* gamma, const void* beta,
                           int N, int K, tfStream_t s) {
  return tf_ln_fold_weights_16(TF_DTYPE_F16, w_out, bias_out, colsum_out, w, bias, gamma, beta, N, K, s);
}

int tf_linear_ln_f16(void* y, const void* x, const void* w_folded, const void* bias_folded, const void* colsum, const void* residual, int M, int N,
                     int K, int act, float eps, tfStream_t s) {
  return tf_linear_ln_16(TF_DTYPE_F16, y, x, w_folded, bias_folded, colsum, residual, M, N, K, act, eps, s);
}
int tf_linear_ln_16(int dtype, void* y, const void* x, const void* w_folded, const void* bias_folded, const void* colsum, const void* residual, int M, int N,
                    int K, int act, float eps, tfStream_t s) {
  TF_REQUIRE_DTYPE("tf_linear_ln_16");
  TF_REQUIRE(y && x && w_folded && bias_folded && colsum, "tf_linear_ln_f16: null tensor");
  TF_REQUIRE(M >= 0 && N >= 1 && K >= 64 && K % 64 == 0, "tf_linear_ln_f16: K=%d must be a positive multiple of 64", K);
  TF_REQUIRE(act == 0 || act == 1, "tf_linear_ln_f16: act=%d", act);
  TF_REQUIRE((act == 1 ? N % 16 == 0 : N % 4 == 0), "tf_linear_ln_f16: N=%d must be a multiple of 4 (16 for GEGLU)", N);
  if (M == 0) return TF_OK;
  GemmP p = {};
  p.x = (const half_t*)x; p.w = (const half_t*)w_folded; p.y = (half_t*)y; p.bias = (const half_t*)bias_folded; p.residual = (const half_t*)residual;
  p.ln_colsum = (const float*)colsum; p.ln_eps = eps;
  p.bf16 = dtype == TF_DTYPE_BF16;
  p.M = M; p.N = act == 1 ? 2 * N : N; p.K = K; p.Kc = K; p.C1 = K; p.C2 = 0; p.C = K;
  p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.HoWo = M; p.S = 1; p.stride = 1; p.pad = 0; p.ups = 0; p.act = act;
  {
    long long xb = (long long)M * K * 2, wb = (long long)p.N * K * 2;
    TF_REQUIRE(xb < (1LL << 31) && wb < (1LL << 31), "tf_linear_ln_f16: tensors must be < 2 GiB each");
    p.x_bytes = (unsigned)xb; p.x2_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
  }
  return run_gemm(p, nullptr, 0, g_force_bm, g_force_bn, g_force_split, tf_hs(s));
}

int tf_gemv_16(int dtype, void* y, const void* x, const void* w, const void* bias, int M, int N, int K, int silu_input, tfStream_t s) {
  TF_REQUIRE_DTYPE("tf_gemv_16");
  TF_REQUIRE(y && x && w && M >= 1 && M <= 8 && N >= 1 && K % 8 == 0, "tf_gemv_f16: needs 1 <= M <= 8 (M=%d) and K %% 8 == 0 (K=%d)", M, K);
  if (dtype == TF_DTYPE_BF16) hipLaunchKernelGGL(k_gemv<true>, dim3(ceil_div(N, 4)), dim3(256), 0, tf_hs(s), (half_t*)y, (const half_t*)x, (const half_t*)w, (const half_t*)bias, M, N, K, silu_input);
  else hipLaunchKernelGGL(k_gemv<false>, dim3(ceil_div(N, 4)), dim3(256), 0, tf_hs(s), (half_t*)y, (const half_t*)x, (const half_t*)w, (const half_t*)bias, M, N, K, silu_input);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_gemv_f16(void* y, const void* x, const void* w, const void* bias, int M, int N, int K, int silu_input, tfStream_t s) {
  return tf_gemv_16(TF_DTYPE_F16, y, x, w, bias, M, N, K, silu_input, s);
}

}  // extern "C"

