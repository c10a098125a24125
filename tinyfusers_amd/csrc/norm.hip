// GroupNorm(+affine,+SiLU) on NHWC f16 and LayerNorm over the last dim.  HBM-bound: 16-B loads, wave64
// shuffles, fp32 statistics.  Reference: ff/group_norm.py:3-21, ff/layer_norm.py:8-49.
#include "common.h"
#include "../../include/tinyfusers_hip.h"

#define GN_MAX_CHUNKS 64
// element type of the 16-bit kernels: half_t (fp16, the UNet path) or bf16_t (tf_*_bf16: the reference's norm tests also run bfloat16,
// tests/group_norm.py:12-19, tests/layer_norm.py:13-27); statistics and arithmetic are fp32 either way
template <typename T> struct vec8 { typedef T type __attribute__((ext_vector_type(8))); };

// ---- GroupNorm pass 1: per-(image, pixel-chunk) partial sums per group ---------------------------
// block = CV * RPB threads (CV = C/8 channel vectors, RPB rows per sweep); thread owns one channel vector.
// partial layout: [N][chunks][G][2] fp32 (sum, sum of squares)
template <typename T = half_t>
__global__ void k_gn_stats(float* __restrict__ partial, const T* __restrict__ x, const T* __restrict__ x2, int HW, int C1, int C2,
                           int G, int chunks, int pix_per_chunk, int CV, int RPB) {
  typedef typename vec8<T>::type V8;
  extern __shared__ float red[];  // [RPB*CV][16]: per-thread channel sums, then reduced in a fixed order
  int n = blockIdx.y, chunk = blockIdx.x;
  int C = C1 + C2, cpg = C / G;
  int t = threadIdx.x;
  int cv = t % CV, rr = t / CV;
  int c = cv * 8;
  if (rr < RPB) {
    const T* base;
    int ld;
    if (c < C1) { base = x + (long long)n * HW * C1 + c; ld = C1; }
    else { base = x2 + (long long)n * HW * C2 + (c - C1); ld = C2; }
    float s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; ss[j] = 0.f; }
    int p0 = chunk * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
    for (int p = p0 + rr; p < p1; p += 4 * RPB) {          // four independent loads in flight per thread
      V8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int q = p + u * RPB;
        v[u] = q < p1 ? *reinterpret_cast<const V8*>(base + (long long)q * ld) : (V8){0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) { float f = (float)v[u][j]; s[j] += f; ss[j] += f * f; }
    }
    float* r = red + (long long)t * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) { r[j] = s[j]; r[8 + j] = ss[j]; }
  }
  __syncthreads();
  // deterministic fold: 8 lanes per group walk the (row-thread, channel) pairs of the group in a fixed strided order,
  // then three xor-shuffles
  float* out = partial + ((long long)n * chunks + chunk) * G * 2;
  const int sub = t & 7, gstep = blockDim.x >> 3;
  const int npairs = RPB * cpg;
  for (int g0 = 0; g0 < G; g0 += gstep) {
    int g = g0 + (t >> 3);
    float S = 0.f, SS = 0.f;
    if (g < G && (t >> 3) < gstep) {
      for (int q = sub; q < npairs; q += 8) {
        int r_ = q / cpg, ch = g * cpg + (q - r_ * cpg);
        const float* r = red + ((long long)r_ * CV + (ch >> 3)) * 16;
        S += r[ch & 7]; SS += r[8 + (ch & 7)];
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { S += __shfl_xor(S, o, 64); SS += __shfl_xor(SS, o, 64); }
    if (g < G && sub == 0 && (t >> 3) < gstep) { out[2 * g] = S; out[2 * g + 1] = SS; }
  }
}

// ---- GroupNorm pass 2: y = silu?((x - mean) * rstd * gamma + beta).  Every block first folds the per-chunk
// partials of its image into (mean, rstd) per group (8 lanes per group strided over the chunks + xor-shuffles, fp64,
// fixed order -> every block gets the same bits) -- cheaper than a separate finalize launch; then many small blocks
// stream the tensor (the kernel is latency-bound otherwise).
#define GN_APPLY_PPT 4
template <int OUT8, typename T = half_t>     // OUT8: 0 = 16-bit output, 1 = e4m3 at scale 1, 2 = block-scaled e4m3 (common.h: mx_quant8; codes, then the scale bytes behind the N x HW x C codes)
__global__ void k_gn_apply(T* __restrict__ y, const T* __restrict__ x, const T* __restrict__ x2, const T* __restrict__ gamma,
                           const T* __restrict__ beta, const float* __restrict__ partial, int HW, int C1, int C2, int G, float eps,
                           int do_silu, int chunks, int pix_per_block, int CV, int RPB, const float* __restrict__ partial2, int chunks2,
                           int G1, int G2, int mr, int nbatch) {
  // nbatch (round 5): a block streams nbatch batches of GN_APPLY_PPT * RPB pixels, the loads of batch b + 1 in flight while batch b is normalised and
  // stored -- the statistics fold below (every block re-reads the image's chunks x G partials from L2: 49 KB at config 5's first level) is then paid
  // once per nbatch * 15 KB of tensor instead of once per 15 KB; the host keeps >= 3 blocks per CU (gn_batches)
  typedef typename vec8<T>::type V8;
  extern __shared__ float st[];  // [G][2] : mean, rstd
  int n = blockIdx.y;
  int C = C1 + C2, cpg = C / G;
  int t = threadIdx.x;
  // this thread's slice of the tensor: issue its first loads (and gamma / beta) BEFORE the statistics fold, so the three
  // dependent global round trips (partials -> gamma/beta -> x) of a latency-bound launch collapse into one
  const int cv = t % CV, rr = t / CV;
  const bool active = rr < RPB;
  const int c = cv * 8;
  const T* base = x;
  int ld = C1;
  if (active) {
    if (c < C1) { base = x + (long long)n * HW * C1 + c; ld = C1; }
    else { base = x2 + (long long)n * HW * C2 + (c - C1); ld = C2; }
  }
  const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
  V8 gm, bt, v[GN_APPLY_PPT];                            // pix_per_block == nbatch * GN_APPLY_PPT * RPB (gn_geometry, gn_batches)
  if (active) {
    if (gamma) { gm = *reinterpret_cast<const V8*>(gamma + c); bt = *reinterpret_cast<const V8*>(beta + c); }
#pragma unroll
    for (int i = 0; i < GN_APPLY_PPT; ++i) {
      int p = p0 + rr + i * RPB;
      if (p < p1) v[i] = *reinterpret_cast<const V8*>(base + (long long)p * ld);
    }
  }
  {
    const int sub = t & 7;
    const int gstep = blockDim.x >> 3;               // blockDim.x is a multiple of 8 (CV*RPB threads, padded below)
    for (int g0 = 0; g0 < G; g0 += gstep) {
      int g = g0 + (t >> 3);
      double S = 0.0, SS = 0.0;
      if (g < G) {
        // this lane's chunks k = sub, sub + 8, ...: 8 independent loads in flight per round, summed in k order.
        // partial2 != NULL (tf_group_norm_apply_cat_f16): the statistics came as partials of EACH source, G1 sub-groups for x and
        // G2 for x2, all of the same width; group g of the concat = the mr adjacent sub-groups [mr g, mr g + mr) of the list
        // [x's G1 | x2's G2] (it may straddle the two tables: 1280 + 640 channels in 32 groups = 3 sub-groups of 20 channels)
        const int nsub = partial2 ? mr : 1;
        for (int j = 0; j < nsub; ++j) {
          const float* pp = partial + (long long)n * chunks * G * 2 + g * 2;
          int nch = chunks, gstride = G * 2;
          if (partial2) {
            const int sg = mr * g + j;
            const bool first = sg < G1;
            nch = first ? chunks : chunks2;
            gstride = (first ? G1 : G2) * 2;
            pp = (first ? partial : partial2) + (long long)n * nch * gstride + (first ? sg : sg - G1) * 2;
          }
          for (int k0 = sub; k0 < nch; k0 += 64) {
            f2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              int k = k0 + 8 * u;
              v[u] = k < nch ? *reinterpret_cast<const f2*>(pp + (long long)k * gstride) : (f2){0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { S += (double)v[u][0]; SS += (double)v[u][1]; }
          }
        }
      }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) { S += __shfl_xor(S, o, 64); SS += __shfl_xor(SS, o, 64); }
      if (g < G && sub == 0) {
        double cnt = (double)HW * cpg;
        double mean = S / cnt;
        double var = SS / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        st[2 * g] = (float)mean;
        st[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
      }
    }
  }
  __syncthreads();
  if (OUT8 != 2 && !active) return;                       // (the block-scaled form shuffles across lanes: every lane stays)
  float a[8], b[8];
  if (active) {
    int g = c / cpg, gend = (g + 1) * cpg;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (c + j >= gend) { ++g; gend += cpg; }
      float mean = st[2 * g], rstd = st[2 * g + 1];
      float gmj = gamma ? (float)gm[j] : 1.0f, btj = gamma ? (float)bt[j] : 0.0f;
      a[j] = rstd * gmj;
      b[j] = btj - mean * a[j];
    }
  }
  T* yo = y + (long long)n * HW * C + c;
  const int bstride = GN_APPLY_PPT * RPB;
  for (int bi = 0; bi < nbatch; ++bi) {
  const int pb = p0 + bi * bstride;
  if (pb >= p1) break;                                    // (block-uniform)
  V8 vn[GN_APPLY_PPT];
  if (bi + 1 < nbatch && active) {                        // the next batch's loads fly while this one is normalised and stored
#pragma unroll
    for (int i = 0; i < GN_APPLY_PPT; ++i) {
      int p = pb + bstride + rr + i * RPB;
      if (p < p1) vn[i] = *reinterpret_cast<const V8*>(base + (long long)p * ld);
    }
  }
#pragma unroll
  for (int i = 0; i < GN_APPLY_PPT; ++i) {
    int p = pb + rr + i * RPB;
    const bool live = active && p < p1;
    if (OUT8 == 2 || live) {
      V8 o;
      f4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
      if (live) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float f = (float)v[i][j] * a[j] + b[j];
          f = do_silu ? silu_f(f) : f;
          o[j] = (T)f;
          if (j < 4) q0[j] = f; else q1[j - 4] = f;
        }
      }
      if constexpr (OUT8 == 2) {
        // 4 consecutive lanes = 4 consecutive channel vectors of one pixel = one 32-channel block (CV % 4 == 0, so the quads never straddle rows)
        unsigned sb;
        const uint2 code = mx_quant8(q0, q1, sb);
        if (live) {
          unsigned char* yb = reinterpret_cast<unsigned char*>(y);
          *reinterpret_cast<uint2*>(yb + ((long long)n * HW + p) * C + c) = code;
          if ((cv & 3) == 0) yb[(long long)gridDim.y * HW * C + ((long long)n * HW + p) * (C >> 5) + (c >> 5)] = (unsigned char)sb;
        }
      }
      else if constexpr (OUT8 == 1) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(y) + (long long)n * HW * C + c + (long long)p * C) = pack8_fp8(q0, q1);   // e4m3 operand of an fp8 conv
      else *reinterpret_cast<V8*>(yo + (long long)p * C) = o;
    }
  }
  if (bi + 1 < nbatch) {
#pragma unroll
    for (int i = 0; i < GN_APPLY_PPT; ++i) v[i] = vn[i];
  }
  }
}

// ---- LayerNorm: LPR lanes per row (8/16/32/64 so that a lane holds <= 5 vectors), 64/LPR rows per wave: several
// independent 16-B loads in flight per lane and only log2(LPR) shuffle steps per reduction.
#define LN_MAXV 5
template <int LPR, typename T = half_t>
__global__ void __launch_bounds__(256) k_layer_norm(T* __restrict__ y, const T* __restrict__ x, const T* __restrict__ gamma,
                                                    const T* __restrict__ beta, int rows, int C, float eps, int out8) {
  typedef typename vec8<T>::type V8;
  constexpr int RPW = 64 / LPR;                          // rows per wave
  int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  int row = (blockIdx.x * 4 + w) * RPW + l / LPR;
  int li = l % LPR;
  int CV = C >> 3;
  bool live = row < rows;
  const T* xr = x + (long long)(live ? row : 0) * C;
  V8 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int cv = li + LPR * i;
    if (cv < CV) {
      v[i] = *reinterpret_cast<const V8*>(xr + cv * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[i][j];
    }
  }
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o, 64);
  float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int cv = li + LPR * i;
    if (cv < CV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = (float)v[i][j] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) q += __shfl_xor(q, o, 64);
  float rstd = rsqrtf(q / (float)C + eps);
  if (out8 == 2) {
    // block-scaled e4m3 (common.h: mx_quant8): 4 consecutive lanes hold the 4 channel vectors of one 32-channel block; every lane stays for
    // the shuffles, dead rows / vectors contribute zeros and store nothing.  Codes at y, scale bytes behind the rows x C codes.
    unsigned char* yb = reinterpret_cast<unsigned char*>(y);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int cv = li + LPR * i;
      if (LPR * i >= CV) break;                            // (wave-uniform: no lane has a vector in this round)
      const bool on = live && cv < CV;
      f4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
      if (on) {
        V8 gm, bt;
        if (gamma) { gm = *reinterpret_cast<const V8*>(gamma + cv * 8); bt = *reinterpret_cast<const V8*>(beta + cv * 8); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float f = ((float)v[i][j] - mean) * rstd;
          if (gamma) f = f * (float)gm[j] + (float)bt[j];
          if (j < 4) q0[j] = f; else q1[j - 4] = f;
        }
      }
      unsigned sb;
      const uint2 code = mx_quant8(q0, q1, sb);
      if (on) {
        *reinterpret_cast<uint2*>(yb + (long long)row * C + cv * 8) = code;
        if ((cv & 3) == 0) yb[(long long)rows * C + (long long)row * (C >> 5) + (cv >> 2)] = (unsigned char)sb;
      }
    }
    return;
  }
  if (!live) return;
  T* yr = y + (long long)row * C;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int cv = li + LPR * i;
    if (cv < CV) {
      V8 o;
      f4 q0, q1;
      if (gamma) {
        V8 gm = *reinterpret_cast<const V8*>(gamma + cv * 8), bt = *reinterpret_cast<const V8*>(beta + cv * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { float f = ((float)v[i][j] - mean) * rstd * (float)gm[j] + (float)bt[j]; o[j] = (T)f; if (j < 4) q0[j] = f; else q1[j - 4] = f; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { float f = ((float)v[i][j] - mean) * rstd; o[j] = (T)f; if (j < 4) q0[j] = f; else q1[j - 4] = f; }
      }
      if (out8) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(y) + (long long)row * C + cv * 8) = pack8_fp8(q0, q1);
      else *reinterpret_cast<V8*>(yr + cv * 8) = o;
    }
  }
}

// ---- LayerNorm for any row length (C not a multiple of 8, or beyond what k_layer_norm keeps in registers): the reference's own
// tests normalise 10-element rows and (C, 10, 10) = 76 800 ... 160 000-element slabs (tests/layer_norm.py:22-71).  WAVE = true: one
// wave per row (4 rows per block); false: one 256-thread block per row.  Two-pass variance; the second and third sweeps hit L2.
template <bool WAVE, typename T = half_t>
__global__ void __launch_bounds__(256) k_layer_norm_any(T* __restrict__ y, const T* __restrict__ x, const T* __restrict__ gamma,
                                                        const T* __restrict__ beta, int rows, long long C, float eps) {
  typedef typename vec8<T>::type V8;
  __shared__ float red[8];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const long long row = WAVE ? (long long)blockIdx.x * 4 + w : blockIdx.x;
  const int t = WAVE ? l : threadIdx.x, nt = WAVE ? 64 : 256;
  const bool live = row < rows;
  const T* xr = x + (live ? row : 0) * C;
  const bool vec = (C & 7) == 0;
  auto block_sum = [&](float v, int slot) {
    v = wave_sum(v);
    if (WAVE) return v;
    if (l == 0) red[slot * 4 + w] = v;
    __syncthreads();
    return red[slot * 4] + red[slot * 4 + 1] + red[slot * 4 + 2] + red[slot * 4 + 3];
  };
  float s = 0.f;
  if (vec) {
    for (long long i = (long long)t * 8; i < C; i += nt * 8) { V8 v = *reinterpret_cast<const V8*>(xr + i); for (int j = 0; j < 8; ++j) s += (float)v[j]; }
  } else {
    for (long long i = t; i < C; i += nt) s += (float)xr[i];
  }
  const float mean = block_sum(s, 0) / (float)C;
  float q = 0.f;
  if (vec) {
    for (long long i = (long long)t * 8; i < C; i += nt * 8) { V8 v = *reinterpret_cast<const V8*>(xr + i); for (int j = 0; j < 8; ++j) { float d = (float)v[j] - mean; q += d * d; } }
  } else {
    for (long long i = t; i < C; i += nt) { float d = (float)xr[i] - mean; q += d * d; }
  }
  const float rstd = rsqrtf(block_sum(q, 1) / (float)C + eps);
  if (!live) return;
  T* yr = y + row * C;
  if (vec) {
    for (long long i = (long long)t * 8; i < C; i += nt * 8) {
      V8 v = *reinterpret_cast<const V8*>(xr + i), o;
      if (gamma) {
        V8 gm = *reinterpret_cast<const V8*>(gamma + i), bt = *reinterpret_cast<const V8*>(beta + i);
        for (int j = 0; j < 8; ++j) o[j] = (T)(((float)v[j] - mean) * rstd * (float)gm[j] + (float)bt[j]);
      } else {
        for (int j = 0; j < 8; ++j) o[j] = (T)(((float)v[j] - mean) * rstd);
      }
      *reinterpret_cast<V8*>(yr + i) = o;
    }
  } else {
    for (long long i = t; i < C; i += nt) {
      float f = ((float)xr[i] - mean) * rstd;
      yr[i] = (T)(gamma ? f * (float)gamma[i] + (float)beta[i] : f);
    }
  }
}

static void gn_geometry(int HW, int C, int N, int* CV, int* RPB, int* threads, int* chunks, int* ppc, int* ablocks, int* appb) {
  *CV = C / 8;
  *RPB = *CV >= 256 ? 1 : 256 / *CV;
  if (*RPB > HW) *RPB = HW > 0 ? HW : 1;
  *threads = *CV * *RPB;
  // stats: a thread sums ~8 pixels (enough loads in flight to hide latency); <= GN_MAX_CHUNKS chunks per image
  int p = *RPB * 8;
  int c = (HW + p - 1) / p;
  if (c > GN_MAX_CHUNKS) { c = GN_MAX_CHUNKS; p = (HW + c - 1) / c; p = ((p + *RPB - 1) / *RPB) * *RPB; c = (HW + p - 1) / p; }
  *ppc = p; *chunks = c;
  // apply: GN_APPLY_PPT pixels per thread, all loaded up front
  int q = *RPB * GN_APPLY_PPT;
  *appb = q; *ablocks = (HW + q - 1) / q;
}

// batches per block of k_gn_apply: as many as keep >= 3 blocks per CU (768) in the launch, at most 8 (small tensors: 1 = the round-1 geometry)
static int gn_batches(int ablocks, int N) {
  int nb = 1;
  while (nb < 8 && (long long)((ablocks + 2 * nb - 1) / (2 * nb)) * N >= 768) nb *= 2;
  return nb;
}

template <typename T>
static int layer_norm_impl(void* y, const void* x, const void* gamma, const void* beta, int rows, int C, float eps, int out8, tfStream_t s) {
  TF_REQUIRE(y && x && rows >= 0, "tf_layer_norm_f16: null tensor");
  TF_REQUIRE((gamma == nullptr) == (beta == nullptr), "tf_layer_norm_f16: gamma and beta must both be given or both NULL");
  TF_REQUIRE(C > 0, "tf_layer_norm_f16: C=%d", C);
  if (rows == 0) return TF_OK;
  TfProfScope prof_(TF_PROF_FAM_LAYER_NORM, (double)rows * C * (2.0 + (out8 ? 1.0 : 2.0)), tf_hs(s));
  if (C % 8 != 0 || C > 64 * 8 * LN_MAXV) {
    // any row length: a wave per row while the row is short, a block per row beyond
    if (C <= 4096) hipLaunchKernelGGL((k_layer_norm_any<true, T>), dim3(ceil_div(rows, 4)), dim3(256), 0, tf_hs(s), (T*)y, (const T*)x,
                                      (const T*)gamma, (const T*)beta, rows, (long long)C, eps);
    else hipLaunchKernelGGL((k_layer_norm_any<false, T>), dim3(rows), dim3(256), 0, tf_hs(s), (T*)y, (const T*)x, (const T*)gamma,
                            (const T*)beta, rows, (long long)C, eps);
    TF_LAUNCH_CHECK();
    return TF_OK;
  }
  int cv = C / 8;
#define LN_LAUNCH(LPR_)                                                                                                            \
  hipLaunchKernelGGL((k_layer_norm<LPR_, T>), dim3(ceil_div(rows, 4 * (64 / LPR_))), dim3(256), 0, tf_hs(s), (T*)y, (const T*)x, \
                     (const T*)gamma, (const T*)beta, rows, C, eps, out8)
  // few rows: one wave per row (most waves in flight); many rows: several rows per wave (more loads per lane)
  if (rows < 8192 || cv > 32 * LN_MAXV) LN_LAUNCH(64);
  else if (cv <= 8 * LN_MAXV) LN_LAUNCH(8);
  else if (cv <= 16 * LN_MAXV) LN_LAUNCH(16);
  else LN_LAUNCH(32);
#undef LN_LAUNCH
  TF_LAUNCH_CHECK();
  return TF_OK;
}

extern "C" size_t tf_group_norm_workspace(int N, int HW, int C, int G);
template <typename T>
static int group_norm_impl(void* y, const void* x, const void* x2, const void* gamma, const void* beta, int N, int HW, int C1, int C2, int G,
                      float eps, int silu, void* workspace, size_t workspace_bytes, tfStream_t s) {
  int C = C1 + C2;
  TF_REQUIRE(y && x && (C2 == 0 || x2), "tf_group_norm_f16: null tensor");
  TF_REQUIRE((gamma == nullptr) == (beta == nullptr), "tf_group_norm_f16: gamma and beta must both be given or both NULL");
  TF_REQUIRE(N >= 0 && HW >= 0 && G >= 1 && C > 0 && C % G == 0, "tf_group_norm_f16: C=%d not divisible by G=%d", C, G);
  TF_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && C / 8 <= 1024, "tf_group_norm_f16: C1=%d C2=%d must be multiples of 8 and C <= 8192", C1, C2);
  TF_REQUIRE(G <= 1024 && N <= 65535, "tf_group_norm_f16: G=%d N=%d out of range", G, N);
  if (N == 0 || HW == 0) return TF_OK;
  if (workspace_bytes < tf_group_norm_workspace(N, HW, C, G) || !workspace) {
    tf_set_error("tf_group_norm_f16: workspace %zu B < required %zu B", workspace_bytes, tf_group_norm_workspace(N, HW, C, G));
    return TF_E_WORKSPACE;
  }
  int CV, RPB, threads, chunks, ppc, ablocks, appb;
  gn_geometry(HW, C, N, &CV, &RPB, &threads, &chunks, &ppc, &ablocks, &appb);
  float* partial = (float*)workspace;
  int tl = (threads + 7) & ~7;                         // the folds work in groups of 8 lanes
  const int gnb = gn_batches(ablocks, N);
  TfProfScope prof_(TF_PROF_FAM_GROUP_NORM, (double)N * HW * C * 6.0, tf_hs(s));     // statistics pass (1 read) + apply (1 read + 1 write), 16-bit
  hipLaunchKernelGGL(k_gn_stats<T>, dim3(chunks, N), dim3(tl), (size_t)threads * 16 * sizeof(float), tf_hs(s), partial, (const T*)x,
                     (const T*)x2, HW, C1, C2, G, chunks, ppc, CV, RPB);
  TF_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_gn_apply<0, T>), dim3((ablocks + gnb - 1) / gnb, N), dim3(tl), 2 * G * sizeof(float), tf_hs(s), (T*)y, (const T*)x, (const T*)x2,
                     (const T*)gamma, (const T*)beta, (const float*)partial, HW, C1, C2, G, eps, silu, chunks, appb * gnb, CV, RPB, (const float*)nullptr, 0, 0, 0, 1, gnb);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

extern "C" {

size_t tf_group_norm_workspace(int N, int HW, int C, int G) {
  (void)HW; (void)C;
  return (size_t)N * GN_MAX_CHUNKS * G * 2 * sizeof(float);   // per-chunk partial (sum, sum of squares) per group
}

int tf_group_norm_f16(void* y, const void* x, const void* x2, const void* gamma, const void* beta, int N, int HW, int C1, int C2, int G,
                      float eps, int silu, void* workspace, size_t workspace_bytes, tfStream_t s) {
  return group_norm_impl<half_t>(y, x, x2, gamma, beta, N, HW, C1, C2, G, eps, silu, workspace, workspace_bytes, s);
}
/* bfloat16 in and out (tests/group_norm.py:12-19 runs bfloat16 next to float16); fp32 statistics, same workspace */
int tf_group_norm_bf16(void* y, const void* x, const void* x2, const void* gamma, const void* beta, int N, int HW, int C1, int C2, int G,
                       float eps, int silu, void* workspace, size_t workspace_bytes, tfStream_t s) {
  return group_norm_impl<bf16_t>(y, x, x2, gamma, beta, N, HW, C1, C2, G, eps, silu, workspace, workspace_bytes, s);
}

int tf_group_norm_apply_f16(void* y, const void* x, const void* gamma, const void* beta, const void* partial, int chunks, int N, int HW, int C, int G,
                            float eps, int silu, tfStream_t s) {
  return tf_group_norm_apply_16(TF_DTYPE_F16, y, x, gamma, beta, partial, chunks, N, HW, C, G, eps, silu, s);
}
int tf_group_norm_apply_16(int dtype, void* y, const void* x, const void* gamma, const void* beta, const void* partial, int chunks, int N, int HW, int C, int G,
                           float eps, int silu, tfStream_t s) {
  TF_REQUIRE(dtype == TF_DTYPE_F16 || dtype == TF_DTYPE_BF16, "tf_group_norm_apply_16: dtype=%d (0 = float16, 1 = bfloat16)", dtype);
  TF_REQUIRE(y && x && partial, "tf_group_norm_apply_f16: null tensor");
  TF_REQUIRE((gamma == nullptr) == (beta == nullptr), "tf_group_norm_apply_f16: gamma and beta must both be given or both NULL");
  TF_REQUIRE(N >= 0 && HW >= 0 && G >= 1 && C > 0 && C % G == 0, "tf_group_norm_apply_f16: C=%d not divisible by G=%d", C, G);
  TF_REQUIRE(C % 8 == 0 && C / 8 <= 1024 && G <= 1024 && N <= 65535, "tf_group_norm_apply_f16: C=%d G=%d N=%d out of range", C, G, N);
  TF_REQUIRE(chunks >= 1 && chunks <= 4096, "tf_group_norm_apply_f16: chunks=%d", chunks);
  if (N == 0 || HW == 0) return TF_OK;
  int CV, RPB, threads, sc, ppc, ablocks, appb;
  gn_geometry(HW, C, N, &CV, &RPB, &threads, &sc, &ppc, &ablocks, &appb);
  int tl = (threads + 7) & ~7;
  const int gnb = gn_batches(ablocks, N);
  TfProfScope prof_(TF_PROF_FAM_GROUP_NORM, (double)N * HW * C * 4.0, tf_hs(s));
  if (dtype == TF_DTYPE_BF16) hipLaunchKernelGGL((k_gn_apply<0, bf16_t>), dim3((ablocks + gnb - 1) / gnb, N), dim3(tl), 2 * G * sizeof(float), tf_hs(s), (bf16_t*)y, (const bf16_t*)x, (const bf16_t*)nullptr,
                     (const bf16_t*)gamma, (const bf16_t*)beta, (const float*)partial, HW, C, 0, G, eps, silu, chunks, appb * gnb, CV, RPB, (const float*)nullptr, 0, 0, 0, 1, gnb);
  else hipLaunchKernelGGL(k_gn_apply<0>, dim3((ablocks + gnb - 1) / gnb, N), dim3(tl), 2 * G * sizeof(float), tf_hs(s), (half_t*)y, (const half_t*)x, (const half_t*)nullptr,
                     (const half_t*)gamma, (const half_t*)beta, (const float*)partial, HW, C, 0, G, eps, silu, chunks, appb * gnb, CV, RPB, (const float*)nullptr, 0, 0, 0, 1, gnb);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

static int gn_apply_cat(void* y, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                        int groups1, const void* partial2, int chunks2, int groups2, int N, int HW, int C1, int C2, int G, float eps, int silu,
                        int out8, tfStream_t s);
int tf_group_norm_apply_cat_f16(void* y, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                                int groups1, const void* partial2, int chunks2, int groups2, int N, int HW, int C1, int C2, int G, float eps, int silu,
                                tfStream_t s) {
  return gn_apply_cat(y, x, x2, gamma, beta, partial, chunks, groups1, partial2, chunks2, groups2, N, HW, C1, C2, G, eps, silu, 0, s);
}
int tf_group_norm_apply_cat_16(int dtype, void* y, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                               int groups1, const void* partial2, int chunks2, int groups2, int N, int HW, int C1, int C2, int G, float eps, int silu,
                               tfStream_t s) {
  TF_REQUIRE(dtype == TF_DTYPE_F16 || dtype == TF_DTYPE_BF16, "tf_group_norm_apply_cat_16: dtype=%d (0 = float16, 1 = bfloat16)", dtype);
  return gn_apply_cat(y, x, x2, gamma, beta, partial, chunks, groups1, partial2, chunks2, groups2, N, HW, C1, C2, G, eps, silu, dtype == TF_DTYPE_BF16 ? 16 : 0, s);   // (out8 = 16: a bfloat16 tensor in, a bfloat16 tensor out)
}
/* the same with an e4m3 (fp8) output -- the operand of an fp8 conv (config 5); x2 / partial2 may be NULL (single source: groups1 = G).
 * mode 1: e4m3 at scale 1 (tf_group_norm_apply_fp8); mode 2: block-scaled e4m3 (tf_group_norm_apply_mx8: codes, then the E8M0 bytes) */
static int gn_apply_8(void* y8, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                      int groups1, const void* partial2, int chunks2, int groups2, int N, int HW, int C1, int C2, int G, float eps, int silu,
                      int mode, tfStream_t s) {
  if (mode == 2) TF_REQUIRE((C1 + C2) % 32 == 0, "tf_group_norm_apply_mx8: C=%d must be a multiple of 32 (one scale per 32 channels)", C1 + C2);
  if (!x2) {
    TF_REQUIRE(y8 && x && partial && C2 == 0 && groups1 == G, "tf_group_norm_apply_fp8: single source needs C2 = 0 and groups1 = G");
    TF_REQUIRE((gamma == nullptr) == (beta == nullptr), "tf_group_norm_apply_fp8: gamma and beta must both be given or both NULL");
    TF_REQUIRE(N >= 0 && HW >= 0 && G >= 1 && C1 > 0 && C1 % G == 0 && C1 % 8 == 0 && C1 / 8 <= 1024 && G <= 1024 && N <= 65535 && chunks >= 1 && chunks <= 4096,
               "tf_group_norm_apply_fp8: C=%d G=%d N=%d chunks=%d", C1, G, N, chunks);
    if (N == 0 || HW == 0) return TF_OK;
    int CV, RPB, threads, sc, ppc, ablocks, appb;
    gn_geometry(HW, C1, N, &CV, &RPB, &threads, &sc, &ppc, &ablocks, &appb);
    int tl = (threads + 7) & ~7;
    const int gnb = gn_batches(ablocks, N);
    TfProfScope prof_(TF_PROF_FAM_GROUP_NORM, (double)N * HW * C1 * 3.0, tf_hs(s));         // fp16 in, e4m3 out
    if (mode == 2) hipLaunchKernelGGL(k_gn_apply<2>, dim3((ablocks + gnb - 1) / gnb, N), dim3(tl), 2 * G * sizeof(float), tf_hs(s), (half_t*)y8, (const half_t*)x, (const half_t*)nullptr,
                       (const half_t*)gamma, (const half_t*)beta, (const float*)partial, HW, C1, 0, G, eps, silu, chunks, appb * gnb, CV, RPB, (const float*)nullptr, 0, 0, 0, 1, gnb);
    else hipLaunchKernelGGL(k_gn_apply<1>, dim3((ablocks + gnb - 1) / gnb, N), dim3(tl), 2 * G * sizeof(float), tf_hs(s), (half_t*)y8, (const half_t*)x, (const half_t*)nullptr,
                       (const half_t*)gamma, (const half_t*)beta, (const float*)partial, HW, C1, 0, G, eps, silu, chunks, appb * gnb, CV, RPB, (const float*)nullptr, 0, 0, 0, 1, gnb);
    TF_LAUNCH_CHECK();
    return TF_OK;
  }
  return gn_apply_cat(y8, x, x2, gamma, beta, partial, chunks, groups1, partial2, chunks2, groups2, N, HW, C1, C2, G, eps, silu, mode, s);
}
int tf_group_norm_apply_fp8(void* y8, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                            int groups1, const void* partial2, int chunks2, int groups2, int N, int HW, int C1, int C2, int G, float eps, int silu,
                            tfStream_t s) {
  return gn_apply_8(y8, x, x2, gamma, beta, partial, chunks, groups1, partial2, chunks2, groups2, N, HW, C1, C2, G, eps, silu, 1, s);
}
int tf_group_norm_apply_mx8(void* y_mx, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                            int groups1, const void* partial2, int chunks2, int groups2, int N, int HW, int C1, int C2, int G, float eps, int silu,
                            tfStream_t s) {
  return gn_apply_8(y_mx, x, x2, gamma, beta, partial, chunks, groups1, partial2, chunks2, groups2, N, HW, C1, C2, G, eps, silu, 2, s);
}
static int gn_apply_cat(void* y, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                        int groups1, const void* partial2, int chunks2, int groups2, int N, int HW, int C1, int C2, int G, float eps, int silu,
                        int out8, tfStream_t s) {
  TF_REQUIRE(y && x && x2 && partial && partial2, "tf_group_norm_apply_cat_f16: null tensor");
  TF_REQUIRE((gamma == nullptr) == (beta == nullptr), "tf_group_norm_apply_cat_f16: gamma and beta must both be given or both NULL");
  const int C = C1 + C2;
  TF_REQUIRE(N >= 0 && HW >= 0 && G >= 1 && C1 > 0 && C2 > 0 && C % G == 0 && groups1 >= 1 && groups2 >= 1 && C1 % groups1 == 0 && C2 % groups2 == 0,
             "tf_group_norm_apply_cat_f16: C1=%d C2=%d G=%d groups1=%d groups2=%d", C1, C2, G, groups1, groups2);
  const int sub = C1 / groups1, cpg = C / G;
  TF_REQUIRE(C2 / groups2 == sub && cpg % sub == 0 && cpg / sub <= 8,
             "tf_group_norm_apply_cat_f16: the partials' sub-groups (%d and %d channels) do not tile the %d-channel groups of the concat", sub, C2 / groups2, cpg);
  TF_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && C / 8 <= 1024 && G <= 1024 && N <= 65535, "tf_group_norm_apply_cat_f16: C1=%d C2=%d G=%d N=%d out of range", C1, C2, G, N);
  TF_REQUIRE(chunks >= 1 && chunks <= 4096 && chunks2 >= 1 && chunks2 <= 4096, "tf_group_norm_apply_cat_f16: chunks=%d chunks2=%d", chunks, chunks2);
  if (N == 0 || HW == 0) return TF_OK;
  int CV, RPB, threads, sc, ppc, ablocks, appb;
  gn_geometry(HW, C, N, &CV, &RPB, &threads, &sc, &ppc, &ablocks, &appb);
  int tl = (threads + 7) & ~7;
  const int gnb = gn_batches(ablocks, N);
  TfProfScope prof_(TF_PROF_FAM_GROUP_NORM, (double)N * HW * C * ((out8 == 1 || out8 == 2) ? 3.0 : 4.0), tf_hs(s));
  if (out8 == 16) hipLaunchKernelGGL((k_gn_apply<0, bf16_t>), dim3((ablocks + gnb - 1) / gnb, N), dim3(tl), 2 * G * sizeof(float), tf_hs(s), (bf16_t*)y, (const bf16_t*)x, (const bf16_t*)x2,
                               (const bf16_t*)gamma, (const bf16_t*)beta, (const float*)partial, HW, C1, C2, G, eps, silu, chunks, appb * gnb, CV, RPB,
                               (const float*)partial2, chunks2, groups1, groups2, cpg / sub, gnb);
  else if (out8 == 2) hipLaunchKernelGGL(k_gn_apply<2>, dim3((ablocks + gnb - 1) / gnb, N), dim3(tl), 2 * G * sizeof(float), tf_hs(s), (half_t*)y, (const half_t*)x, (const half_t*)x2,
                               (const half_t*)gamma, (const half_t*)beta, (const float*)partial, HW, C1, C2, G, eps, silu, chunks, appb * gnb, CV, RPB,
                               (const float*)partial2, chunks2, groups1, groups2, cpg / sub, gnb);
  else if (out8) hipLaunchKernelGGL(k_gn_apply<1>, dim3((ablocks + gnb - 1) / gnb, N), dim3(tl), 2 * G * sizeof(float), tf_hs(s), (half_t*)y, (const half_t*)x, (const half_t*)x2,
                               (const half_t*)gamma, (const half_t*)beta, (const float*)partial, HW, C1, C2, G, eps, silu, chunks, appb * gnb, CV, RPB,
                               (const float*)partial2, chunks2, groups1, groups2, cpg / sub, gnb);
  else hipLaunchKernelGGL(k_gn_apply<0>, dim3((ablocks + gnb - 1) / gnb, N), dim3(tl), 2 * G * sizeof(float), tf_hs(s), (half_t*)y, (const half_t*)x, (const half_t*)x2,
                          (const half_t*)gamma, (const half_t*)beta, (const float*)partial, HW, C1, C2, G, eps, silu, chunks, appb * gnb, CV, RPB,
                          (const float*)partial2, chunks2, groups1, groups2, cpg / sub, gnb);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

int tf_group_norm_apply2_f16(void* y, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                             const void* partial2, int chunks2, int N, int HW, int C1, int G, float eps, int silu, tfStream_t s) {
  TF_REQUIRE(G >= 2 && G % 2 == 0, "tf_group_norm_apply2_f16: G=%d must be even", G);
  return tf_group_norm_apply_cat_f16(y, x, x2, gamma, beta, partial, chunks, G, partial2, chunks2, G, N, HW, C1, C1, G, eps, silu, s);
}

int tf_layer_norm_f16(void* y, const void* x, const void* gamma, const void* beta, int rows, int C, float eps, tfStream_t s) {
  return layer_norm_impl<half_t>(y, x, gamma, beta, rows, C, eps, 0, s);
}
/* bfloat16 in and out (tests/layer_norm.py:13-27 runs bfloat16 next to float16); fp32 statistics */
int tf_layer_norm_bf16(void* y, const void* x, const void* gamma, const void* beta, int rows, int C, float eps, tfStream_t s) {
  return layer_norm_impl<bf16_t>(y, x, gamma, beta, rows, C, eps, 0, s);
}
/* LayerNorm with an e4m3 (fp8) output: the operand of an fp8 Linear (config 5's FeedForward); C a multiple of 8, <= 2560 */
int tf_layer_norm_fp8(void* y8, const void* x, const void* gamma, const void* beta, int rows, int C, float eps, tfStream_t s) {
  TF_REQUIRE(C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAXV, "tf_layer_norm_fp8: C=%d must be a multiple of 8 and <= %d", C, 64 * 8 * LN_MAXV);
  return layer_norm_impl<half_t>(y8, x, gamma, beta, rows, C, eps, 1, s);
}
/* LayerNorm with a block-scaled e4m3 output (rows x C codes, then rows x C/32 E8M0 bytes): the operand of tf_linear_mx8; C a multiple of 32 */
int tf_layer_norm_mx8(void* y_mx, const void* x, const void* gamma, const void* beta, int rows, int C, float eps, tfStream_t s) {
  TF_REQUIRE(C > 0 && C % 32 == 0 && C <= 64 * 8 * LN_MAXV, "tf_layer_norm_mx8: C=%d must be a multiple of 32 and <= %d", C, 64 * 8 * LN_MAXV);
  return layer_norm_impl<half_t>(y_mx, x, gamma, beta, rows, C, eps, 2, s);
}

}  // extern "C"
