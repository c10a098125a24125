// Implicit-GEMM convolution / linear on MFMA (v_mfma_f32_16x16x32_f16), gfx950.
//
//   Y[m, n] = act( sum_k  X_gather[m, k] * Wt[n, k]  + bias[n] + bias_nc[img(m), n] ) + residual[m, n]
//
// m = output pixel (img, ho, wo) of an NHWC tensor (or a token row for Linear), n = output channel,
// k = (r, s, c) with c innermost, matching the packed "KRSC" weight.  The gather folds in zero padding,
// stride, the nearest-2x upsample (vision/unet.py:81-83) and the channel concat (vision/unet.py:72).
// Reference ops replaced: conv_2d/Conv2d (vision/conv2d.py:9-58), Linear (ff/linear.py:112-121),
// GEGLU's split+gelu (ff/nn.py:10-12), the emb / residual adds of vision/resnet.py:28-30.
//
// Tiling: block = 4 waves (2 x 2), block tile BM x BN, BK = 64.  Both operands are K-contiguous 128-B
// rows, staged global -> LDS with global_load_lds_dwordx4 (LDS image lane-linear, XOR swizzle applied on
// the per-lane SOURCE chunk and again on the ds_read_b128), two LDS stages, one barrier per K tile.
// The weight tile is the MFMA "A" operand and the activation tile the "B" operand, so each lane ends up
// with 4 consecutive output channels of one pixel: 8-byte stores, vector bias/residual loads.
#include "common.h"
#include "../../include/tinyfusers_hip.h"
#include <vector>

struct GemmP {
  const half_t* x; const half_t* x2; const half_t* w; half_t* y;
  const half_t* bias; const half_t* bias_nc; const half_t* residual; float* partial;
  long long bias_nc_stride;
  const half_t* zeros;
  int M, N, K;          // N = rows of w (2x the output width for GEGLU)
  int C1, C2, C;
  int H, W, Ho, Wo, HoWo;
  int S, stride, pad, ups;
  int ktiles, ktiles_per_split, splitk;
  int act;              // 0 none, 1 GEGLU
  int ntm, ntn;         // tile counts
};

__device__ __forceinline__ void glds16(const half_t* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN>
__global__ void __launch_bounds__(256) k_igemm(const GemmP p) {
  constexpr int TM = BM / 2, TN = BN / 2, MJ = TM / 16, NI = TN / 16;
  constexpr int A_ROUNDS = BM / 32, B_ROUNDS = BN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile order: blocks b and b+8 share an XCD's L2; give every XCD a contiguous run of tiles
  // (n fastest, so a run re-uses the same activation rows and sweeps the weight tiles).
  const int nblk = p.ntm * p.ntn;
  int bid = blockIdx.x;
  {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / p.ntn, tile_n = bid - tile_m * p.ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int split = blockIdx.y;
  const int kt_begin = split * p.ktiles_per_split;
  const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);

  // ---- per-thread staging geometry: each glds round moves 32 rows x 128 B (8 rows per wave) -----
  const int rir = wid * 8 + (lane >> 3);                   // row within a round
  const int cs = (lane & 7) ^ ((rir >> 1) & 7);            // source 16-B chunk (swizzle on the source side)
  // activation rows owned by this thread
  int a_hi0[A_ROUNDS], a_wi0[A_ROUNDS];
  long long a_img[A_ROUNDS];
#pragma unroll
  for (int i = 0; i < A_ROUNDS; ++i) {
    int m = m0 + i * 32 + rir;
    if (m < p.M) {
      int img = m / p.HoWo, rem = m - img * p.HoWo;
      int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hi0[i] = ho * p.stride - p.pad;
      a_wi0[i] = wo * p.stride - p.pad;
      a_img[i] = (long long)img * p.H * p.W;
    } else {
      a_hi0[i] = -(1 << 28);   // always out of range -> zero page
      a_wi0[i] = 0;
      a_img[i] = 0;
    }
  }
  const int Hl = p.H << p.ups, Wl = p.W << p.ups;   // logical (post-upsample) input extent

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
    int kg = kt * 64 + cs * 8;              // this thread's K position (same for all rounds)
    // ---- activations
    int tap = kg / p.C, c = kg - tap * p.C;
    int r = tap / p.S, s = tap - r * p.S;
    const half_t* xs; int ld;
    if (c < p.C1) { xs = p.x + c; ld = p.C1; } else { xs = p.x2 + (c - p.C1); ld = p.C2; }
    bool kvalid = kg < p.K;
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) {
      int hi = a_hi0[i] + r, wi = a_wi0[i] + s;
      bool ok = kvalid && hi >= 0 && hi < Hl && wi >= 0 && wi < Wl;
      const half_t* src = ok ? xs + (a_img[i] + (long long)(hi >> p.ups) * p.W + (wi >> p.ups)) * ld : p.zeros;
      glds16(src, base + (i * 32 + wid * 8) * 128);
    }
    // ---- weights
#pragma unroll
    for (int i = 0; i < B_ROUNDS; ++i) {
      int n = n0 + i * 32 + rir;
      const half_t* src = (kvalid && n < p.N) ? p.w + (long long)n * p.K + kg : p.zeros;
      glds16(src, base + A_BYTES + (i * 32 + wid * 8) * 128);
    }
  };

  const int wave_m = wid & 1, wave_n = wid >> 1;
  const int lr = lane & 15, lg = lane >> 4;
  f4 acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  if (kt_begin < kt_end) stage(0, kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < kt_end) stage(buf ^ 1, kt + 1);
    const char* sa = smem + buf * STAGE;             // activation rows
    const char* sb = sa + A_BYTES;                   // weight rows
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8 wf[NI], xf[MJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int row = wave_n * TN + i * 16 + lr;
        int ch = (ks * 4 + lg) ^ ((row >> 1) & 7);
        wf[i] = *reinterpret_cast<const h8*>(sb + row * 128 + ch * 16);
      }
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        int row = wave_m * TM + j * 16 + lr;
        int ch = (ks * 4 + lg) ^ ((row >> 1) & 7);
        xf[j] = *reinterpret_cast<const h8*>(sa + row * 128 + ch * 16);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds n = nb + 4*lg + {0..3} for pixel m = mb + lr --------------------------
  if (p.splitk > 1) {
    float* part = p.partial + (long long)split * p.M * p.N;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      int m = m0 + wave_m * TM + j * 16 + lr;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int n = n0 + wave_n * TN + i * 16 + lg * 4;
        if (n + 3 < p.N && (p.N & 3) == 0) {
          *reinterpret_cast<f4*>(part + (long long)m * p.N + n) = acc[i][j];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (n + e < p.N) part[(long long)m * p.N + n + e] = acc[i][j][e];
        }
      }
    }
    return;
  }
  if (p.act == 1) {
    // GEGLU: w rows come in 16-row blocks alternating value / gate, so acc[2i] / acc[2i+1] pair up per lane
    const int No = p.N >> 1;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      int m = m0 + wave_m * TM + j * 16 + lr;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i + 1 < NI; i += 2) {
        int n = n0 + wave_n * TN + i * 16 + lg * 4;    // packed row index of the value block
        if (n >= p.N) continue;
        int no = (n >> 5) * 16 + (n & 15);              // output column
        h4 ba = *reinterpret_cast<const h4*>(p.bias + n), bg = *reinterpret_cast<const h4*>(p.bias + n + 16);
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = acc[i][j][e] + (float)ba[e], g = acc[i + 1][j][e] + (float)bg[e];
          o[e] = (half_t)(a * gelu_f(g));
        }
        if (p.residual) {
          h4 rv = *reinterpret_cast<const h4*>(p.residual + (long long)m * No + no);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)((float)o[e] + (float)rv[e]);
        }
        *reinterpret_cast<h4*>(p.y + (long long)m * No + no) = o;
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    int m = m0 + wave_m * TM + j * 16 + lr;
    if (m >= p.M) continue;
    int img = p.bias_nc ? m / p.HoWo : 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int n = n0 + wave_n * TN + i * 16 + lg * 4;
      if (n >= p.N) continue;
      f4 v = acc[i][j];
      if (n + 3 < p.N && (p.N & 3) == 0) {
        if (p.bias) { h4 b = *reinterpret_cast<const h4*>(p.bias + n); for (int e = 0; e < 4; ++e) v[e] += (float)b[e]; }
        if (p.bias_nc) { h4 b = *reinterpret_cast<const h4*>(p.bias_nc + (long long)img * p.bias_nc_stride + n); for (int e = 0; e < 4; ++e) v[e] += (float)b[e]; }
        if (p.residual) { h4 b = *reinterpret_cast<const h4*>(p.residual + (long long)m * p.N + n); for (int e = 0; e < 4; ++e) v[e] += (float)b[e]; }
        h4 o;
        for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
        *reinterpret_cast<h4*>(p.y + (long long)m * p.N + n) = o;
      } else {
        for (int e = 0; e < 4; ++e) {
          if (n + e >= p.N) break;
          float f = v[e];
          if (p.bias) f += (float)p.bias[n + e];
          if (p.bias_nc) f += (float)p.bias_nc[(long long)img * p.bias_nc_stride + n + e];
          if (p.residual) f += (float)p.residual[(long long)m * p.N + n + e];
          p.y[(long long)m * p.N + n + e] = (half_t)f;
        }
      }
    }
  }
}

// split-K reduce + epilogue: y[m,n] = sum_z partial[z,m,n] + bias + bias_nc + residual   (N % 4 == 0 fast path)
__global__ void __launch_bounds__(256) k_splitk_reduce(half_t* __restrict__ y, const float* __restrict__ partial, const half_t* __restrict__ bias,
                                                       const half_t* __restrict__ bias_nc, const half_t* __restrict__ residual, int M, int N,
                                                       int HoWo, int splitk, long long bnc_stride) {
  long long total = (long long)M * N;
  long long gs = (long long)gridDim.x * 256;
  if ((N & 3) == 0) {
    long long nv = total >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += gs) {
      long long e0 = i << 2;
      int m = (int)(e0 / N), n = (int)(e0 - (long long)m * N);
      f4 v = *reinterpret_cast<const f4*>(partial + e0);
      for (int z = 1; z < splitk; ++z) {
        f4 u = *reinterpret_cast<const f4*>(partial + (long long)z * total + e0);
        v += u;
      }
      if (bias) { h4 b = *reinterpret_cast<const h4*>(bias + n); for (int e = 0; e < 4; ++e) v[e] += (float)b[e]; }
      if (bias_nc) { h4 b = *reinterpret_cast<const h4*>(bias_nc + (long long)(m / HoWo) * bnc_stride + n); for (int e = 0; e < 4; ++e) v[e] += (float)b[e]; }
      if (residual) { h4 b = *reinterpret_cast<const h4*>(residual + e0); for (int e = 0; e < 4; ++e) v[e] += (float)b[e]; }
      h4 o;
      for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
      *reinterpret_cast<h4*>(y + e0) = o;
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += gs) {
      int m = (int)(i / N), n = (int)(i - (long long)m * N);
      float v = 0.f;
      for (int z = 0; z < splitk; ++z) v += partial[(long long)z * total + i];
      if (bias) v += (float)bias[n];
      if (bias_nc) v += (float)bias_nc[(long long)(m / HoWo) * bnc_stride + n];
      if (residual) v += (float)residual[i];
      y[i] = (half_t)v;
    }
  }
}

// ---- weight-streaming GEMV for M <= 8 (time-embedding MLP, ResBlock emb_layers): one wave per output row
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
          float xf = (float)xv[j];
          if (silu_in) xf = silu_f(xf);
          acc[m] += xf * (float)wv8[j];
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    if (m < M) {
      float v = wave_sum(acc[m]);
      if (l == 0) y[(long long)m * N + n] = (half_t)(v + (bias ? (float)bias[n] : 0.f));
    }
  }
}

// ------------------------------------------------------------------------------------------------
static half_t* g_zeros = nullptr;
static int ensure_zeros() {
  if (g_zeros) return TF_OK;
  TF_HIP(hipMalloc((void**)&g_zeros, 4096));
  TF_HIP(hipMemset(g_zeros, 0, 4096));
  return TF_OK;
}

// per-launch event profiling of this kernel family (bench.py roofline leg)
static bool g_prof = false;
static double g_prof_ms = 0.0, g_prof_flops = 0.0;
static long long g_prof_launches = 0;
struct ProfRec { hipEvent_t a, b; double flops; };
static std::vector<ProfRec> g_prof_pending;

struct TileCfg { int bm, bn, splitk; };

static TileCfg choose_tiles(int M, int N, int K, int act, bool allow_split) {
  static const int cand[][2] = {{128, 160}, {64, 160}, {128, 128}, {64, 128}, {128, 64}, {64, 64}};
  const int ncand = 6;
  const double CUS = 256.0;
  int ktiles = (K + 63) / 64;
  TileCfg best = {64, 64, 1};
  double best_t = 1e30;
  for (int ci = 0; ci < ncand; ++ci) {
    int bm = cand[ci][0], bn = cand[ci][1];
    if (act == 1 && (bn % 64) != 0) continue;           // GEGLU pairs 16-row blocks inside a wave tile
    int ntm = (M + bm - 1) / bm, ntn = (N + bn - 1) / bn;
    double tiles = (double)ntm * ntn;
    // relative MFMA efficiency of the tile shape (LDS bytes per MFMA) -- refined from measurements
    double eff = (bm == 128 ? 1.0 : 0.82) * (bn >= 128 ? 1.0 : 0.8);
    int max_split = (allow_split && act == 0) ? 32 : 1;
    for (int sk = 1; sk <= max_split; sk *= 2) {
      if (sk > 1 && ktiles / sk < 4) break;
      double blocks = tiles * sk;
      double waves = ceil(blocks / CUS);
      double per_block = (double)bm * bn * ((ktiles + sk - 1) / sk) * 64.0 / eff;
      double t = waves * per_block;
      // padding waste is already in bm*bn; split-K pays an fp32 round trip of the output
      if (sk > 1) t += (double)M * N * (sk + 1) * 4.0 * 40.0 / CUS;   // ~bytes -> mfma-equivalent cost units
      t += 3.0e4 * 64.0;                                              // fixed per-launch latency
      if (t < best_t) { best_t = t; best = {bm, bn, sk}; }
    }
  }
  return best;
}

template <int BM, int BN>
static int launch_cfg(const GemmP& p, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_igemm<BM, BN>), dim3(p.ntm * p.ntn, p.splitk), dim3(256), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

static int run_gemm(GemmP p, void* workspace, size_t workspace_bytes, int force_bm, int force_bn, int force_split, hipStream_t st) {
  int rc = ensure_zeros();
  if (rc) return rc;
  p.zeros = g_zeros;
  p.ktiles = (p.K + 63) / 64;
  TileCfg c = choose_tiles(p.M, p.N, p.K, p.act, true);
  if (force_bm) { c.bm = force_bm; c.bn = force_bn; c.splitk = force_split > 0 ? force_split : 1; }
  if (c.splitk > 1) {
    size_t need = (size_t)c.splitk * p.M * p.N * sizeof(float);
    if (!workspace || workspace_bytes < need) c.splitk = 1;   // degrade gracefully: correctness does not depend on split-K
  }
  p.splitk = c.splitk;
  p.ktiles_per_split = (p.ktiles + c.splitk - 1) / c.splitk;
  p.splitk = (p.ktiles + p.ktiles_per_split - 1) / p.ktiles_per_split;
  p.partial = (float*)workspace;
  p.ntm = (p.M + c.bm - 1) / c.bm;
  p.ntn = (p.N + c.bn - 1) / c.bn;
  ProfRec rec;
  if (g_prof) {
    TF_HIP(hipEventCreate(&rec.a)); TF_HIP(hipEventCreate(&rec.b));
    rec.flops = 2.0 * p.M * (double)(p.act == 1 ? p.N : p.N) * p.K;
    TF_HIP(hipEventRecord(rec.a, st));
  }
  if (c.bm == 128 && c.bn == 160) rc = launch_cfg<128, 160>(p, st);
  else if (c.bm == 64 && c.bn == 160) rc = launch_cfg<64, 160>(p, st);
  else if (c.bm == 128 && c.bn == 128) rc = launch_cfg<128, 128>(p, st);
  else if (c.bm == 64 && c.bn == 128) rc = launch_cfg<64, 128>(p, st);
  else if (c.bm == 128 && c.bn == 64) rc = launch_cfg<128, 64>(p, st);
  else if (c.bm == 64 && c.bn == 64) rc = launch_cfg<64, 64>(p, st);
  else { tf_set_error("run_gemm: no kernel for tile %dx%d", c.bm, c.bn); return TF_E_UNSUPPORTED; }
  if (rc) return rc;
  if (p.splitk > 1) {
    long long nv = ((long long)p.M * p.N) >> 2;
    int grid = (int)((nv + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_splitk_reduce, dim3(grid), dim3(256), 0, st, p.y, (const float*)p.partial, p.bias, p.bias_nc, p.residual, p.M, p.N,
                       p.HoWo, p.splitk, p.bias_nc_stride);
    TF_LAUNCH_CHECK();
  }
  if (g_prof) { TF_HIP(hipEventRecord(rec.b, st)); g_prof_pending.push_back(rec); }
  return TF_OK;
}

static size_t gemm_workspace(int M, int N, int K, int act) {
  TileCfg c = choose_tiles(M, N, K, act, true);
  return c.splitk > 1 ? (size_t)c.splitk * M * N * sizeof(float) : 0;
}

// test hook: force a tile configuration (0 = heuristic)
static int g_force_bm = 0, g_force_bn = 0, g_force_split = 0;

extern "C" {

int tf_gemm_force_config(int bm, int bn, int splitk) { g_force_bm = bm; g_force_bn = bn; g_force_split = splitk; return TF_OK; }

int tf_prof_enable(int on) {
  g_prof = on != 0;
  if (on) { g_prof_ms = 0.0; g_prof_flops = 0.0; g_prof_launches = 0; g_prof_pending.clear(); }
  return TF_OK;
}
int tf_prof_read(double* ms, double* flops, long long* launches) {
  for (auto& r : g_prof_pending) {
    float t = 0.f;
    TF_HIP(hipEventSynchronize(r.b));
    TF_HIP(hipEventElapsedTime(&t, r.a, r.b));
    g_prof_ms += t; g_prof_flops += r.flops; g_prof_launches += 1;
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  g_prof_pending.clear();
  if (ms) *ms = g_prof_ms;
  if (flops) *flops = g_prof_flops;
  if (launches) *launches = g_prof_launches;
  return TF_OK;
}

static int conv_geometry(int H, int W, int R, int S, int stride, int pad, int ups, int* Ho, int* Wo) {
  int Hl = H << ups, Wl = W << ups;
  *Ho = (Hl + 2 * pad - R) / stride + 1;
  *Wo = (Wl + 2 * pad - S) / stride + 1;
  return (*Ho > 0 && *Wo > 0) ? 0 : 1;
}

size_t tf_conv2d_workspace(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample) {
  int Ho, Wo;
  if (stride < 1 || conv_geometry(H, W, R, S, stride, pad, upsample ? 1 : 0, &Ho, &Wo)) return 0;
  return gemm_workspace(N * Ho * Wo, Cout, R * S * (C1 + C2), 0);
}

int tf_conv2d_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                  const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                  void* workspace, size_t workspace_bytes, tfStream_t s) {
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
  p.M = N * Ho * Wo; p.N = Cout; p.C1 = C1; p.C2 = C2; p.C = C1 + C2; p.K = R * S * p.C;
  p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.HoWo = Ho * Wo; p.S = S; p.stride = stride; p.pad = pad; p.ups = ups; p.act = 0;
  return run_gemm(p, workspace, workspace_bytes, g_force_bm, g_force_bn, g_force_split, tf_hs(s));
}

size_t tf_linear_workspace(int M, int N, int K, int act) { return gemm_workspace(M, act == 1 ? 2 * N : N, K, act); }

int tf_linear_f16(void* y, const void* x, const void* w, const void* bias, const void* residual, int M, int N, int K, int act,
                  void* workspace, size_t workspace_bytes, tfStream_t s) {
  TF_REQUIRE(y && x && w, "tf_linear_f16: null tensor");
  TF_REQUIRE(M >= 0 && N >= 1 && K >= 8 && K % 8 == 0, "tf_linear_f16: K=%d must be a positive multiple of 8", K);
  TF_REQUIRE(act == 0 || act == 1, "tf_linear_f16: act=%d", act);
  TF_REQUIRE(act == 0 || (bias && N % 16 == 0), "tf_linear_f16: GEGLU needs a bias and N %% 16 == 0 (N=%d)", N);
  if (M == 0) return TF_OK;
  GemmP p = {};
  p.x = (const half_t*)x; p.w = (const half_t*)w; p.y = (half_t*)y; p.bias = (const half_t*)bias; p.residual = (const half_t*)residual;
  p.M = M; p.N = act == 1 ? 2 * N : N; p.K = K; p.C1 = K; p.C2 = 0; p.C = K;
  p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.HoWo = M; p.S = 1; p.stride = 1; p.pad = 0; p.ups = 0; p.act = act;
  return run_gemm(p, workspace, workspace_bytes, g_force_bm, g_force_bn, g_force_split, tf_hs(s));
}

int tf_gemv_f16(void* y, const void* x, const void* w, const void* bias, int M, int N, int K, int silu_input, tfStream_t s) {
  TF_REQUIRE(y && x && w && M >= 1 && M <= 8 && N >= 1 && K % 8 == 0, "tf_gemv_f16: needs 1 <= M <= 8 (M=%d) and K %% 8 == 0 (K=%d)", M, K);
  hipLaunchKernelGGL(k_gemv, dim3(ceil_div(N, 4)), dim3(256), 0, tf_hs(s), (half_t*)y, (const half_t*)x, (const half_t*)w, (const half_t*)bias, M, N, K, silu_input);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

}  // extern "C"
