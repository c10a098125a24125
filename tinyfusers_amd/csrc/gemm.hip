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
// the per-lane SOURCE chunk and again on the ds_read_b128) into a 4-slot LDS ring: one raw s_barrier per
// K tile and a counted s_waitcnt vmcnt(N) that leaves two tiles in flight across it (at batch 1 most shapes
// run one block per CU, so the pipeline, not occupancy, has to hide the L2/HBM latency).
// The weight tile is the MFMA "A" operand and the activation tile the "B" operand, so each lane ends up
// with 4 consecutive output channels of one pixel: 8-byte stores, vector bias/residual loads.
#include "common.h"
#include <type_traits>
#include "../../include/tinyfusers_hip.h"
#include <vector>

struct GemmP {
  const half_t* x; const half_t* x2; const half_t* w; half_t* y;
  // extra K segment after the R*S taps (tf_conv2d_fused_f16): a 1x1 projection of a second activation (pair) x3 | x4 read
  // at the output pixel itself -- the ResBlock's skip_connection folded into its last conv (vision/resnet.py:24, :31)
  const half_t* x3; const half_t* x4;
  int C3, C4, Kc;       // Kc = R*S*(C1+C2): where the extra segment starts inside K
  unsigned x3_bytes, x4_bytes;
  const half_t* bias; const half_t* bias_nc; const half_t* residual; float* partial;
  long long bias_nc_stride;
  const float* ln_colsum;   // LayerNorm folded into this GEMM (see tf_linear_ln_f16): colsum[n] = sum_k w'[n,k]; NULL = off
  float ln_eps;
  unsigned x_bytes, x2_bytes, w_bytes;
  int M, N, K;          // N = rows of w (2x the output width for GEGLU)
  int C1, C2, C;
  int H, W, Ho, Wo, HoWo;
  int S, stride, pad, ups;
  int ktiles, ktiles_per_split, splitk;
  int act;              // 0 none, 1 GEGLU
  int ntm, ntn;         // tile counts
  int order;            // block -> tile order inside an XCD's run: 0 = n fastest (share activation rows), 1 = m fastest (share the weight tile)
  unsigned dv_howo_mul, dv_howo_shr, dv_wo_mul, dv_wo_shr;   // magic numbers: n / HoWo, n / Wo without a divide
  int dbg;              // diagnostic builds only (tools/gemm_bench.py): 1 no stores, 2 no MFMA, 4 no staging
  // GroupNorm statistics of the OUTPUT emitted by the epilogue (tf_conv2d_fused_f16): per (image, chunk, group) partial
  // (sum, sum of squares) of the fp16-rounded outputs, in the layout k_gn_apply folds; NULL = off
  float* gn_part;
  int gn_G, gn_cpg, gn_chunks;
  // k_igemm_patch geometry (patch_setup): pieces / pixels of one activation patch, bytes of a ring slot, ring depth, log2(W)
  int pt_ppc, pt_ppix, pt_stage, pt_ns, pt_log2w;
  // GroupNorm (+ SiLU) of the INPUT applied inside this launch (tf_conv2d_gn_f16; vision/resnet.py:8-22 GN -> SiLU -> conv,
  // attention/attention.py:66-68 GN -> 1x1 conv): the statistics arrive as the producers' partials (the layout k_gn_apply folds),
  // the consumer waves fold them into a per-channel (a, b) table in LDS during the prologue, and the loader waves normalise the
  // activation pieces they staged -- in LDS, once per piece -- before the consumers read them.  gi_part == NULL: off.
  const float* gi_part; const float* gi_part2;
  const half_t* gi_gamma; const half_t* gi_beta;
  int gi_chunks, gi_chunks2, gi_G, gi_G1, gi_G2, gi_mr, gi_silu;
  float gi_eps;
  int gi_off;           // byte offset of the table in LDS: [G] (mean, rstd) then [C] (a, b), fp32 pairs
  // GroupNorm (+ SiLU) of the OUTPUT applied by the split-K reduce (tf_conv2d_fused_norm_f16): when the shape runs split-K, the reduce
  // kernel owns whole (image, group) slabs, so it can finish the statistics AND write the normalised tensor z next to y
  half_t* on_z; const half_t* on_gamma; const half_t* on_beta; float on_eps; int on_silu; int* on_applied;
  // fp8 (OCP e4m3) operands (k_igemm8, BASELINE config 5): x / x2 / w hold ONE byte per element, wscale[n] is the per-output-channel
  // weight scale applied to the fp32 accumulators in the epilogue (activations use scale 1: normalised tensors); out8: y is stored as
  // e4m3 as well (the GEGLU output that feeds the next fp8 GEMM)
  const float* wscale; int out8, fp8;
  // bfloat16 operands, bias, residual and output (tf_linear_bf16 / tf_conv2d_bf16): the plain deep ring with the bf16 MFMA, no split-K
  int bf16;
  // tf_linear_f32out_f16: the raw fp32 accumulators go to out32[m, n] (no bias / residual / activation, never split along K) -- the
  // q k^T scores of the unfused attention path, which must not be rounded to fp16 before the softmax; NULL = off
  float* out32;
  int c4_chunk;         // k_gemm_c4: consecutive tiles a block takes before it strides on by gridDim chunks (launch_c4)
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;   // 128-bit buffer resource
typedef bf16_t b8v __attribute__((ext_vector_type(8)));   // MFMA operand of the bfloat16 instances (same register image as h8)

// n / d for n < 2^31 via a precomputed multiplier: q = (umulhi(mul, n) + n) >> shr   (round-up method)
__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned shr) {
  return (int)(((unsigned long long)__umulhi(mul, (unsigned)n) + (unsigned)n) >> shr);
}
static void fast_div_magic(unsigned d, unsigned* mul, unsigned* shr) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  *mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  *shr = l;
}
#define TF_OOB 0x80000000u   // voffset beyond every tensor: the buffer range check returns 0 -> zero padding in LDS

// LDS-DMA: 16 B per lane, LDS destination = wave-uniform base + lane*16; out-of-range lanes write zeros
__device__ __forceinline__ void bload_lds16(rsrc_t rsrc, unsigned voffset_bytes, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset_bytes, 0, 0, 0);
}

// s_waitcnt lgkmcnt(0) of the consumers' K loop as the BUILTIN (simm16 0xC07F: vmcnt 63, expcnt 7, lgkmcnt 0), not inline asm: the
// compiler's own wait-count pass cannot see inside an asm string, so with the asm form it assumed the fragments read one tile earlier
// could still be in flight and put s_waitcnt lgkmcnt(8 / 1 / 0) INSIDE the MFMA block -- which waits for the ds_reads of the NEXT tile
// issued just before it (LDS returns in order) and serialises the LDS reads with the MFMAs they were meant to hide under.
__device__ __forceinline__ void wait_lds_reads() {
  __builtin_amdgcn_s_waitcnt(0xC07F);
  asm volatile("" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// counted wait with a runtime stage count: leaves k * LPS of this wave's loads in flight (k clamped to [0, MAXK])
template <int LPS, int MAXK>
__device__ __forceinline__ void wait_stages(int k) {
  if constexpr (MAXK == 0) { wait_vm<0>(); }
  else {
    if (k >= MAXK) wait_vm<(MAXK * LPS > 63 ? 63 : MAXK * LPS)>();
    else wait_stages<LPS, MAXK - 1>(k);
  }
}

// ---- GroupNorm of the input inside the GEMM (GemmP::gi_*) ----------------------------------------------------------------
// LDS accesses that touch (or sit next to) LDS-DMA landing zones go through inline asm: for an LDS access the compiler cannot
// disambiguate from an outstanding LDS-DMA it inserts s_waitcnt vmcnt(0), which would drain the whole ring.
__device__ __forceinline__ unsigned lds_off(const char* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ h8 lds_read16(unsigned a) {
  h8 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ void lds_read16x2(unsigned a0, unsigned a1, h8& v0, h8& v1) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ void lds_write16(unsigned a, h8 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory"); }

// Prologue, run by the four CONSUMER waves (t = 0..255) while the loaders' first LDS-DMA stages are in flight: fold the statistics
// partials of image `img` into (mean, rstd) per group -- the very fold of k_gn_apply (8 lanes per group strided over the chunks,
// fp64, fixed order: the same bits) -- then a[c] = rstd * gamma[c], b[c] = beta[c] - mean * a[c] for every input channel.
// Contains two workgroup barriers (A: statistics in LDS, B: table in LDS); the loader waves execute the matching pair.
__device__ __forceinline__ void gi_prologue(const GemmP& p, char* smem, int img, int t) {
  f2* st = reinterpret_cast<f2*>(smem + p.gi_off);
  f2* ab = st + p.gi_G;
  const int G = p.gi_G, C = p.C, cpg = C / G, HW = p.H * p.W;
  const int sub = t & 7;
  for (int g0 = 0; g0 < G; g0 += 32) {
    const int g = g0 + (t >> 3);
    double S = 0.0, SS = 0.0;
    if (g < G) {
      const int nsub = p.gi_part2 ? p.gi_mr : 1;
      for (int j = 0; j < nsub; ++j) {
        const float* pp = p.gi_part + (long long)img * p.gi_chunks * G * 2 + g * 2;
        int nch = p.gi_chunks, gstride = G * 2;
        if (p.gi_part2) {
          const int sg = p.gi_mr * g + j;
          const bool first = sg < p.gi_G1;
          nch = first ? p.gi_chunks : p.gi_chunks2;
          gstride = (first ? p.gi_G1 : p.gi_G2) * 2;
          pp = (first ? p.gi_part : p.gi_part2) + (long long)img * nch * gstride + (first ? sg : sg - p.gi_G1) * 2;
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
      st[g] = (f2){(float)mean, (float)(1.0 / sqrt(var + (double)p.gi_eps))};
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                           // barrier A: (mean, rstd) of every group
  asm volatile("" ::: "memory");
  for (int c = t; c < C; c += 256) {
    f2 m = st[c / cpg];
    float gm = p.gi_gamma ? (float)p.gi_gamma[c] : 1.0f, bt = p.gi_beta ? (float)p.gi_beta[c] : 0.0f;
    float a = m[1] * gm;
    ab[c] = (f2){a, bt - m[0] * a};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                           // barrier B: the (a, b) table
  asm volatile("" ::: "memory");
}
// the (a, b) of the 8 consecutive channels c0 .. c0 + 7 from the LDS table (loader waves)
__device__ __forceinline__ void gi_load_ab(const GemmP& p, char* smem, int c0, float (&a)[8], float (&b)[8]) {
  const unsigned base = lds_off(smem + p.gi_off) + (unsigned)(p.gi_G + c0) * 8u;
  h8 r0, r1, r2, r3;
  lds_read16x2(base, base + 16, r0, r1);
  lds_read16x2(base + 32, base + 48, r2, r3);
  f4 q0 = __builtin_bit_cast(f4, r0), q1 = __builtin_bit_cast(f4, r1), q2 = __builtin_bit_cast(f4, r2), q3 = __builtin_bit_cast(f4, r3);
  a[0] = q0[0]; b[0] = q0[1]; a[1] = q0[2]; b[1] = q0[3];
  a[2] = q1[0]; b[2] = q1[1]; a[3] = q1[2]; b[3] = q1[3];
  a[4] = q2[0]; b[4] = q2[1]; a[5] = q2[2]; b[5] = q2[3];
  a[6] = q3[0]; b[6] = q3[1]; a[7] = q3[2]; b[7] = q3[3];
}
// normalise one 16-byte element vector: x * a + b, optional SiLU, rounded to fp16 exactly as k_gn_apply does; `keep` = false
// leaves zeros (zero padding of the convolution is applied AFTER the normalisation: the padded pixels must stay zero)
__device__ __forceinline__ h8 gi_apply(h8 x, const float (&a)[8], const float (&b)[8], int do_silu, bool keep) {
  h8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float f = (float)x[j] * a[j] + b[j];
    o[j] = (half_t)(do_silu ? silu_f(f) : f);
  }
  if (!keep) o = (h8){0, 0, 0, 0, 0, 0, 0, 0};
  return o;
}

// deep-variant ring depth: as many slots as 160 KiB of LDS hold (<= 8): LDS-DMA ingest is latency x bytes-in-flight bound
constexpr int ring_slots(int bm, int bn) { int s = 163840 / ((bm + bn) * 128); return s > 8 ? 8 : s; }

// GENERIC = false: every channel count is a multiple of 64, so a 64-wide K tile lies inside one filter tap and one
// concat source and (tap, channel) advance as wave-uniform scalars; GENERIC = true recomputes them per lane.
// ---- epilogue (consumer waves): write the wave's TM x TN fp32 tile through a per-wave row-major LDS scratch so
// that global stores / residual loads are 16-B coalesced row segments instead of MFMA-layout 8-B fragments.
template <int BM, int BN>
__device__ __forceinline__ void igemm_scratch_write(const GemmP& p, f4 (&acc)[BN / 32][BM / 32], const f4 (&csum)[BN / 32], char* smem, int w4, int lane) {
  constexpr int TM = BM / 2, TN = BN / 2, MJ = TM / 16, NI = TN / 16;
  constexpr int RS = TN + 4;                               // row stride (floats) keeps the f4 writes ~conflict-free
  const int lr = lane & 15, lg = lane >> 4;
  if (p.ln_colsum) {
    // LayerNorm fold: y = rstd[m] * (x . w'^T - mean[m] * colsum[n]); (mean, rstd) of every row of the block were
    // produced by the loader waves from the activation tiles they staged (LDS table behind the transpose scratch)
    const f2* stats = reinterpret_cast<const f2*>(smem + 4 * TM * RS * 4);
    const int wave_m = w4 & 1;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      f2 st_ = stats[wave_m * TM + j * 16 + lr];
      float mean = st_[0], rstd = st_[1];
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i][j] = rstd * (acc[i][j] - mean * csum[i]);
    }
  }
  float* sc = reinterpret_cast<float*>(smem) + (size_t)w4 * (TM * RS);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j)
      *reinterpret_cast<f4*>(sc + (j * 16 + lr) * RS + i * 16 + lg * 4) = acc[i][j];
}

// all 8 waves: wave (w4, half) stores rows [half*TM/2, (half+1)*TM/2) of consumer w4's tile
// LB: bias and the time embedding (bias_nc) come from an fp32 LDS table `lb` the kernel filled for its tile ([0][BN]: bias, [1 + i][BN]:
//   bias_nc of image lb_img0 + i, i < 2) instead of per-item global loads -- the only loads left in the epilogue are the residual's.
template <int BM, int BN, bool OUT8 = false, bool BF = false, int KBMAX = 4, bool LB = false>     // OUT8: the output is stored as e4m3 (fp8 kernels only; a template parameter keeps it out of the fp16 kernels); BF: bias / residual / output are bfloat16; KBMAX: items whose loads are in flight together
__device__ __forceinline__ void igemm_epilogue(const GemmP& p, char* smem, int m0, int n0, int split, int w4, int half, int lane, const float* lb = nullptr, int lb_n0 = 0, int lb_m1 = 0) {
  typedef typename std::conditional<BF, bf16_t, half_t>::type E;
  typedef E E8 __attribute__((ext_vector_type(8)));
  const E* const e_bias = reinterpret_cast<const E*>(p.bias);
  const E* const e_bias_nc = reinterpret_cast<const E*>(p.bias_nc);
  const E* const e_res = reinterpret_cast<const E*>(p.residual);
  E* const e_y = reinterpret_cast<E*>(p.y);
  constexpr int TM = BM / 2, TN = BN / 2;
  const int wave_m = w4 & 1, wave_n = w4 >> 1;
  constexpr int RS = TN + 4, ROWS = TM / 2;
  float* sc = reinterpret_cast<float*>(smem) + (size_t)w4 * (TM * RS) + (size_t)half * ROWS * RS;
  const int mb = m0 + wave_m * TM + half * ROWS;
  const int nb = n0 + wave_n * TN;                       // first (packed) column
  // Both hot paths below run in two passes over a wave's items (a fixed, small count: fully unrolled): pass 1 issues EVERY global load
  // (bias, time embedding, residual) of the wave, pass 2 consumes them.  Written as one loop, each iteration's loads sat behind the
  // previous iteration's store (they may alias as far as the compiler knows), i.e. up to five dependent L2 / HBM round trips per wave:
  // 7 us of a 256 x 160 tile's epilogue, 1-2 us of every short launch.
  if (p.act == 1) {
    // GEGLU: packed columns come in 16-wide blocks value|gate; out column = (n>>5)*16 + (n&15)
    constexpr int CPR = TN / 16;                          // 8-wide output chunks per row
    constexpr int ITEMS = ROWS * CPR, ITER = (ITEMS + 63) / 64, KB = ITER < KBMAX ? ITER : KBMAX;   // (batches of at most 4: 48 VGPRs of loads in flight)
    const int No = p.N >> 1;
#pragma unroll
    for (int k0 = 0; k0 < ITER; k0 += KB) {
    E8 ba[KB], bg[KB], rv[KB];
    bool ok[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int idx = lane + 64 * (k0 + k);
      const int row = idx / CPR, c8 = idx - row * CPR;
      const int m = mb + row, n = nb + 32 * (c8 >> 1) + 8 * (c8 & 1);
      ok[k] = k0 + k < ITER && idx < ITEMS && m < p.M && n < p.N;
      if (ok[k]) {
        if constexpr (!LB) { ba[k] = *reinterpret_cast<const E8*>(e_bias + n); bg[k] = *reinterpret_cast<const E8*>(e_bias + n + 16); }
        if (p.residual) rv[k] = *reinterpret_cast<const E8*>(e_res + (long long)m * No + ((n >> 5) * 16 + (n & 15)));
      }
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      if (!ok[k]) continue;
      const int idx = lane + 64 * (k0 + k);
      const int row = idx / CPR, c8 = idx - row * CPR;
      const int m = mb + row;
      const int pc = 32 * (c8 >> 1) + 8 * (c8 & 1);      // packed column of the value chunk inside the wave tile
      const int n = nb + pc;
      const int no = (n >> 5) * 16 + (n & 15);
      const float* r = sc + row * RS + pc;
      f4 a0 = *reinterpret_cast<const f4*>(r), a1 = *reinterpret_cast<const f4*>(r + 4);
      f4 g0 = *reinterpret_cast<const f4*>(r + 16), g1 = *reinterpret_cast<const f4*>(r + 20);
      E8 o;
      if constexpr (LB) {
        const float* t = lb + (n - lb_n0);
        f4 b0 = *reinterpret_cast<const f4*>(t), b1 = *reinterpret_cast<const f4*>(t + 4), c0 = *reinterpret_cast<const f4*>(t + 16), c1 = *reinterpret_cast<const f4*>(t + 20);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (E)((a0[e] + b0[e]) * gelu_f(g0[e] + c0[e]));
          o[4 + e] = (E)((a1[e] + b1[e]) * gelu_f(g1[e] + c1[e]));
        }
      } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (E)((a0[e] + (float)ba[k][e]) * gelu_f(g0[e] + (float)bg[k][e]));
        o[4 + e] = (E)((a1[e] + (float)ba[k][4 + e]) * gelu_f(g1[e] + (float)bg[k][4 + e]));
      }
      }
      if (p.residual) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (E)((float)o[e] + (float)rv[k][e]);
      }
      if constexpr (OUT8) {
        f4 q0, q1;
        for (int e = 0; e < 4; ++e) { q0[e] = (float)o[e]; q1[e] = (float)o[4 + e]; }
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.y) + (long long)m * No + no) = pack8_fp8(q0, q1);
      } else {
        *reinterpret_cast<E8*>(e_y + (long long)m * No + no) = o;
      }
    }
    }
    return;
  }
  if ((p.N & 7) == 0 && p.splitk <= 1 && !p.out32) {
    // the common case: 16-byte rows segments of an fp16 (or e4m3) output with bias + time embedding + residual
    constexpr int CPR = TN / 8;
    constexpr int ITEMS = ROWS * CPR, ITER = (ITEMS + 63) / 64, KB = ITER < KBMAX ? ITER : KBMAX;
#pragma unroll
    for (int k0 = 0; k0 < ITER; k0 += KB) {
    E8 bv[KB], cv[KB], rv[KB];
    bool ok[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int idx = lane + 64 * (k0 + k);
      const int row = idx / CPR, c8 = idx - row * CPR;
      const int m = mb + row, n = nb + c8 * 8;
      ok[k] = k0 + k < ITER && idx < ITEMS && m < p.M && n < p.N;
      if (ok[k]) {
        const long long o = (long long)m * p.N + n;
        if constexpr (!LB) {
          if (p.bias) bv[k] = *reinterpret_cast<const E8*>(e_bias + n);
          if (p.bias_nc) cv[k] = *reinterpret_cast<const E8*>(e_bias_nc + (long long)(m / p.HoWo) * p.bias_nc_stride + n);
        }
        if (p.residual) rv[k] = *reinterpret_cast<const E8*>(e_res + o);
      }
    }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      if (!ok[k]) continue;
      const int idx = lane + 64 * (k0 + k);
      const int row = idx / CPR, c8 = idx - row * CPR;
      const int m = mb + row, n = nb + c8 * 8;
      float* r = sc + row * RS + c8 * 8;
      f4 v0 = *reinterpret_cast<const f4*>(r), v1 = *reinterpret_cast<const f4*>(r + 4);
      const long long o = (long long)m * p.N + n;
      if constexpr (LB) {
        const float* t = lb + (n - lb_n0);
        if (p.bias) { v0 += *reinterpret_cast<const f4*>(t); v1 += *reinterpret_cast<const f4*>(t + 4); }
        if (p.bias_nc) { const float* u = t + (m >= lb_m1 ? 2 * BN : BN); v0 += *reinterpret_cast<const f4*>(u); v1 += *reinterpret_cast<const f4*>(u + 4); }
      } else {
      if (p.bias) { for (int e = 0; e < 4; ++e) { v0[e] += (float)bv[k][e]; v1[e] += (float)bv[k][4 + e]; } }
      if (p.bias_nc) { for (int e = 0; e < 4; ++e) { v0[e] += (float)cv[k][e]; v1[e] += (float)cv[k][4 + e]; } }
      }
      if (p.residual) { for (int e = 0; e < 4; ++e) { v0[e] += (float)rv[k][e]; v1[e] += (float)rv[k][4 + e]; } }
      E8 out;
      for (int e = 0; e < 4; ++e) { out[e] = (E)v0[e]; out[4 + e] = (E)v1[e]; }
      if constexpr (OUT8) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.y) + o) = pack8_fp8(v0, v1);
      else *reinterpret_cast<E8*>(e_y + o) = out;
      if (p.gn_part) {   // the statistics pass sums what the consumer will read: the fp16-rounded outputs
        for (int e = 0; e < 4; ++e) { v0[e] = (float)out[e]; v1[e] = (float)out[4 + e]; }
        *reinterpret_cast<f4*>(r) = v0; *reinterpret_cast<f4*>(r + 4) = v1;
      }
    }
    }
    return;
  }
  constexpr int CPR = TN / 8;
  const bool vec = (p.N & 7) == 0;
  float* part = p.splitk > 1 ? p.partial + (long long)split * p.M * p.N : p.out32;
  for (int idx = lane; idx < ROWS * CPR; idx += 64) {
    int row = idx / CPR, c8 = idx - row * CPR;
    int m = mb + row, n = nb + c8 * 8;
    if (m >= p.M || n >= p.N) continue;
    const float* r = sc + row * RS + c8 * 8;
    f4 v0 = *reinterpret_cast<const f4*>(r), v1 = *reinterpret_cast<const f4*>(r + 4);
    const long long o = (long long)m * p.N + n;
    if (part) {
      if (vec) { *reinterpret_cast<f4*>(part + o) = v0; *reinterpret_cast<f4*>(part + o + 4) = v1; }
      else { for (int e = 0; e < 8 && n + e < p.N; ++e) part[o + e] = e < 4 ? v0[e] : v1[e - 4]; }
      continue;
    }
    const long long bo = p.bias_nc ? (long long)(m / p.HoWo) * p.bias_nc_stride + n : 0;
    if (vec) {
      if (p.bias) { E8 b = *reinterpret_cast<const E8*>(e_bias + n); for (int e = 0; e < 4; ++e) { v0[e] += (float)b[e]; v1[e] += (float)b[4 + e]; } }
      if (p.bias_nc) { E8 b = *reinterpret_cast<const E8*>(e_bias_nc + bo); for (int e = 0; e < 4; ++e) { v0[e] += (float)b[e]; v1[e] += (float)b[4 + e]; } }
      if (p.residual) { E8 b = *reinterpret_cast<const E8*>(e_res + o); for (int e = 0; e < 4; ++e) { v0[e] += (float)b[e]; v1[e] += (float)b[4 + e]; } }
      E8 out;
      for (int e = 0; e < 4; ++e) { out[e] = (E)v0[e]; out[4 + e] = (E)v1[e]; }
      if constexpr (OUT8) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.y) + o) = pack8_fp8(v0, v1);
      else *reinterpret_cast<E8*>(e_y + o) = out;
      if (p.gn_part) {   // the statistics pass below sums what the consumer will read: the fp16-rounded outputs
        float* rw = sc + row * RS + c8 * 8;
        for (int e = 0; e < 4; ++e) { v0[e] = (float)out[e]; v1[e] = (float)out[4 + e]; }
        *reinterpret_cast<f4*>(rw) = v0; *reinterpret_cast<f4*>(rw + 4) = v1;
      }
    } else {
      for (int e = 0; e < 8 && n + e < p.N; ++e) {
        float f = e < 4 ? v0[e] : v1[e - 4];
        if (p.bias) f += (float)e_bias[n + e];
        if (p.bias_nc) f += (float)e_bias_nc[bo + e];
        if (p.residual) f += (float)e_res[o + e];
        e_y[o + e] = (E)f;
      }
    }
  }
}

// ---- GroupNorm statistics of the block's output tile (all 8 waves, after igemm_epilogue left the rounded outputs in
// the scratch).  Fixed summation order everywhere -> bitwise reproducible:
//   1. every wave sums its ROWS x TN region by columns (lane = column: conflict-free ds_read_b32 down the rows);
//   2. the per-(row stripe, channel) sums meet in an LDS table; one barrier;
//   3. one wave per group touched by the tile (lane = channel, 4 stripe reads, xor-shuffle tree) writes the partial for
//      (image, chunk = 2 * m-tile + piece, group).  A group that straddles two n-tiles (cpg <= 64 <= BN: at most two)
//      gets piece 0 from the tile holding its first channel and piece 1 from the next; a tile that holds a whole
//      group writes piece 1 = 0 itself, so every slot has exactly one writer and no zero-fill is needed.
template <int BM, int BN>
__device__ __forceinline__ void igemm_gn_stats(const GemmP& p, char* smem, int m0, int n0, int w4, int half, int lane) {
  constexpr int TM = BM / 2, TN = BN / 2, RS = TN + 4, ROWS = TM / 2;
  const int wave_m = w4 & 1, wave_n = w4 >> 1;
  const float* sc = reinterpret_cast<const float*>(smem) + (size_t)w4 * (TM * RS) + (size_t)half * ROWS * RS;
  f2* cs = reinterpret_cast<f2*>(smem + 4 * TM * RS * 4 + BM * 8);       // [4 stripes][BN]
  const int stripe = wave_m * 2 + half;
  const int ncols = min(p.N - n0, BN);
  for (int c = lane; c < TN; c += 64) {
    float s_ = 0.f, q_ = 0.f;
    if (wave_n * TN + c < ncols) {
#pragma unroll 8
      for (int r = 0; r < ROWS; ++r) { float v = sc[r * RS + c]; s_ += v; q_ += v * v; }
    }
    cs[stripe * BN + wave_n * TN + c] = (f2){s_, q_};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                           // barrier W: the column sums of all 8 waves are in LDS
  asm volatile("" ::: "memory");
  // one WAVE per group (round-robin over the 8 waves): lane = channel of the group, 4 stripe reads, shuffle tree
  const int cpg = p.gn_cpg;
  const int g_lo = n0 / cpg, g_hi = (n0 + ncols - 1) / cpg;
  const int img = m0 / p.HoWo, mt = (m0 - img * p.HoWo) / BM;
  for (int g = g_lo + (w4 + 4 * half); g <= g_hi; g += 8) {
    const int cb = max(g * cpg, n0) - n0, ce = min((g + 1) * cpg, n0 + ncols) - n0;
    float S = 0.f, Q = 0.f;
    if (cb + lane < ce) {
#pragma unroll
      for (int st_ = 0; st_ < 4; ++st_) { f2 v = cs[st_ * BN + cb + lane]; S += v[0]; Q += v[1]; }
    }
    S = wave_sum(S); Q = wave_sum(Q);
    if (lane != 0) continue;
    if (BN % cpg == 0) {                                  // groups never straddle n-tiles: one chunk per m-tile
      *reinterpret_cast<f2*>(p.gn_part + ((long long)(img * p.gn_chunks + mt) * p.gn_G + g) * 2) = (f2){S, Q};
      continue;
    }
    float* dst = p.gn_part + ((long long)(img * p.gn_chunks + 2 * mt) * p.gn_G + g) * 2;
    const bool starts = g * cpg >= n0, ends = (g + 1) * cpg <= n0 + ncols;
    if (starts) {
      *reinterpret_cast<f2*>(dst) = (f2){S, Q};
      if (ends) *reinterpret_cast<f2*>(dst + p.gn_G * 2) = (f2){0.f, 0.f};
    } else {
      *reinterpret_cast<f2*>(dst + p.gn_G * 2) = (f2){S, Q};
    }
  }
}

// GENERIC = false: every channel count is a multiple of 64, so a 64-wide K tile lies inside one filter tap and one
// concat source and (tap, channel) advance as wave-uniform scalars; GENERIC = true recomputes them per lane.
//
// 8 waves with split roles: waves 0-3 are CONSUMERS (2 x 2 wave tiles: ds_read_b128 fragments + MFMA), waves 4-7 are
// LOADERS (LDS-DMA only).  One consumer and one loader share each SIMD, so the loader's LDS-DMA issue stalls (~60-100
// cycles per 1-KiB piece) never hold up MFMA issue; a single s_barrier per K tile hands ring slots back and forth.
// WIDE = true is the short-K variant: 2-slot ring, one fragment set, <= 128 VGPRs, so TWO blocks share a CU and one
// block's prologue / epilogue overlaps the other's K loop (shapes with many tiles and few K tiles per tile);
// WIDE = false is the deep variant: 4-slot ring, fragments of tile t+1 prefetched during tile t, one block per CU.
// ALL8 = true (deep variant only): the consumer waves issue LPC of the WEIGHT pieces of every stage themselves.  Data that is
// not L2 resident (each layer's weights arrive cold from HBM / Infinity Cache) streams at a rate set by the number of waves
// that have loads outstanding, not by the pieces each keeps in flight (tools/ingest_waves.hip: 28 GB/s per CU with 4
// issuing waves, 44-52 GB/s with 8), so the weight-bound shapes gain from eight issuing waves what the L2-resident ones
// lose in MFMA issue slots; one of the autotuned variants.
// GI = true: the instance that can normalise its input (GemmP::gi_*); a template parameter so that the launches without it run the very
// code they ran before the feature existed (its branches and SGPRs cost 3 % of the step when they sat in every instance)
template <int BM, int BN, bool GENERIC, bool WIDE, bool ALL8 = false, bool GI = false, bool BF = false>   // BF: bfloat16 operands / outputs (plain deep ring only)
__global__ void __launch_bounds__(512, WIDE ? 4 : 2) k_igemm(const GemmP p) {
  static_assert(!BF || (!WIDE && !ALL8 && !GI && BM != 256), "the bfloat16 instances use the plain deep ring");
  constexpr int TM = BM / 2, TN = BN / 2, MJ = TM / 16, NI = TN / 16;
  constexpr int NG = (BM + BN) / 8;                       // 8-row staging groups: activation rows first, then weight rows
  // weight pieces per CONSUMER wave per stage (the last 4 LPC groups); one more per wave measured 1-3 % slower on every shape
  constexpr int LPC = ALL8 ? (BN >= 128 ? (BM + BN >= 256 ? 3 : 2) : 1) : 0;
  constexpr int LPS = NG / 4 - LPC;                       // LDS-DMA pieces per loader wave per stage
  static_assert(!(ALL8 && (WIDE || GENERIC)), "ALL8 is a deep-ring, 64-channel-aligned variant");
  static_assert(4 * LPC <= BN / 8, "the consumers take weight groups only");
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NS = WIDE ? 2 : ring_slots(BM, BN);      // ring slots
  static_assert(NG % 4 == 0 && BM % 32 == 0 && BN % 32 == 0, "tile shape");
  static_assert((NS - 2) * LPS <= 63 || WIDE, "vmcnt immediate is 6 bits");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wid >= 4;
  const int w4 = wid & 3;
  // XCD-aware work order: blocks b and b+8 share an XCD (and its L2), so every XCD gets a contiguous run of work items
  // (bijective remap).  Inside a run either n is fastest (neighbours re-use the same activation rows and sweep the
  // weight tiles) or m is fastest (neighbours share one weight tile: each weight byte leaves HBM / Infinity Cache once);
  // the host picks the order per shape (it is one of the autotuned knobs).  Speed only: any order is correct.
  const int ntiles = p.ntm * p.ntn;
  const int nblk = ntiles * p.splitk;
  int bid = blockIdx.x;
  {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = bid / ntiles;
  const int tid_ = bid - split * ntiles;
  int tile_m, tile_n;
  if (p.order == 0) { tile_m = tid_ / p.ntn; tile_n = tid_ - tile_m * p.ntn; }
  else { tile_n = tid_ / p.ntm; tile_m = tid_ - tile_n * p.ntm; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt_begin = split * p.ktiles_per_split;
  const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);

  if (loader) {
    // =============================== LOADER WAVES ===============================================
    const rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x), 0, p.x2_bytes, 0x00020000);
    const rsrc_t rs_x3 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x3 ? p.x3 : p.x), 0, p.x3_bytes, 0x00020000);
    const rsrc_t rs_x4 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x4 ? p.x4 : p.x), 0, p.x4_bytes, 0x00020000);
    const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    // loader wave w4 owns groups g = w4 + 4 i; lane -> row 8 g + (lane >> 3), 16-B chunk lane & 7.
    // XOR swizzle on the SOURCE chunk (LDS image stays lane-linear): chunk ^ ((row >> 1) & 7); g = w4 (mod 4), so the
    // swizzle term (4 (g & 1) + (sub >> 1)) & 7 is a per-thread constant.
    const int sub = lane >> 3;
    const int cs = (lane & 7) ^ ((4 * (w4 & 1) + (sub >> 1)) & 7);
    int g_a[LPS], g_b[LPS], g_c[LPS];                     // A row: (hi0, wi0, pixel base); W row: (-, -, byte offset)
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
      const int row = 8 * (w4 + 4 * i) + sub;
      g_a[i] = -(1 << 28); g_b[i] = 0; g_c[i] = (int)TF_OOB;
      if (row < BM) {
        int m = m0 + row;
        if (m < p.M) {
          int img = fast_div(m, p.dv_howo_mul, p.dv_howo_shr), rem = m - img * p.HoWo;
          int ho = fast_div(rem, p.dv_wo_mul, p.dv_wo_shr), wo = rem - ho * p.Wo;
          g_a[i] = ho * p.stride - p.pad;
          g_b[i] = wo * p.stride - p.pad;
          g_c[i] = img * p.H * p.W;
        }
      } else {
        int n = n0 + row - BM;
        if (n < p.N) g_c[i] = (int)((unsigned)(n * p.K + cs * 8) * 2u);
      }
    }
    const int Hl = p.H << p.ups, Wl = p.W << p.ups;        // logical (post-upsample) input extent
    // LayerNorm fold: this wave also sums (x, x^2) over the activation rows it staged (it reads back its own LDS-DMA
    // pieces once they have landed: idle VALU of the loaders, nothing added to the consumers' MFMA stream)
    const bool ln_on = p.ln_colsum != nullptr;
    float ls[LPS], lq[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) { ls[i] = 0.f; lq[i] = 0.f; }
    auto ln_tile = [&](int slot) {
      typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
      const hh2 one2 = {(_Float16)1.0f, (_Float16)1.0f};
      const char* base = smem + slot * STAGE;
#pragma unroll
      for (int i = 0; i < LPS; ++i) {
        const int g = w4 + 4 * i;
        if (g * 8 < BM) {
          h8 x = *reinterpret_cast<const h8*>(base + g * 1024 + lane * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hh2 v = {x[2 * e], x[2 * e + 1]};
            ls[i] = __builtin_amdgcn_fdot2(v, one2, ls[i], false);
            lq[i] = __builtin_amdgcn_fdot2(v, v, lq[i], false);
          }
        }
      }
    };
    // GroupNorm of the input (1x1 convolutions: k = input channel): normalise this wave's activation pieces of K tile kt where
    // they landed (LDS position lane -> source chunk cs), before the barrier that hands the tile to the consumers
    constexpr bool gi_on = GI;
    auto gi_tile = [&](int slot, int kt) {
      if (!GI || kt * 64 >= p.Kc) return;                  // the extra 1x1 segment stays raw
      float ga[8], gb[8];
      gi_load_ab(p, smem, kt * 64 + cs * 8, ga, gb);
      const unsigned base = lds_off(smem + slot * STAGE) + lane * 16;
#pragma unroll
      for (int i = 0; i < LPS; ++i) {
        const int g = w4 + 4 * i;
        if (g * 8 < BM) {
          h8 x = lds_read16(base + g * 1024);
          lds_write16(base + g * 1024, gi_apply(x, ga, gb, p.gi_silu, true));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    int st_r, st_s, st_c;                                  // wave-uniform (tap, channel) of the next tile to stage
    {
      int kg0 = kt_begin * 64;
      if (kg0 < p.Kc) {
        int tap = kg0 / p.C;
        st_c = kg0 - tap * p.C;
        st_r = tap / p.S;
        st_s = tap - st_r * p.S;
      } else {                                             // inside the extra 1x1 segment: st_r = -1 marks it
        st_r = -1; st_s = 0; st_c = kg0 - p.Kc;
      }
    }
    auto stage = [&](int buf, int kt) {
      if (p.dbg & 4) return;
      char* base = smem + buf * STAGE;
      int r, s_, cc, ld;
      bool kvalid = true, second, extra;
      if (GENERIC) {
        int kg = kt * 64 + cs * 8;
        kvalid = kg < p.K;
        extra = kg >= p.Kc;
        if (!extra) {
          int tap = kg / p.C;
          int c = kg - tap * p.C;
          r = tap / p.S; s_ = tap - r * p.S;
          second = c >= p.C1;
          ld = second ? p.C2 : p.C1;
          cc = second ? c - p.C1 : c;
        } else {
          int c = kg - p.Kc;
          r = p.pad; s_ = p.pad;                           // the output pixel itself: hi = ho * stride
          second = c >= p.C3;
          ld = second ? p.C4 : p.C3;
          cc = second ? c - p.C3 : c;
        }
      } else {
        extra = st_r < 0;                                  // all wave-uniform (SGPR)
        if (!extra) {
          r = st_r; s_ = st_s;
          second = st_c >= p.C1;
          ld = second ? p.C2 : p.C1;
          cc = (second ? st_c - p.C1 : st_c) + cs * 8;
          st_c += 64;
          if (st_c >= p.C) { st_c = 0; if (++st_s == p.S) { st_s = 0; if ((++st_r) * p.S * p.C >= p.Kc) st_r = -1; } }
        } else {
          r = p.pad; s_ = p.pad;
          second = st_c >= p.C3;
          ld = second ? p.C4 : p.C3;
          cc = (second ? st_c - p.C3 : st_c) + cs * 8;
          st_c += 64;
        }
      }
      const unsigned kb = (unsigned)kt * 128u;
      // non-GENERIC: the source tensor of this K tile is wave-uniform -> ONE descriptor built here from the argument
      // block instead of four kept alive for the whole kernel (the kernel is SGPR-bound: 106 of 106)
      rsrc_t rs_a = rs_w;
      if (!GENERIC) {
        const half_t* sp = extra ? (second ? p.x4 : p.x3) : (second ? p.x2 : p.x);
        const unsigned sbytes = extra ? (second ? p.x4_bytes : p.x3_bytes) : (second ? p.x2_bytes : p.x_bytes);
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)sp, 0, sbytes, 0x00020000);
      }
#pragma unroll
      for (int i = 0; i < LPS; ++i) {
        const int g = w4 + 4 * i;                          // wave-uniform
        char* dst = base + g * 1024;
        if (g * 8 < BM) {
          int hi = g_a[i] + r, wi = g_b[i] + s_;
          bool ok = kvalid && (unsigned)hi < (unsigned)Hl && (unsigned)wi < (unsigned)Wl;
          int pix = g_c[i] + (hi >> p.ups) * p.W + (wi >> p.ups);
          unsigned off = ok ? (unsigned)(pix * ld + cc) * 2u : TF_OOB;
          if (GENERIC) {                                   // per-lane source: one masked issue per descriptor
            if (extra) { if (second) bload_lds16(rs_x4, off, dst); else bload_lds16(rs_x3, off, dst); }
            else { if (second) bload_lds16(rs_x2, off, dst); else bload_lds16(rs_x, off, dst); }
          } else {
            bload_lds16(rs_a, off, dst);                   // this K tile's (wave-uniform) source
          }
        } else {
          unsigned wo = (unsigned)g_c[i];
          unsigned off = (kvalid && wo != TF_OOB) ? wo + kb : TF_OOB;
          bload_lds16(rs_w, off, dst);
        }
      }
    };
    const int nt = kt_end - kt_begin;
    if (WIDE) {
      // 2-slot ring: barrier(it) hands tile it to the consumers and slot (it-1) % 2 back; tile it+1 is in flight
      // while tile it is multiplied.
      if (nt > 0) stage(0, kt_begin);
      if (gi_on) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }   // barriers A, B of gi_prologue (consumer waves)
      for (int it = 0; it < nt; ++it) {
        wait_vm<0>();
        if (gi_on) gi_tile(it & 1, kt_begin + it);
        __builtin_amdgcn_s_barrier();                     // barrier(it)
        asm volatile("" ::: "memory");
        if (it + 1 < nt) stage((it + 1) & 1, kt_begin + it + 1);
        if (ln_on) ln_tile(it & 1);                       // off the barrier's critical path; slot refilled after barrier(it+1)
      }
    } else {
      // Ring protocol (NS slots, tile t lives in slot t % NS).  Barrier P hands tile 0 to the consumers; barrier(it)
      // guarantees tile it+1 has landed (the consumers prefetch its fragments while multiplying tile it) and hands
      // slot it % NS back (the consumers drained their reads of tile it before arriving).  NS-1 tiles stay in flight.
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_)
        if (s_ < nt) stage(s_, kt_begin + s_);
      if (gi_on) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }   // barriers A, B of gi_prologue (consumer waves)
      wait_stages<LPS, NS - 1>(nt - 1);                    // tile 0 landed; up to NS-1 newer stages in flight
      if (gi_on && nt > 0) gi_tile(0, kt_begin);
      __builtin_amdgcn_s_barrier();                       // barrier P
      if (ln_on && nt > 0) ln_tile(0);                    // slot 0 is refilled only after barrier(0)
      asm volatile("" ::: "memory");
      for (int it = 0; it < nt; ++it) {
        if (it + 1 < nt) wait_stages<LPS, NS - 2>(nt - 2 - it);   // tile it+1 landed (ring holds up to tile it+NS-1 here)
        if (gi_on && it + 1 < nt) gi_tile((it + 1) % NS, kt_begin + it + 1);
        __builtin_amdgcn_s_barrier();                     // barrier(it)
        asm volatile("" ::: "memory");
        if (it + NS < nt) stage(it % NS, kt_begin + it + NS);
        if (ln_on && it + 1 < nt) ln_tile((it + 1) % NS);  // tile it+1 stays in its slot until barrier(it+1)
      }
    }
    // LayerNorm fold: finish (mean, rstd) of the rows this wave staged while the consumers drain their last MFMAs
    f2 lstat[LPS];
    if (ln_on) {
      const float invK = 1.0f / (float)p.K;
#pragma unroll
      for (int i = 0; i < LPS; ++i) {
        float s_ = ls[i], q_ = lq[i];
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { s_ += __shfl_xor(s_, o, 64); q_ += __shfl_xor(q_, o, 64); }
        float mean = s_ * invK;
        float var = fmaxf(q_ * invK - mean * mean, 0.f);
        lstat[i] = (f2){mean, rsqrtf(var + p.ln_eps)};
      }
    }
    __builtin_amdgcn_s_barrier();                         // barrier X: matches the consumers' "ring is free" barrier
    asm volatile("" ::: "memory");
    if (p.dbg & 1) return;
    if (ln_on) {
      constexpr int TMl = BM / 2, TNl = BN / 2;
      f2* stats = reinterpret_cast<f2*>(smem + 4 * TMl * (TNl + 4) * 4);
#pragma unroll
      for (int i = 0; i < LPS; ++i) {
        const int g = w4 + 4 * i;
        if (g * 8 < BM && (lane & 7) == 0) stats[8 * g + sub] = lstat[i];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // barrier Z
      asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                         // barrier Y: the consumers' tiles are in the LDS scratch
    asm volatile("" ::: "memory");
    igemm_epilogue<BM, BN, false, BF>(p, smem, m0, n0, split, w4, 1, lane);
    if (p.gn_part) igemm_gn_stats<BM, BN>(p, smem, m0, n0, w4, 1, lane);
    return;
  }

  // ================================= CONSUMER WAVES ===============================================
  if constexpr (GI) gi_prologue(p, smem, m0 / p.HoWo, w4 * 64 + lane);   // first: its global loads must not wait behind this wave's own DMA (ALL8)
  const int wave_m = w4 & 1, wave_n = w4 >> 1;
  const int lr = lane & 15, lg = lane >> 4;
  f4 acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  // LayerNorm fold: this lane's colsum values, fetched now so their latency hides under the K loop -- except on the 128x160 tile,
  // which sits at the 256-VGPR limit (20 registers held for the whole K loop made its ALL8 and GENERIC forms spill): there they are
  // fetched in the epilogue (no LayerNorm-folded shape of the step runs that tile)
  constexpr bool CSUM_LATE = BM == 128 && BN == 160;
  f4 csum[NI];
  auto load_csum = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int n = n0 + wave_n * TN + i * 16 + lg * 4;
      csum[i] = (p.ln_colsum && n + 3 < p.N) ? *reinterpret_cast<const f4*>(p.ln_colsum + n) : (f4){0.f, 0.f, 0.f, 0.f};
    }
  };
  if constexpr (!CSUM_LATE) load_csum();
  // fragment addresses inside a stage (swizzled chunk for k-step 0; k-step 1 is chunk ^ 4)
  int wa[NI], xa[MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int row = wave_n * TN + i * 16 + lr;
    wa[i] = BM * 128 + row * 128 + ((lg ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    int row = wave_m * TM + j * 16 + lr;
    xa[j] = row * 128 + ((lg ^ ((row >> 1) & 7)) << 4);
  }
  const int nt = kt_end - kt_begin;
  // ALL8: this wave's share of the weight rows (groups NG - 4 LPC + w4 + 4 i), same lane -> (row, swizzled chunk) map as the loaders
  unsigned cw[LPC > 0 ? LPC : 1];
  rsrc_t rs_cw;
  if constexpr (ALL8) {
    rs_cw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const int sub = lane >> 3;
    const int cs = (lane & 7) ^ ((4 * (w4 & 1) + (sub >> 1)) & 7);
#pragma unroll
    for (int i = 0; i < LPC; ++i) {
      int n = n0 + 8 * (NG - 4 * LPC + w4 + 4 * i) + sub - BM;
      cw[i] = n < p.N ? (unsigned)(n * p.K + cs * 8) * 2u : TF_OOB;
    }
  }
  auto cstage = [&](int buf, int kt) {                    // the consumer's pieces of stage (buf, kt)
    if constexpr (ALL8) {
      char* base = smem + buf * STAGE;
#pragma unroll
      for (int i = 0; i < LPC; ++i)
        bload_lds16(rs_cw, cw[i] != TF_OOB ? cw[i] + (unsigned)kt * 128u : TF_OOB, base + (NG - 4 * LPC + w4 + 4 * i) * 1024);
    }
  };
  if constexpr (BM == 256) {
    // 256-row tile (large problems: every CU still gets tiles): the accumulators take 128 VGPRs, so the fragments are pipelined per
    // 32-deep k-step instead of per K tile -- set A holds k-step 0, set B k-step 1 (48 VGPRs each): while the 32 MFMAs of one set issue,
    // the 12 ds_read_b128 of the other are in flight.  Same barrier protocol as the deep ring (one per K tile).
    static_assert(!WIDE && !ALL8 && !GENERIC && !GI, "the 256-row tile has the plain deep ring only");
    h8 wfA[NI], xfA[MJ], wfB[NI], xfB[MJ];
    auto read_k = [&](int slot, int k2, h8 (&wf)[NI], h8 (&xf)[MJ]) {
      const char* sb = smem + slot * STAGE;
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const h8*>(sb + (wa[i] ^ (k2 * 64)));
#pragma unroll
      for (int j = 0; j < MJ; ++j) xf[j] = *reinterpret_cast<const h8*>(sb + (xa[j] ^ (k2 * 64)));
    };
    auto mma1 = [&](h8 (&wf)[NI], h8 (&xf)[MJ]) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    };
    __builtin_amdgcn_s_barrier();                         // barrier P: tile 0 landed
    asm volatile("" ::: "memory");
    if (nt > 0) read_k(0, 0, wfA, xfA);
    for (int it = 0; it < nt; ++it) {
      read_k(it % NS, 1, wfB, xfB);                       // (it, k-step 1) in flight under the MFMAs of (it, k-step 0)
      __builtin_amdgcn_sched_barrier(0);
      mma1(wfA, xfA);
      __builtin_amdgcn_sched_barrier(0);
      wait_lds_reads();  // every fragment of tile it is in registers: its slot may be refilled
      __builtin_amdgcn_s_barrier();                       // barrier(it): tile it+1 landed
      asm volatile("" ::: "memory");
      if (it + 1 < nt) read_k((it + 1) % NS, 0, wfA, xfA);
      __builtin_amdgcn_sched_barrier(0);
      mma1(wfB, xfB);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (WIDE) {
    for (int it = 0; it < nt; ++it) {
      __builtin_amdgcn_s_barrier();                       // barrier(it): tile it landed
      asm volatile("" ::: "memory");
      const char* sb = smem + (it & 1) * STAGE;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        h8 wf[NI], xf[MJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const h8*>(sb + (wa[i] ^ (k2 * 64)));
#pragma unroll
        for (int j = 0; j < MJ; ++j) xf[j] = *reinterpret_cast<const h8*>(sb + (xa[j] ^ (k2 * 64)));
        if (p.dbg & 2) {
#pragma unroll
          for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(wf[i]));
#pragma unroll
          for (int j = 0; j < MJ; ++j) asm volatile("" ::"v"(xf[j]));
          continue;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);

      }
    }
  } else {
  // fragments of the current and of the next K tile (software pipeline across the barrier: the ds_reads of tile t+1
  // are in flight while the MFMAs of tile t issue)
  h8 wfA[2][NI], xfA[2][MJ], wfB[2][NI], xfB[2][MJ];
  auto read_frags = [&](int slot, h8 (&wf)[2][NI], h8 (&xf)[2][MJ]) {
    const char* sb = smem + slot * STAGE;
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[0][i] = *reinterpret_cast<const h8*>(sb + wa[i]);
#pragma unroll
    for (int j = 0; j < MJ; ++j) xf[0][j] = *reinterpret_cast<const h8*>(sb + xa[j]);
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[1][i] = *reinterpret_cast<const h8*>(sb + (wa[i] ^ 64));
#pragma unroll
    for (int j = 0; j < MJ; ++j) xf[1][j] = *reinterpret_cast<const h8*>(sb + (xa[j] ^ 64));
  };
  auto mma = [&](h8 (&wf)[2][NI], h8 (&xf)[2][MJ]) {
    if (p.dbg & 2) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
        for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(wf[k2][i]));
#pragma unroll
        for (int j = 0; j < MJ; ++j) asm volatile("" ::"v"(xf[k2][j]));
      }
      return;
    }
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          if constexpr (BF) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, wf[k2][i]), __builtin_bit_cast(b8v, xf[k2][j]), acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[k2][i], xf[k2][j], acc[i][j], 0, 0, 0);
        }
  };
  if constexpr (ALL8) {
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_)
      if (s_ < nt) cstage(s_, kt_begin + s_);
    wait_stages<LPC, NS - 1>(nt - 1);                      // this wave's pieces of tile 0 landed
  }
  __builtin_amdgcn_s_barrier();                           // barrier P: tile 0 landed
  asm volatile("" ::: "memory");
  if (nt > 0) read_frags(0, wfA, xfA);
  for (int it = 0; it < nt; it += 2) {
    wait_lds_reads();    // fragments of tile it are in registers: its slot may be refilled
    if constexpr (ALL8) { if (it + 1 < nt) wait_stages<LPC, NS - 2>(nt - 2 - it); }   // ... and this wave's pieces of tile it+1 landed
    __builtin_amdgcn_s_barrier();                         // barrier(it): tile it+1 landed
    asm volatile("" ::: "memory");
    if constexpr (ALL8) { if (it + NS < nt) cstage(it % NS, kt_begin + it + NS); }
    if (it + 1 < nt) read_frags((it + 1) % NS, wfB, xfB);
    __builtin_amdgcn_sched_barrier(0);
    mma(wfA, xfA);
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 >= nt) break;
    wait_lds_reads();
    if constexpr (ALL8) { if (it + 2 < nt) wait_stages<LPC, NS - 2>(nt - 3 - it); }
    __builtin_amdgcn_s_barrier();                         // barrier(it+1)
    asm volatile("" ::: "memory");
    if constexpr (ALL8) { if (it + 1 + NS < nt) cstage((it + 1) % NS, kt_begin + it + 1 + NS); }
    if (it + 2 < nt) read_frags((it + 2) % NS, wfA, xfA);
    __builtin_amdgcn_sched_barrier(0);
    mma(wfB, xfB);
    __builtin_amdgcn_sched_barrier(0);
  }
  }
  __builtin_amdgcn_s_barrier();                           // barrier X: every consumer is done with the ring
  asm volatile("" ::: "memory");
  if (p.dbg & 1) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  if constexpr (CSUM_LATE) load_csum();
  if (p.ln_colsum) {
    __builtin_amdgcn_s_barrier();                         // barrier Z: the loaders' (mean, rstd) table is in LDS
    asm volatile("" ::: "memory");
  }
  igemm_scratch_write<BM, BN>(p, acc, csum, smem, w4, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                           // barrier Y
  asm volatile("" ::: "memory");
  igemm_epilogue<BM, BN, false, BF>(p, smem, m0, n0, split, w4, 0, lane);
  if (p.gn_part) igemm_gn_stats<BM, BN>(p, smem, m0, n0, w4, 0, lane);
}

// =====================================================================================================================
// PATCH variant for 3x3 / stride 1 / pad 1 convolutions (no up-sampling, channel counts multiples of 64) whose m-tile is a
// whole number of image rows.  k_igemm stages the activation tile of every filter tap separately: nine overlapping
// copies of the same (rows + 2) x (W + 2) pixel patch.  Here the K loop runs channel-group major -- for each 64-channel
// group the 9 taps -- and the patch of a group is brought into LDS ONCE (two patch buffers); the consumers read the
// tap (dy, dx) fragments at pixel offset dy * (W + 2) + dx inside it.  Ring slots then hold the weight tile only (plus
// the activation tile of the K tiles of the extra 1x1 segment, which keep the k_igemm layout).  LDS-DMA pieces per K
// tile and loader wave: 5 + 7/9 instead of 7 at 64x160 (W = 64), 5 + 1 instead of 9 at 128x160.
//   K-tile order t: conv part t < 9 G1: group g = t / 9, tap = t % 9; extra part: tile t - 9 G1 of the 1x1 segment.
//   patch(G) lives in buffer G & 1.  Its pieces ride on the stages of group G-1 from tap 4 >= NS-1 on (the buffer was last read
//   for group G-2, whose last tile is behind every barrier those stages are issued after), two pieces per stage; the first
//   patch of a split (and what the skipped stages would have carried) is issued in the prologue.
//   Stages carry different numbers of loads, so the counted vmcnt waits follow the schedule (W in the loader loop).
// MEASURED (tools/patch_bench.py, MI355X): 1.02-1.13x k_igemm on the long-K 3x3 shapes of the step (0.95-1.0x on the shortest
// ones: the prologue stages a whole patch before the first barrier); one candidate of the per-shape autotuner.
// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [0, 63]: computed jump into a table of (s_waitcnt, s_branch) pairs
// (8 bytes each) -- the counter is an immediate field, and a compare chain costs more than the K tile it guards.
__device__ __forceinline__ void wait_vm_dyn(int n) {
  n = __builtin_amdgcn_readfirstlane(n < 0 ? 0 : n > 63 ? 63 : n);
  asm volatile(
      "s_getpc_b64 s[96:97]\n"                 // address of the next instruction
      "s_lshl_b32 s98, %0, 3\n"                // 4 bytes each from here to the table: 6 instructions = 24 bytes
      "s_add_u32 s96, s96, s98\n"
      "s_addc_u32 s97, s97, 0\n"
      "s_add_u32 s96, s96, 24\n"
      "s_addc_u32 s97, s97, 0\n"
      "s_setpc_b64 s[96:97]\n"
      "s_waitcnt vmcnt(0)\n s_branch 1f\n"
      "s_waitcnt vmcnt(1)\n s_branch 1f\n"
      "s_waitcnt vmcnt(2)\n s_branch 1f\n"
      "s_waitcnt vmcnt(3)\n s_branch 1f\n"
      "s_waitcnt vmcnt(4)\n s_branch 1f\n"
      "s_waitcnt vmcnt(5)\n s_branch 1f\n"
      "s_waitcnt vmcnt(6)\n s_branch 1f\n"
      "s_waitcnt vmcnt(7)\n s_branch 1f\n"
      "s_waitcnt vmcnt(8)\n s_branch 1f\n"
      "s_waitcnt vmcnt(9)\n s_branch 1f\n"
      "s_waitcnt vmcnt(10)\n s_branch 1f\n"
      "s_waitcnt vmcnt(11)\n s_branch 1f\n"
      "s_waitcnt vmcnt(12)\n s_branch 1f\n"
      "s_waitcnt vmcnt(13)\n s_branch 1f\n"
      "s_waitcnt vmcnt(14)\n s_branch 1f\n"
      "s_waitcnt vmcnt(15)\n s_branch 1f\n"
      "s_waitcnt vmcnt(16)\n s_branch 1f\n"
      "s_waitcnt vmcnt(17)\n s_branch 1f\n"
      "s_waitcnt vmcnt(18)\n s_branch 1f\n"
      "s_waitcnt vmcnt(19)\n s_branch 1f\n"
      "s_waitcnt vmcnt(20)\n s_branch 1f\n"
      "s_waitcnt vmcnt(21)\n s_branch 1f\n"
      "s_waitcnt vmcnt(22)\n s_branch 1f\n"
      "s_waitcnt vmcnt(23)\n s_branch 1f\n"
      "s_waitcnt vmcnt(24)\n s_branch 1f\n"
      "s_waitcnt vmcnt(25)\n s_branch 1f\n"
      "s_waitcnt vmcnt(26)\n s_branch 1f\n"
      "s_waitcnt vmcnt(27)\n s_branch 1f\n"
      "s_waitcnt vmcnt(28)\n s_branch 1f\n"
      "s_waitcnt vmcnt(29)\n s_branch 1f\n"
      "s_waitcnt vmcnt(30)\n s_branch 1f\n"
      "s_waitcnt vmcnt(31)\n s_branch 1f\n"
      "s_waitcnt vmcnt(32)\n s_branch 1f\n"
      "s_waitcnt vmcnt(33)\n s_branch 1f\n"
      "s_waitcnt vmcnt(34)\n s_branch 1f\n"
      "s_waitcnt vmcnt(35)\n s_branch 1f\n"
      "s_waitcnt vmcnt(36)\n s_branch 1f\n"
      "s_waitcnt vmcnt(37)\n s_branch 1f\n"
      "s_waitcnt vmcnt(38)\n s_branch 1f\n"
      "s_waitcnt vmcnt(39)\n s_branch 1f\n"
      "s_waitcnt vmcnt(40)\n s_branch 1f\n"
      "s_waitcnt vmcnt(41)\n s_branch 1f\n"
      "s_waitcnt vmcnt(42)\n s_branch 1f\n"
      "s_waitcnt vmcnt(43)\n s_branch 1f\n"
      "s_waitcnt vmcnt(44)\n s_branch 1f\n"
      "s_waitcnt vmcnt(45)\n s_branch 1f\n"
      "s_waitcnt vmcnt(46)\n s_branch 1f\n"
      "s_waitcnt vmcnt(47)\n s_branch 1f\n"
      "s_waitcnt vmcnt(48)\n s_branch 1f\n"
      "s_waitcnt vmcnt(49)\n s_branch 1f\n"
      "s_waitcnt vmcnt(50)\n s_branch 1f\n"
      "s_waitcnt vmcnt(51)\n s_branch 1f\n"
      "s_waitcnt vmcnt(52)\n s_branch 1f\n"
      "s_waitcnt vmcnt(53)\n s_branch 1f\n"
      "s_waitcnt vmcnt(54)\n s_branch 1f\n"
      "s_waitcnt vmcnt(55)\n s_branch 1f\n"
      "s_waitcnt vmcnt(56)\n s_branch 1f\n"
      "s_waitcnt vmcnt(57)\n s_branch 1f\n"
      "s_waitcnt vmcnt(58)\n s_branch 1f\n"
      "s_waitcnt vmcnt(59)\n s_branch 1f\n"
      "s_waitcnt vmcnt(60)\n s_branch 1f\n"
      "s_waitcnt vmcnt(61)\n s_branch 1f\n"
      "s_waitcnt vmcnt(62)\n s_branch 1f\n"
      "s_waitcnt vmcnt(63)\n s_branch 1f\n"
      "1:\n"
      :: "s"(n) : "s96", "s97", "s98", "scc", "memory");
}

#define TF_PATCH_PPW 9     // patch pieces per loader wave at most (33 pieces: BM = 128, W = 64)

template <int BM, int BN, bool GI = false>
__global__ void __launch_bounds__(512, 2) k_igemm_patch(const GemmP p) {
  constexpr int TM = BM / 2, TN = BN / 2, MJ = TM / 16, NI = TN / 16;
  constexpr int BNP = BN / 32;                            // weight pieces per loader wave per K tile
  constexpr int AXP = BM / 32;                            // activation pieces per loader wave of an extra (1x1) K tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wid >= 4;
  const int w4 = wid & 3;
  const int ntiles = p.ntm * p.ntn;
  const int nblk = ntiles * p.splitk;
  int bid = blockIdx.x;
  {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;      // XCD-aware order, as in k_igemm
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = bid / ntiles;
  const int tid_ = bid - split * ntiles;
  int tile_m, tile_n;
  if (p.order == 0) { tile_m = tid_ / p.ntn; tile_n = tid_ - tile_m * p.ntn; }
  else { tile_n = tid_ / p.ntm; tile_m = tid_ - tile_n * p.ntm; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt_begin = split * p.ktiles_per_split;
  const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);
  const int nt = kt_end - kt_begin;

  const int PC = p.W + 2;                                 // patch row pitch (pixels)
  const int PPC = p.pt_ppc;                               // 1-KiB pieces (8 pixels x 64 channels) of one patch
  const int PB = PPC * 1024;
  char* const ring = smem + 2 * PB;
  const int STG = p.pt_stage, NS = p.pt_ns;
  const int G1 = p.C >> 6, T1 = 9 * G1;

  if (loader) {
    // =============================== LOADER WAVES ===============================================
    const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const int sub = lane >> 3;
    const int cs = (lane & 7) ^ ((4 * (w4 & 1) + (sub >> 1)) & 7);      // source chunk of this lane (see k_igemm)
    unsigned gw[BNP];
    int ga[AXP], pp[TF_PATCH_PPW];
#pragma unroll
    for (int i = 0; i < BNP; ++i) {
      int n = n0 + 8 * (w4 + 4 * i) + sub;
      gw[i] = n < p.N ? (unsigned)(n * p.K + cs * 8) * 2u : TF_OOB;
    }
#pragma unroll
    for (int i = 0; i < AXP; ++i) ga[i] = m0 + 8 * (w4 + 4 * i) + sub;   // the 1x1 segment reads the output pixel itself
    {
      const int img = fast_div(m0, p.dv_howo_mul, p.dv_howo_shr);
      const int y0 = (m0 - img * p.HoWo) >> p.pt_log2w;                 // first image row of the tile
#pragma unroll
      for (int i = 0; i < TF_PATCH_PPW; ++i) {
        int q = 8 * (w4 + 4 * i) + sub;
        int pr = q / PC, pc = q - pr * PC;
        int y = y0 + pr - 1, x = pc - 1;
        bool ok = q < p.pt_ppix && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        pp[i] = ok ? (img * p.H + y) * p.W + x : -1;
      }
    }
    const int nv = PPC > w4 ? (PPC - w4 + 3) >> 2 : 0;     // pieces of a patch this wave issues (those with w4 + 4 i < PPC; <= TF_PATCH_PPW)
    constexpr int PQ = 2, TAP0 = 4;                        // pieces of the next patch carried per stage, from tap TAP0 on (NS - 1 <= TAP0)
    auto patch_count = [&](int lo, int hi) { return max(0, min(hi, nv) - min(lo, nv)); };
    auto patch_pieces = [&](int G, int lo, int hi) {      // generic range (prologue only)
      const int c0 = G << 6;
      const bool second = c0 >= p.C1;
      const int ld = second ? p.C2 : p.C1;
      const int cc = (second ? c0 - p.C1 : c0) + cs * 8;
      const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? p.x2 : p.x), 0, second ? p.x2_bytes : p.x_bytes, 0x00020000);
      char* base = smem + (G & 1) * PB;
#pragma unroll
      for (int i = 0; i < TF_PATCH_PPW; ++i) {
        const int pi = w4 + 4 * i;
        if (i >= lo && i < hi && pi < PPC) {
          unsigned off = pp[i] >= 0 ? (unsigned)(pp[i] * ld + cc) * 2u : TF_OOB;
          bload_lds16(rs, off, base + pi * 1024);
        }
      }
    };
    auto patch_pair = [&](int G, int j) {                 // pieces 2 j and 2 j + 1 of patch(G): the stage of tap TAP0 + j carries them
      const int c0 = G << 6;
      const bool second = c0 >= p.C1;
      const int ld = second ? p.C2 : p.C1;
      const int cc = (second ? c0 - p.C1 : c0) + cs * 8;
      const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? p.x2 : p.x), 0, second ? p.x2_bytes : p.x_bytes, 0x00020000);
      char* base = smem + (G & 1) * PB + w4 * 1024;
      auto one = [&](int v, int i) {
        if (w4 + 4 * i < PPC) bload_lds16(rs, v >= 0 ? (unsigned)(v * ld + cc) * 2u : TF_OOB, base + i * 4096);
      };
      switch (j) {                                         // static register indices
        case 0: one(pp[0], 0); one(pp[1], 1); break;
        case 1: one(pp[2], 2); one(pp[3], 3); break;
        case 2: one(pp[4], 4); one(pp[5], 5); break;
        case 3: one(pp[6], 6); one(pp[7], 7); break;
        default: one(pp[8], 8); break;
      }
    };
    // GroupNorm (+ SiLU) of the input: every loader wave normalises the patch pieces IT staged (its own vmcnt covers their
    // landing), in LDS, once per piece instead of once per tap; padding pixels (pp < 0) stay zero
    constexpr bool gi_on = GI;
    float na[8], nb[8];
    int ab_group = -1;
    auto gi_piece = [&](int G, int v, int i) {            // piece w4 + 4 i of patch(G); v = pp[i]
      if (w4 + 4 * i >= PPC) return;
      const unsigned a = lds_off(smem + (G & 1) * PB) + (unsigned)(w4 + 4 * i) * 1024u + lane * 16;
      lds_write16(a, gi_apply(lds_read16(a), na, nb, p.gi_silu, v >= 0));
    };
    auto gi_range = [&](int G, int lo, int hi) {          // pieces lo <= i < hi of patch(G) (static register indices)
      if (ab_group != G) { gi_load_ab(p, smem, (G << 6) + cs * 8, na, nb); ab_group = G; }
#pragma unroll
      for (int i = 0; i < TF_PATCH_PPW; ++i)
        if (i >= lo && i < hi) gi_piece(G, pp[i], i);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // does the stage of conv tile (g, tap) carry pieces of patch(g + 1)?  and how many loads does stage (g, tap) issue
    auto carries = [&](int g, int tap) { return tap >= TAP0 && g + 1 < G1 && 9 * (g + 1) < kt_end; };
    auto count = [&](int g, int tap) {                    // g >= G1: a K tile of the extra 1x1 segment
      if (g >= G1) return BNP + AXP;
      return BNP + (carries(g, tap) ? patch_count((tap - TAP0) * PQ, (tap - TAP0 + 1) * PQ) : 0);
    };
    int sg = G1, stap = 0;                                 // (group, tap) of the next tile to stage
    auto stage = [&](int slot) -> int {                   // stages tile (sg, stap) into ring slot `slot`; returns its load count
      char* base = ring + slot * STG;
      const bool extra = sg >= G1;
      const unsigned koff = extra ? (unsigned)(p.Kc + ((sg - G1) << 6)) : (unsigned)(stap * p.C + (sg << 6));
#pragma unroll
      for (int i = 0; i < BNP; ++i) {
        unsigned off = gw[i] != TF_OOB ? gw[i] + koff * 2u : TF_OOB;
        bload_lds16(rs_w, off, base + (w4 + 4 * i) * 1024);
      }
      const int n = count(sg, stap);
      if (extra) {
        const int c0 = (sg - G1) << 6;
        const bool second = c0 >= p.C3;
        const int ld = second ? p.C4 : p.C3;
        const int cc = (second ? c0 - p.C3 : c0) + cs * 8;
        const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? p.x4 : p.x3), 0, second ? p.x4_bytes : p.x3_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < AXP; ++i) bload_lds16(rs, (unsigned)(ga[i] * ld + cc) * 2u, base + BN * 128 + (w4 + 4 * i) * 1024);
        ++sg;
      } else {
        if (carries(sg, stap)) patch_pair(sg + 1, stap - TAP0);
        if (++stap == 9) { stap = 0; ++sg; }
      }
      return n;
    };
    int pro_g = -1, pro_next = 0;                          // prologue patches to normalise: patch(pro_g) whole, pieces [0, pro_next) of patch(pro_g + 1)
    if (kt_begin < T1) {
      sg = kt_begin / 9; stap = kt_begin - 9 * sg;
      patch_pieces(sg, 0, TF_PATCH_PPW);                   // the first patch of this split, whole
      pro_g = sg;
      if (stap > TAP0 && carries(sg, stap - 1)) patch_pieces(sg + 1, 0, (stap - TAP0) * PQ);   // what the skipped stages carry
      if (stap >= TAP0 && carries(sg, stap)) pro_next = (stap - TAP0 + 1) * PQ;   // ... plus what tile 0's own stage carries: all landed with tile 0
    } else {
      sg = G1 + (kt_begin - T1);
    }
    // W = loads issued after the stage of the tile the consumers need next; (ng, ntap) = that tile's successor
    int ng = sg, ntap = stap, W = 0;
    auto advance = [&]() { if (ng >= G1) ++ng; else if (++ntap == 9) { ntap = 0; ++ng; } };
    advance();                                             // tile 1
    {
      int s_ = 0;
      for (; s_ < NS && s_ < nt; ++s_) { int n = stage(s_); if (s_ > 0) W += n; }
      if (gi_on) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }   // barriers A, B of gi_prologue (consumer waves)
      wait_vm_dyn(W);                                      // tile 0 (and everything issued before it) landed
      if (gi_on && pro_g >= 0) {
        gi_range(pro_g, 0, TF_PATCH_PPW);
        if (pro_next > 0) gi_range(pro_g + 1, 0, pro_next);
      }
    }
    __builtin_amdgcn_s_barrier();                         // barrier P
    asm volatile("" ::: "memory");
    int slot = 0;
    for (int it = 0; it < nt; ++it) {
      if (it + 1 < nt) {
        const int lg_ = ng, lt_ = ntap;                    // tile it+1 = (group, tap)
        W -= count(ng, ntap);                              // tile it+1 must have landed: only newer stages may be in flight
        advance();
        wait_vm_dyn(W);
        // the pieces of patch(group + 1) that rode on tile it+1's stage have landed with it: normalise them now (the consumers
        // read that patch from tile 9 (group + 1) on, behind barrier(9 group + 8) at the earliest)
        if (gi_on && lg_ < G1 && carries(lg_, lt_)) gi_range(lg_ + 1, (lt_ - TAP0) * PQ, (lt_ - TAP0 + 1) * PQ);
      }
      __builtin_amdgcn_s_barrier();                       // barrier(it)
      asm volatile("" ::: "memory");
      if (it + NS < nt) W += stage(slot);
      if (++slot == NS) slot = 0;
    }
    __builtin_amdgcn_s_barrier();                         // barrier X
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // barrier Y: the consumers' tiles are in the LDS scratch
    asm volatile("" ::: "memory");
    igemm_epilogue<BM, BN>(p, smem, m0, n0, split, w4, 1, lane);
    if (p.gn_part) igemm_gn_stats<BM, BN>(p, smem, m0, n0, w4, 1, lane);
    return;
  }

  // ================================= CONSUMER WAVES ===============================================
  if constexpr (GI) gi_prologue(p, smem, m0 / p.HoWo, w4 * 64 + lane);
  const int wave_m = w4 & 1, wave_n = w4 >> 1;
  const int lr = lane & 15, lg = lane >> 4;
  f4 acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  f4 csum[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) csum[i] = (f4){0.f, 0.f, 0.f, 0.f};
  int wa[NI], xe[MJ], q0[MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int row = wave_n * TN + i * 16 + lr;
    wa[i] = row * 128 + ((lg ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    int row = wave_m * TM + j * 16 + lr;
    xe[j] = BN * 128 + row * 128 + ((lg ^ ((row >> 1) & 7)) << 4);
    q0[j] = (row >> p.pt_log2w) * PC + (row & (p.W - 1));              // patch pixel of tap (0, 0) for this output row
  }
  int rt = kt_begin, rg = 0, rdy = 0, rdx = 0;             // next tile to read: index, group, tap
  if (kt_begin < T1) { rg = kt_begin / 9; int tap = kt_begin - 9 * rg; rdy = tap / 3; rdx = tap - 3 * rdy; }
  h8 wfA[2][NI], xfA[2][MJ], wfB[2][NI], xfB[2][MJ];
  auto read_frags = [&](int slot, h8 (&wf)[2][NI], h8 (&xf)[2][MJ]) {
    const char* sb = ring + slot * STG;
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[0][i] = *reinterpret_cast<const h8*>(sb + wa[i]);
    if (rt < T1) {
      const char* pb = smem + (rg & 1) * PB;
      const int dq = rdy * PC + rdx;
      int a[MJ];
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        int q = q0[j] + dq;
        a[j] = (q << 7) + ((lg ^ ((q >> 1) & 7)) << 4);
        xf[0][j] = *reinterpret_cast<const h8*>(pb + a[j]);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[1][i] = *reinterpret_cast<const h8*>(sb + (wa[i] ^ 64));
#pragma unroll
      for (int j = 0; j < MJ; ++j) xf[1][j] = *reinterpret_cast<const h8*>(pb + (a[j] ^ 64));
      if (++rdx == 3) { rdx = 0; if (++rdy == 3) { rdy = 0; ++rg; } }
    } else {
#pragma unroll
      for (int j = 0; j < MJ; ++j) xf[0][j] = *reinterpret_cast<const h8*>(sb + xe[j]);
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[1][i] = *reinterpret_cast<const h8*>(sb + (wa[i] ^ 64));
#pragma unroll
      for (int j = 0; j < MJ; ++j) xf[1][j] = *reinterpret_cast<const h8*>(sb + (xe[j] ^ 64));
    }
    ++rt;
  };
  auto mma = [&](h8 (&wf)[2][NI], h8 (&xf)[2][MJ]) {
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[k2][i], xf[k2][j], acc[i][j], 0, 0, 0);
  };
  __builtin_amdgcn_s_barrier();                           // barrier P: tile 0 (and its patch) landed
  asm volatile("" ::: "memory");
  int rslot = 0;
  if (nt > 0) { read_frags(0, wfA, xfA); rslot = 1; }
  for (int it = 0; it < nt; it += 2) {
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();                         // barrier(it): tile it+1 landed
    asm volatile("" ::: "memory");
    if (it + 1 < nt) { read_frags(rslot, wfB, xfB); if (++rslot == NS) rslot = 0; }
    __builtin_amdgcn_sched_barrier(0);
    mma(wfA, xfA);
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 >= nt) break;
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();                         // barrier(it+1)
    asm volatile("" ::: "memory");
    if (it + 2 < nt) { read_frags(rslot, wfA, xfA); if (++rslot == NS) rslot = 0; }
    __builtin_amdgcn_sched_barrier(0);
    mma(wfB, xfB);
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_barrier();                           // barrier X: every consumer is done with the ring and the patches
  asm volatile("" ::: "memory");
  igemm_scratch_write<BM, BN>(p, acc, csum, smem, w4, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                           // barrier Y
  asm volatile("" ::: "memory");
  igemm_epilogue<BM, BN>(p, smem, m0, n0, split, w4, 0, lane);
  if (p.gn_part) igemm_gn_stats<BM, BN>(p, smem, m0, n0, w4, 0, lane);
}

// =====================================================================================================================
// fp8 (OCP e4m3) variant, BASELINE config 5: same 4 loader + 4 consumer waves, LDS-DMA ring, raw barriers and epilogue as the deep
// k_igemm, with one byte per operand element:
//   * a K tile is still 64 elements = 64-BYTE rows, so a 1-KiB LDS-DMA piece covers 16 rows (lane -> row pair lane >> 3, half
//     (lane >> 2) & 1, 16-byte chunk lane & 3) and a stage is (BM + BN) * 64 bytes: half the ingest per FLOP of the fp16 kernel, which
//     is what bounds that one -- and room for a 256-row tile at 8 waves x 256 VGPRs;
//   * swizzle for the 64-byte pitch: chunk ^ (-(row >> 2) & 3), conflict-free for the four 16-lane groups of a ds_read_b128
//     (MI355X_MICROARCH.md, LDS table) -- applied on the source chunk by the loaders and again on the read;
//   * ONE ds_read_b128 per 16-row fragment and K tile: lane group lg takes the 16 elements k = 16 lg .. 16 lg + 15, their low 8 bytes feed
//     the first v_mfma_f32_16x16x32_fp8_fp8, the high 8 the second.  Both operands are cut the same way, so every k meets its partner
//     (the MFMA sums over k in whatever order the lanes hold it);
//   * the per-output-channel weight scale multiplies the fp32 accumulators before the shared epilogue (bias, time embedding, residual,
//     GEGLU, GroupNorm statistics, split-K partials, optional e4m3 output).
// Channel counts are multiples of 64 (taps and concat sources advance as wave-uniform scalars); no extra 1x1 segment, no LayerNorm fold.
constexpr int ring_slots8(int bm, int bn) { int s = 163840 / ((bm + bn) * 64); return s > 8 ? 8 : s; }

template <int BM, int BN>
__global__ void __launch_bounds__(512, 2) k_igemm8(const GemmP p) {
  constexpr int TM = BM / 2, TN = BN / 2, MJ = TM / 16, NI = TN / 16;
  constexpr int NG = (BM + BN) / 16;                      // 16-row staging pieces: activation rows first, then weight rows
  constexpr int LPS = NG / 4;
  constexpr int STAGE = (BM + BN) * 64;
  constexpr int NS = ring_slots8(BM, BN);
  static_assert(NG % 4 == 0 && BM % 32 == 0 && BN % 32 == 0, "tile shape");
  static_assert((NS - 2) * LPS <= 63, "vmcnt immediate is 6 bits");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned char* X = reinterpret_cast<const unsigned char*>(p.x);
  const unsigned char* X2 = reinterpret_cast<const unsigned char*>(p.x2);
  const unsigned char* Wt = reinterpret_cast<const unsigned char*>(p.w);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wid >= 4;
  const int w4 = wid & 3;
  const int ntiles = p.ntm * p.ntn;
  const int nblk = ntiles * p.splitk;
  int bid = blockIdx.x;
  {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;      // XCD-aware order, as in k_igemm
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = bid / ntiles;
  const int tid_ = bid - split * ntiles;
  int tile_m, tile_n;
  if (p.order == 0) { tile_m = tid_ / p.ntn; tile_n = tid_ - tile_m * p.ntn; }
  else { tile_n = tid_ / p.ntm; tile_m = tid_ - tile_n * p.ntm; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt_begin = split * p.ktiles_per_split;
  const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);
  const int nt = kt_end - kt_begin;

  if (loader) {
    const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, p.w_bytes, 0x00020000);
    // lane -> row 2 (lane >> 3) + ((lane >> 2) & 1) of its 16-row piece, LDS chunk lane & 3; source chunk = LDS chunk ^ swizzle(row),
    // swizzle(row) = -(row >> 2) & 3 = -(lane >> 4) & 3 for every piece (pieces start at multiples of 16 rows)
    const int rin = 2 * (lane >> 3) + ((lane >> 2) & 1);
    const int kc = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
    int g_a[LPS], g_b[LPS], g_c[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
      const int row = 16 * (w4 + 4 * i) + rin;
      g_a[i] = -(1 << 28); g_b[i] = 0; g_c[i] = (int)TF_OOB;
      if (row < BM) {
        int m = m0 + row;
        if (m < p.M) {
          int img = fast_div(m, p.dv_howo_mul, p.dv_howo_shr), rem = m - img * p.HoWo;
          int ho = fast_div(rem, p.dv_wo_mul, p.dv_wo_shr), wo = rem - ho * p.Wo;
          g_a[i] = ho * p.stride - p.pad;
          g_b[i] = wo * p.stride - p.pad;
          g_c[i] = img * p.H * p.W;
        }
      } else {
        int n = n0 + row - BM;
        if (n < p.N) g_c[i] = (int)((unsigned)n * (unsigned)p.K + (unsigned)kc * 16u);
      }
    }
    const int Hl = p.H << p.ups, Wl = p.W << p.ups;
    int st_r, st_s, st_c;
    {
      int kg0 = kt_begin * 64, tap = kg0 / p.C;
      st_c = kg0 - tap * p.C;
      st_r = tap / p.S;
      st_s = tap - st_r * p.S;
    }
    auto stage = [&](int buf, int kt) {
      char* base = smem + buf * STAGE;
      const int r = st_r, s_ = st_s;
      const bool second = st_c >= p.C1;
      const int ld = second ? p.C2 : p.C1;
      const int cc = (second ? st_c - p.C1 : st_c) + kc * 16;
      st_c += 64;
      if (st_c >= p.C) { st_c = 0; if (++st_s == p.S) { st_s = 0; ++st_r; } }
      const rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? X2 : X), 0, second ? p.x2_bytes : p.x_bytes, 0x00020000);
      const unsigned kb = (unsigned)kt * 64u;
#pragma unroll
      for (int i = 0; i < LPS; ++i) {
        const int g = w4 + 4 * i;
        char* dst = base + g * 1024;
        if (g * 16 < BM) {
          int hi = g_a[i] + r, wi = g_b[i] + s_;
          bool ok = (unsigned)hi < (unsigned)Hl && (unsigned)wi < (unsigned)Wl;
          int pix = g_c[i] + (hi >> p.ups) * p.W + (wi >> p.ups);
          bload_lds16(rs_a, ok ? (unsigned)(pix * ld + cc) : TF_OOB, dst);
        } else {
          unsigned wo = (unsigned)g_c[i];
          bload_lds16(rs_w, wo != TF_OOB ? wo + kb : TF_OOB, dst);
        }
      }
    };
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_)
      if (s_ < nt) stage(s_, kt_begin + s_);
    wait_stages<LPS, NS - 1>(nt - 1);
    __builtin_amdgcn_s_barrier();                         // barrier P
    asm volatile("" ::: "memory");
    for (int it = 0; it < nt; ++it) {
      if (it + 1 < nt) wait_stages<LPS, NS - 2>(nt - 2 - it);
      __builtin_amdgcn_s_barrier();                       // barrier(it)
      asm volatile("" ::: "memory");
      if (it + NS < nt) stage(it % NS, kt_begin + it + NS);
    }
    __builtin_amdgcn_s_barrier();                         // barrier X
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // barrier Y
    asm volatile("" ::: "memory");
    if (p.out8) igemm_epilogue<BM, BN, true>(p, smem, m0, n0, split, w4, 1, lane);
    else igemm_epilogue<BM, BN, false>(p, smem, m0, n0, split, w4, 1, lane);
    if (p.gn_part) igemm_gn_stats<BM, BN>(p, smem, m0, n0, w4, 1, lane);
    return;
  }

  // ================================= CONSUMER WAVES ===============================================
  const int wave_m = w4 & 1, wave_n = w4 >> 1;
  const int lr = lane & 15, lg = lane >> 4;
  f4 acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  f4 wsc[NI];                                             // per-output-channel weight scales of this lane's 4 consecutive channels
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int n = n0 + wave_n * TN + i * 16 + lg * 4;
    wsc[i] = (f4){1.f, 1.f, 1.f, 1.f};
    if (p.wscale) for (int e = 0; e < 4; ++e) if (n + e < p.N) wsc[i][e] = p.wscale[n + e];
  }
  int wa[NI], xa[MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int row = wave_n * TN + i * 16 + lr;
    wa[i] = BM * 64 + row * 64 + ((lg ^ ((0 - (row >> 2)) & 3)) << 4);
  }
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    int row = wave_m * TM + j * 16 + lr;
    xa[j] = row * 64 + ((lg ^ ((0 - (row >> 2)) & 3)) << 4);
  }
  typedef long l2v __attribute__((ext_vector_type(2)));
  l2v wfA[NI], xfA[MJ], wfB[NI], xfB[MJ];
  auto read_frags = [&](int slot, l2v (&wf)[NI], l2v (&xf)[MJ]) {
    const char* sb = smem + slot * STAGE;
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const l2v*>(sb + wa[i]);
#pragma unroll
    for (int j = 0; j < MJ; ++j) xf[j] = *reinterpret_cast<const l2v*>(sb + xa[j]);
  };
  auto mma = [&](l2v (&wf)[NI], l2v (&xf)[MJ]) {
    // both halves of a fragment pair back to back on the same accumulator (a 16x16x32 chain issues at the full rate on one
    // accumulator): with the k halves as the outer loop the compiler ping-pongs the whole accumulator set between two register banks
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf[i][0], xf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf[i][1], xf[j][1], acc[i][j], 0, 0, 0);
      }
  };
  __builtin_amdgcn_s_barrier();                           // barrier P: tile 0 landed
  asm volatile("" ::: "memory");
  if (nt > 0) read_frags(0, wfA, xfA);
  for (int it = 0; it < nt; it += 2) {
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();                         // barrier(it): tile it+1 landed
    asm volatile("" ::: "memory");
    if (it + 1 < nt) read_frags((it + 1) % NS, wfB, xfB);
    __builtin_amdgcn_sched_barrier(0);
    mma(wfA, xfA);
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 >= nt) break;
    wait_lds_reads();
    __builtin_amdgcn_s_barrier();                         // barrier(it+1)
    asm volatile("" ::: "memory");
    if (it + 2 < nt) read_frags((it + 2) % NS, wfA, xfA);
    __builtin_amdgcn_sched_barrier(0);
    mma(wfB, xfB);
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_barrier();                           // barrier X: every consumer is done with the ring
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] *= wsc[i];
  f4 csum[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) csum[i] = (f4){0.f, 0.f, 0.f, 0.f};
  igemm_scratch_write<BM, BN>(p, acc, csum, smem, w4, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                           // barrier Y
  asm volatile("" ::: "memory");
  if (p.out8) igemm_epilogue<BM, BN, true>(p, smem, m0, n0, split, w4, 0, lane);
  else igemm_epilogue<BM, BN, false>(p, smem, m0, n0, split, w4, 0, lane);
  if (p.gn_part) igemm_gn_stats<BM, BN>(p, smem, m0, n0, w4, 0, lane);
}

// =====================================================================================================================
// PING-PONG variant for problems that fill the chip with 256-row tiles (more images per GPU, 96 x 96 latents: BASELINE config 5):
// ALL EIGHT waves load and compute.  The loader / consumer split of k_igemm leaves the matrix pipe to four waves and tops out at
// ~1.1 PFLOP/s; here the block tile is 256 x BN (BN = 128 / 160 / 256), the waves form a 4 (pixels) x 2 (channels) grid of 64 x BN/2
// accumulator tiles (64 ... 128 VGPRs), and the two halves of the workgroup -- waves 0-3 and 4-7, one of each per SIMD -- run the same
// program ONE BARRIER APART: while one half issues its 16-40 MFMAs of a 32-deep k-step, the other half reads the fragments of its
// next k-step from LDS and issues its share of the LDS-DMA for a later K tile; at the next s_barrier they swap.  So the matrix pipe
// of every SIMD always has a wave feeding it and the DMA issue cost (60-180 cycles per 1-KiB piece) hides under the partner's MFMAs.
//   ring: NS = 3 slots (BN <= 160) or 2 (BN = 256) of (256 + BN) x 128 B, tile t in slot t % NS; during tile t every wave issues its
//     pieces of tile t + NS - 1 (activation pieces with k-step 0, weight pieces with k-step 1) into the slot of tile t - 1.
//   RAW: a wave's counted s_waitcnt vmcnt for its pieces of tile t+1 sits in the last half-phase before the barrier that precedes the
//     FIRST half's k-step 0 of tile t+1 (first half: behind its MFMAs of (t, k1); second half: at the end of its load segment of
//     (t, k1)); every read of tile t+1 comes behind that barrier.
//   WAR: every load segment ends with s_waitcnt lgkmcnt(0) IN FRONT OF its barrier, so behind a barrier all reads issued before it
//     are done; the second half's last reads of tile t-1 end before the barrier in front of the first half's (t, k0) segment, which is
//     the earliest place a DMA into that slot is issued.
//   LDS-DMA is issued from inline asm (M0 + buffer_load ... lds): the compiler does not see an LDS write and therefore puts no
//     s_waitcnt vmcnt(0) in front of the fragment reads; all vmcnt bookkeeping is the counted waits above.
// Epilogue: the accumulators go through the 2 x 2-wave-tile scratch of k_igemm in two passes of BM / 2 rows (igemm_epilogue<BM / 2, BN>), so
// bias / time embedding / residual / GEGLU / split-K partials / GroupNorm statistics are the shared code, chunked as a 128-row tile.
// Channel counts on the 64 grid (taps and concat sources advance as wave-uniform scalars), no LayerNorm fold, no input GroupNorm.
typedef int i4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i4v raw_rsrc(const void* base, unsigned bytes) {
  unsigned long long a = (unsigned long long)base;
  i4v r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void dma16(i4v rsrc, unsigned voffset_bytes, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane((int)lds_base)), "v"(voffset_bytes), "s"(rsrc) : "memory");   // M0 has no other user in this kernel
}

// NP = half-phases per K tile and wave group: 2 = one per 32-deep k-step (fragments of one k-step in registers), 1 = the whole K tile per
//   phase (both k-steps' fragments in registers, half the barriers; needs the 3-slot ring: with two slots the second half would issue a
//   tile's pieces and wait for them in the same segment).
// FASTA = the lean activation addressing for stride-1 convolutions without up-sampling (and linears): per piece a pixel index and a
//   bit mask of the taps that fall inside the image, so a tile's source offset is one mad + one mask test instead of the bounds
//   arithmetic of the general gather (the load segments, not the MFMAs, set this kernel's pace: every VALU / SALU instruction in them counts).
// DBG: the ablation build (p.dbg: 1 no epilogue, 2 no MFMA, 4 no staging in the loop, 8 no fragment reads)
// F8 = OCP e4m3 operands (BASELINE config 5) on the block-scaled MFMA v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales: 128-deep K per
//   instruction at twice the fp16 rate.  The LDS image is the fp16 kernel's byte for byte -- a K tile is 128 BYTES of every row, i.e. 128
//   e4m3 elements -- and so are the fragment reads: lane group lg takes chunk lg and chunk lg + 4 of its row (k = 16 lg .. 16 lg + 15 and
//   64 + 16 lg ..), both operands cut the same way, so every k meets its partner whatever order the instruction walks them in; the two
//   16-byte reads are the low and the high half of ONE MFMA's 32-byte operand.  A K tile is two 64-channel HALVES that may lie in
//   different taps / source tensors (320 channels = 2.5 tiles): H2 = true issues every activation piece as two half-masked loads with
//   their own descriptor and offsets (same count every tile: the vmcnt bookkeeping stays static); H2 = false (every channel count a
//   multiple of 128) one load.  Per-output-channel weight scales multiply the accumulators in front of the shared epilogue.
// BM = 256 or 192 rows: 192 (wave tiles of 48 rows) exists for the tile COUNT -- 96 x 96 latents give M = 9216 * images rows, and
//   e.g. 73728 x 320 is 576 tiles of 256 x 160 = 2.25 rounds on 256 CUs but 768 tiles of 192 x 160 = 3 rounds exactly.
// LNF = the LayerNorm fold (tf_linear_ln_f16: Linear(LN(x)) = rstd[m] (x . w'^T - mean[m] colsum[n]) + bias'[n]): the row statistics come from
//   the activation FRAGMENTS the wave multiplies anyway -- lane (lr, lg) holds the 8 k-values k = 8 lg .. of row lr of every fragment, so
//   8 v_dot2_f32_f16 per fragment (in the MFMA block's spare issue slots) keep (sum, sum of squares) of that row's share, two lane
//   shuffles at the end complete the row -- and they end up in exactly the lanes whose accumulators belong to that row.
template <int BN, int NP, bool FASTA, bool DBG = false, bool F8 = false, bool H2 = false, int BM = 256, bool LNF = false>
__global__ void __launch_bounds__(512, 2) k_igemm_pp(const GemmP p) {
  static_assert(!LNF || !F8, "the LayerNorm fold is an fp16 path");
  constexpr int TN = BN / 2, MJ = BM / 64, NI = TN / 16;
  constexpr int APW = BM / 64;                            // activation pieces (8 rows x 128 B) per wave and stage: BM / 8 pieces in front of the weight pieces
  constexpr int ES = F8 ? 1 : 2;                          // bytes per element
  constexpr int APL = (F8 && H2) ? 2 * APW : APW;         // activation loads per wave and K tile
  static_assert(BM == 256 || BM == 192, "block rows");
  static_assert(!F8 || NP == 1, "the 128-deep MFMA takes both 64-byte halves of a row at once");
  static_assert(F8 || !H2, "half-masked activation loads are the fp8 kernel's");
  constexpr int NWG = BN / 8;                             // weight pieces (8 rows x 128 B) of a stage
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NS = (163840 / STAGE) >= 3 ? 3 : 2;
  constexpr int D = NS - 1;                               // K tiles in flight ahead of the one being multiplied
  constexpr int WPW = (NWG + 7) / 8;                      // weight pieces per wave (the last one only on waves < NWG % 8 where that is not 0)
  constexpr int WREM = NWG % 8;
  constexpr int KF = NP == 1 ? 2 : 1;                     // k-steps whose fragments are held at once
  static_assert(TN % 16 == 0 && BN % 32 == 0, "tile shape");
  static_assert(NP == 2 || (NP == 1 && NS >= 3), "one phase per K tile needs the 3-slot ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;                               // 0: first half (runs one barrier ahead), 1: second half
  const int wm = wid & 3, wn = wid >> 2;                  // wave tile: pixels 64 wm .., channels TN wn ..
  const int ntiles = p.ntm * p.ntn;
  const int nblk = ntiles * p.splitk;
  int bid = blockIdx.x;
  {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;      // XCD-aware order, as in k_igemm
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = bid / ntiles;
  const int tid_ = bid - split * ntiles;
  int tile_m, tile_n;
  if (p.order == 0) { tile_m = tid_ / p.ntn; tile_n = tid_ - tile_m * p.ntn; }
  else { tile_n = tid_ / p.ntm; tile_m = tid_ - tile_n * p.ntm; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt_begin = split * p.ktiles_per_split;
  const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);
  const int nt = kt_end - kt_begin;

  // ---- staging state: wave w owns activation pieces w + 8 i (i < 4) and weight pieces w + 8 i (i < WPW, below NWG)
  const i4v rs_w = raw_rsrc(p.w, p.w_bytes);
  const int sub = lane >> 3;
  const int cs = (lane & 7) ^ ((4 * (wid & 1) + (sub >> 1)) & 7);       // source chunk of this lane: XOR swizzle on the SOURCE side (see k_igemm)
  // general gather: (hi0, wi0, first pixel of the image) per piece; FASTA: (pixel index of the output position, tap-validity mask, -)
  int g_a[APW], g_b[APW], g_c[APW];
  unsigned gw[WPW];
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    const int m = m0 + 8 * (wid + 8 * i) + sub;
    g_a[i] = FASTA ? 0 : -(1 << 28); g_b[i] = 0; g_c[i] = 0;
    if (m < p.M) {
      int img = fast_div(m, p.dv_howo_mul, p.dv_howo_shr), rem = m - img * p.HoWo;
      int ho = fast_div(rem, p.dv_wo_mul, p.dv_wo_shr), wo = rem - ho * p.Wo;
      if constexpr (FASTA) {
        // stride 1, no up-sampling: input pixel of tap (r, s) = output position + (r - pad) W + (s - pad); bit r S + s of the mask tells
        // whether it lies inside the image, bit 31 marks a live row (the extra 1x1 segment and 1x1 convolutions read the position itself)
        g_a[i] = img * p.H * p.W + ho * p.W + wo;
        unsigned mask = 0x80000000u;
        for (int r = 0; r < p.S; ++r)
          for (int s_ = 0; s_ < p.S; ++s_)
            if ((unsigned)(ho - p.pad + r) < (unsigned)p.H && (unsigned)(wo - p.pad + s_) < (unsigned)p.W) mask |= 1u << (r * p.S + s_);
        g_b[i] = (int)mask;
      } else {
        g_a[i] = ho * p.stride - p.pad;
        g_b[i] = wo * p.stride - p.pad;
        g_c[i] = img * p.H * p.W;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int g = wid + 8 * i, n = n0 + 8 * g + sub;
    gw[i] = (g < NWG && n < p.N) ? (unsigned)(n * p.K) * ES + cs * 16u : TF_OOB;
  }
  const int klim = p.K * ES - cs * 16;                     // this lane's 16 bytes of K tile kt lie inside the row iff kt * 128 < klim (fp8: K need not be a multiple of 128)
  const int Hl = p.H << p.ups, Wl = p.W << p.ups;
  const unsigned lds0 = lds_off(smem);
  int st_r, st_s, st_c;                                    // wave-uniform (tap, channel) of the next 64-channel slab whose activation pieces are staged
  int hrem = (p.K >> 6) - kt_begin * (F8 ? 2 : 1);         // 64-channel slabs from this split's first one to the end of K
  {
    int kg0 = kt_begin * (F8 ? 128 : 64);
    if (kg0 < p.Kc) {
      int tap = kg0 / p.C;
      st_c = kg0 - tap * p.C;
      st_r = tap / p.S;
      st_s = tap - st_r * p.S;
    } else { st_r = -1; st_s = 0; st_c = kg0 - p.Kc; }
  }
  // The scalars of a tile's activation pieces are prepared one half-phase early, in the MFMA shadow: kernel-argument loads and the tap
  // bookkeeping would otherwise sit between the fragment reads and the DMA issue of a load segment (and their s_waitcnt lgkmcnt(0)
  // would wait for the LDS reads as well).  General gather: (r, s, first channel, row pitch); FASTA: (tap bit, byte offset of the tap
  // + first channel, row pitch in bytes).
  int a_r = 0, a_s = 0, a_c0 = 0, a_ld = 0;
  int a_lo = 0, a_hi = 0, a_nb = 0;                        // descriptor words of the tile's source tensor (base low / high, bytes)
  const int ups = p.ups, Wd = p.W, pad_ = p.pad, S_ = p.S;
  // every kernel argument the per-tile bookkeeping needs, read ONCE: an s_load inside the K loop costs its full latency in a wave that
  // has nothing else to issue
  const int C1_ = p.C1, C2_ = p.C2, C3_ = p.C3, C4_ = p.C4, Cc_ = p.C, Kc_ = p.Kc;
  const unsigned long long px1 = (unsigned long long)p.x, px2 = (unsigned long long)(p.x2 ? p.x2 : p.x);
  const unsigned long long px3 = (unsigned long long)(p.x3 ? p.x3 : p.x), px4 = (unsigned long long)(p.x4 ? p.x4 : p.x);
  const int nb1 = (int)p.x_bytes, nb2 = (int)p.x2_bytes, nb3 = (int)p.x3_bytes, nb4 = (int)p.x4_bytes;
  // Tiles come in runs: the 64-channel tiles of one (tap, source tensor) differ only in the first channel.  run_left = tiles of the
  // current run still to be prepared after the last one; inside a run the bookkeeping is one add (a few SALU instructions instead of
  // ~60: they sit in the MFMA half of a phase and lengthen it one for one).  (st_r, st_s, st_c) is normalised lazily: at the head of a run.
  int run_left = 0;
  auto prep_act = [&]() {
    if (--hrem < 0) {                                      // past the end of K (the second half of an fp8 kernel's last tile): nothing valid
      a_r = FASTA ? 0 : -(1 << 28);
      run_left = 0;
    } else if (run_left > 0) {
      --run_left;
      st_c += 64;
      a_c0 += FASTA ? 64 * ES : 64;
    } else {
      bool second;
      int r, s_, c0, ld, seg_end;
      unsigned long long px;
      if (st_r >= 0 && st_c >= Cc_) { st_c = 0; if (++st_s == S_) { st_s = 0; if ((++st_r) * S_ * Cc_ >= Kc_) st_r = -1; } }
      const bool extra = st_r < 0;
      if (!extra) {
        r = st_r; s_ = st_s;
        second = st_c >= C1_;
        ld = second ? C2_ : C1_;
        c0 = second ? st_c - C1_ : st_c;
        seg_end = second ? Cc_ : C1_;
        px = second ? px2 : px1; a_nb = second ? nb2 : nb1;
      } else {
        r = pad_; s_ = pad_;                               // the extra 1x1 segment reads the output pixel itself
        second = st_c >= C3_;
        ld = second ? C4_ : C3_;
        c0 = second ? st_c - C3_ : st_c;
        seg_end = second ? C3_ + C4_ : C3_;
        px = second ? px4 : px3; a_nb = second ? nb4 : nb3;
      }
      a_lo = (int)(unsigned)px; a_hi = (int)((unsigned)(px >> 32) & 0xffffu);
      run_left = ((seg_end - st_c) >> 6) - 1;
      st_c += 64;
      if constexpr (FASTA) {
        a_r = extra ? (int)0x80000000u : (1 << (r * S_ + s_));
        a_c0 = (((r - pad_) * Wd + (s_ - pad_)) * ld + c0) * ES;
        a_ld = ld * ES;
      } else { a_r = r; a_s = s_; a_c0 = c0; a_ld = ld; }
    }
  };
  // fp8: a K tile = two slabs; prep2() prepares both and keeps the first one's scalars aside
  int b_r = 0, b_s = 0, b_c0 = 0, b_ld = 0, b_lo = 0, b_hi = 0, b_nb = 0;
  auto prep_tile = [&]() {
    prep_act();
    if constexpr (F8) {
      b_r = a_r; b_s = a_s; b_c0 = a_c0; b_ld = a_ld; b_lo = a_lo; b_hi = a_hi; b_nb = a_nb;     // slab 0 -> b_*, slab 1 -> a_*
      prep_act();
    }
  };
  auto stage_act = [&](int slot) {
    const unsigned base = lds0 + (unsigned)slot * STAGE + (unsigned)wid * 1024u;
    // (the scalars are wave-uniform by construction; the readfirstlanes are no-ops that keep them in SGPRs whatever the compiler's
    // divergence analysis makes of the bookkeeping's control flow)
    auto one = [&](int lo, int hi, int nb, int r_, int s_, int c0_, int ld_, int cq, int hsel) {
      // cq: this lane's 16-byte chunk inside the slab; hsel < 0: every lane issues, else only the lanes of half hsel
      i4v rs;
      rs[0] = __builtin_amdgcn_readfirstlane(lo); rs[1] = __builtin_amdgcn_readfirstlane(hi);
      rs[2] = __builtin_amdgcn_readfirstlane(nb); rs[3] = 0x00020000;
      const int s_r = __builtin_amdgcn_readfirstlane(r_), s_c0 = __builtin_amdgcn_readfirstlane(c0_), s_ld = __builtin_amdgcn_readfirstlane(ld_);
      const bool mine = hsel < 0 || (cs >> 2) == hsel;
      if constexpr (FASTA) {
        const int vc = s_c0 + cq * 16;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
          unsigned off = __umul24((unsigned)g_a[i], (unsigned)s_ld) + (unsigned)vc;
          if (mine) dma16(rs, (g_b[i] & s_r) ? off : TF_OOB, base + (unsigned)i * 8192u);
        }
      } else {
        const int s_s = __builtin_amdgcn_readfirstlane(s_);
        const int cc = s_c0 + cq * (16 / ES);
#pragma unroll
        for (int i = 0; i < APW; ++i) {
          int hi_ = g_a[i] + s_r, wi = g_b[i] + s_s;
          bool ok = (unsigned)hi_ < (unsigned)Hl && (unsigned)wi < (unsigned)Wl;
          int pix = g_c[i] + (hi_ >> ups) * Wd + (wi >> ups);
          if (mine) dma16(rs, ok ? (unsigned)(pix * s_ld + cc) * ES : TF_OOB, base + (unsigned)i * 8192u);
        }
      }
    };
    if constexpr (!F8) one(a_lo, a_hi, a_nb, a_r, a_s, a_c0, a_ld, cs, -1);
    else if constexpr (H2) {
      one(b_lo, b_hi, b_nb, b_r, b_s, b_c0, b_ld, cs & 3, 0);
      one(a_lo, a_hi, a_nb, a_r, a_s, a_c0, a_ld, cs & 3, 1);
    } else one(b_lo, b_hi, b_nb, b_r, b_s, b_c0, b_ld, cs, -1);       // channel counts on the 128 grid: the two slabs of a tile are 128 contiguous bytes
  };
  auto stage_w = [&](int slot, int kt) {
    const unsigned base = lds0 + (unsigned)slot * STAGE + (unsigned)(BM / 8 + wid) * 1024u;
    const unsigned kb = (unsigned)kt * 128u;
#pragma unroll
    for (int i = 0; i < WPW; ++i)
      if (WREM == 0 || i < WPW - 1 || wid < WREM) dma16(rs_w, (gw[i] != TF_OOB && (!F8 || (int)kb < klim)) ? gw[i] + kb : TF_OOB, base + (unsigned)i * 8192u);
  };
  // "this wave's pieces of every tile but the newest one (NEWEST) / of every tile (!NEWEST) have landed"
  auto wait_landed = [&](auto newest) {
    if constexpr (decltype(newest)::value && D >= 2) {
      if (WREM == 0 || wid < WREM) wait_vm<APL + WPW>(); else wait_vm<APL + WPW - 1>();
    } else wait_vm<0>();
  };

  // ---- fragment addresses inside a stage: the swizzle term depends on lane only (tile offsets are multiples of 16 rows)
  const int lr = lane & 15, lg = lane >> 4;
  const int fo = lr * 128 + ((lg ^ ((lr >> 1) & 7)) << 4);
  const int xo = wm * (BM / 4) * 128 + fo;                 // + j * 2048
  const int wo_ = (BM + wn * TN) * 128 + fo;               // + i * 2048
  f4 acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  h8 wf[KF][NI], xf[KF][MJ];
  if constexpr (DBG) {
#pragma unroll
    for (int f = 0; f < KF; ++f) {
#pragma unroll
      for (int j = 0; j < MJ; ++j) xf[f][j] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[f][i] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  auto read_k = [&](const char* sb, int k2, int f) {      // fragments of k-step k2 into register set f
    if constexpr (DBG) { if (p.dbg & 8) return; }
#pragma unroll
    for (int j = 0; j < MJ; ++j) xf[f][j] = *reinterpret_cast<const h8*>(sb + ((xo + j * 2048) ^ (k2 * 64)));
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[f][i] = *reinterpret_cast<const h8*>(sb + ((wo_ + i * 2048) ^ (k2 * 64)));
  };
  float ls[MJ], lq[MJ];                                    // LNF: this lane's share of (sum x, sum x^2) of row lr of every pixel tile
#pragma unroll
  for (int j = 0; j < MJ; ++j) { ls[j] = 0.f; lq[j] = 0.f; }
  auto mma = [&]() {                                       // the MFMAs of every k-step held in registers
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LNF) {
      typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
      const hh2 one2 = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
      for (int f = 0; f < KF; ++f)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hh2 v = {xf[f][j][2 * e], xf[f][j][2 * e + 1]};
            ls[j] = __builtin_amdgcn_fdot2(v, one2, ls[j], false);
            lq[j] = __builtin_amdgcn_fdot2(v, v, lq[j], false);
          }
    }
    if constexpr (DBG) {
      if (p.dbg & 2) {
#pragma unroll
        for (int f = 0; f < KF; ++f) {
#pragma unroll
          for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(wf[f][i]));
#pragma unroll
          for (int j = 0; j < MJ; ++j) asm volatile("" ::"v"(xf[f][j]));
        }
        return;
      }
    }
    __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
      typedef int v8i __attribute__((ext_vector_type(8)));
      typedef int v4i __attribute__((ext_vector_type(4)));
      v8i xv[MJ];
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        v4i lo = __builtin_bit_cast(v4i, xf[0][j]), hi = __builtin_bit_cast(v4i, xf[KF - 1][j]);
        xv[j] = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        v4i lo = __builtin_bit_cast(v4i, wf[0][i]), hi = __builtin_bit_cast(v4i, wf[KF - 1][i]);
        const v8i wv = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int j = 0; j < MJ; ++j)      // e4m3 x e4m3, block scales 2^0 (E8M0 0x7F) on both sides
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, xv[j], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
    } else {
#pragma unroll
      for (int f = 0; f < KF; ++f)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[f][i], xf[f][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // bias and time-embedding values of this tile's columns, fetched now (latency under the K loop) and handed to the epilogue through an
  // LDS table: thread t < BN holds column n0 + t; a tile spans at most two images (the host admits the kernel only where HoWo >= BM)
  const int lb_img0 = m0 / p.HoWo;
  float lb_b = 0.f, lb_c0 = 0.f, lb_c1 = 0.f;
  if (tid < BN && n0 + tid < p.N && p.splitk <= 1) {
    if (p.bias) lb_b = (float)p.bias[n0 + tid];
    if (p.bias_nc) {
      lb_c0 = (float)p.bias_nc[(long long)lb_img0 * p.bias_nc_stride + n0 + tid];
      if ((lb_img0 + 1) * p.HoWo < p.M) lb_c1 = (float)p.bias_nc[(long long)(lb_img0 + 1) * p.bias_nc_stride + n0 + tid];
    }
  }
  // ---- prologue: the first D tiles, whole
#pragma unroll
  for (int s_ = 0; s_ < D; ++s_)
    if (s_ < nt) { prep_tile(); stage_act(s_); stage_w(s_, kt_begin + s_); }
  if (D < nt) prep_tile();                                 // the scalars of tile D: its pieces ride on tile 0
  if (D >= 2 && nt >= 2) wait_landed(std::true_type{}); else wait_landed(std::false_type{});     // tile 0 landed
  barrier();                                               // P: tile 0 is visible to every wave
  if (grp == 1) barrier();                                 // the second half falls one barrier behind
  int rs = 0, ws = D % NS;                                 // ring slot of tile t / of tile t + D
  int ktw = kt_begin + D;                                  // K tile whose weight pieces are staged next
  // one K tile.  MORE: tile t + D exists (its pieces are issued during this tile, and the wait for tile t + 1 leaves them in flight);
  // NEXT: tile t + 1 exists (it must have landed before the barrier in front of the first half's next load segment).
  auto tile = [&](auto more_c, auto next_c) {
    constexpr bool MORE = decltype(more_c)::value, NEXT = decltype(next_c)::value;
    const char* sb = smem + rs * STAGE;
    bool more = MORE;
    if constexpr (DBG) { if (p.dbg & 4) more = false; }
    if constexpr (NP == 1) {
      read_k(sb, 0, 0);
      read_k(sb, 1, 1);
      if (more) { stage_act(ws); stage_w(ws, ktw); }
      if constexpr (NEXT) { if (grp == 1) wait_landed(more_c); }
      wait_lds_reads();
      barrier();
      mma();
      if constexpr (MORE) prep_tile();                     // scalars of tile t + 1 + D (harmless past the end: arguments only)
      if constexpr (NEXT) { if (grp == 0) wait_landed(more_c); }
      barrier();
    } else {
      read_k(sb, 0, 0);
      if (more) stage_act(ws);
      wait_lds_reads();
      barrier();
      mma();
      barrier();
      read_k(sb, 1, 0);
      if (more) stage_w(ws, ktw);
      if constexpr (NEXT) { if (grp == 1) wait_landed(more_c); }
      wait_lds_reads();
      barrier();
      mma();
      if constexpr (MORE) prep_tile();
      if constexpr (NEXT) { if (grp == 0) wait_landed(more_c); }
      barrier();
    }
    if (++rs == NS) rs = 0;
    if (++ws == NS) ws = 0;
    ++ktw;
  };
  {
    int t = 0;
    for (; t + D < nt; ++t) tile(std::true_type{}, std::true_type{});          // steady state
    for (; t + 1 < nt; ++t) tile(std::false_type{}, std::true_type{});         // drain: nothing left to stage
    tile(std::false_type{}, std::false_type{});                                // last tile
  }
  if (grp == 0) barrier();                                 // the first half waits for the second: every wave is done with the ring

  if constexpr (DBG) {
    if (p.dbg & 1) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) asm volatile("" ::"v"(acc[i][j]));
      return;
    }
  }
  if constexpr (F8) {                                      // per-output-channel weight scales (this lane's 4 consecutive channels of every n-tile)
    if (p.wscale) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int n = n0 + wn * TN + i * 16 + lg * 4;
        f4 w = {1.f, 1.f, 1.f, 1.f};
        for (int e = 0; e < 4; ++e) if (n + e < p.N) w[e] = p.wscale[n + e];
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] *= w;
      }
    }
  }
  // ---- epilogue: two passes of 128 rows through the shared scratch (wave (wm, wn) is quadrant (wm & 1, wn) of sub-block wm >> 1)
  f4 csum[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) csum[i] = (f4){0.f, 0.f, 0.f, 0.f};
  f2 lstat[MJ];
  if constexpr (LNF) {
    const float invK = 1.0f / (float)p.K;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      float s_ = ls[j], q_ = lq[j];
      s_ += __shfl_xor(s_, 16, 64); q_ += __shfl_xor(q_, 16, 64);
      s_ += __shfl_xor(s_, 32, 64); q_ += __shfl_xor(q_, 32, 64);
      const float mean = s_ * invK;
      lstat[j] = (f2){mean, rsqrtf(fmaxf(q_ * invK - mean * mean, 0.f) + p.ln_eps)};
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = n0 + wn * TN + i * 16 + lg * 4;
      if (n + 3 < p.N) csum[i] = *reinterpret_cast<const f4*>(p.ln_colsum + n);
    }
  }
  constexpr int BS = BM / 2;                              // rows of an epilogue pass
  float* const lbt = reinterpret_cast<float*>(smem + 4 * (BS / 2) * (TN + 4) * 4 + BS * 8 + 4 * BN * 8);    // behind the scratch, the LayerNorm table and the statistics table
  if (tid < BN) { lbt[tid] = lb_b; lbt[BN + tid] = lb_c0; lbt[2 * BN + tid] = lb_c1; }                     // (visible behind the first pass's barrier)
  const int lb_m1 = (lb_img0 + 1) * p.HoWo;
#pragma unroll
  for (int sm = 0; sm < 2; ++sm) {
    if ((wm >> 1) == sm) {
      if constexpr (LNF) {
        // (mean, rstd) of this wave's rows into the table igemm_scratch_write reads them from; the wave with the other channel half
        // writes the very same values to the very same slots, and every wave reads back only what it wrote itself
        f2* stats = reinterpret_cast<f2*>(smem + 4 * (BS / 2) * (TN + 4) * 4);
        if (lg == 0) {
#pragma unroll
          for (int j = 0; j < MJ; ++j) stats[(wm & 1) * (BS / 2) + j * 16 + lr] = lstat[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      igemm_scratch_write<BS, BN>(p, acc, csum, smem, (wm & 1) | (wn << 1), lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
    // (two items' loads in flight at a time: half of the accumulators is still live during the first pass)
    if (F8 && p.out8) igemm_epilogue<BS, BN, true, false, 2, true>(p, smem, m0 + sm * BS, n0, split, wid & 3, wid >> 2, lane, lbt, n0, lb_m1);
    else igemm_epilogue<BS, BN, false, false, 2, true>(p, smem, m0 + sm * BS, n0, split, wid & 3, wid >> 2, lane, lbt, n0, lb_m1);
    if (p.gn_part && m0 + sm * BS < p.M) igemm_gn_stats<BS, BN>(p, smem, m0 + sm * BS, n0, wid & 3, wid >> 2, lane);   // (block-uniform: the barrier inside is safe)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
  }
}

// =====================================================================================================================
// SHORT-K kernel (round 3): Linear / 1x1 convolution with K <= a few K tiles and many output tiles -- q|k|v, to_out, GEGLU projection
// (ff/linear.py:112-121, ff/nn.py:5-12, attention/attention.py:35-41 of the reference).  For these shapes every part of a launch of the
// kernels above is near a bound of its own -- block dispatch + prologue ~3.4 us per round of blocks, operand re-reads from L2, the MFMAs,
// the output stores at the HBM write rate -- but the parts run one AFTER the other (tools/geglu_dbg.py: 8.5 + 10 + 9 + 5 ~ 31.6 us for
// 8192 x 2560 x 320, where max() would be 10): a block is a serial chain and a CU holds two of them.  This kernel removes the seams:
//   * PERSISTENT: 2 blocks of 4 waves per CU for the whole launch, each walking its own list of 128 x 128 tiles -- no dispatch or argument
//     loads per tile, and the stores of tile i drain while tile i + 1 loads and multiplies (nothing ever waits for a store);
//   * all four waves load and compute (2 x 2 wave tiles of 64 x 64); 2-slot LDS-DMA ring, one s_barrier per K tile; the first K tile of
//     the NEXT output tile is issued before the epilogue of this one (cross-tile prefetch: the ring slot it lands in is not the one the
//     epilogue borrows);
//   * epilogue without a block barrier: LayerNorm fold / bias / GEGLU in registers on the accumulators (a lane owns 4 consecutive channels of
//     a pixel), rounded to fp16, transposed through a PRIVATE per-wave LDS patch (half a wave tile at a time) into 16-byte row segments,
//     residual added there, stored;
//   * the two blocks of a CU are independent programs: one's epilogue and first-tile latency overlap the other's MFMAs.
// S = 1 / stride 1 / no padding (rows are contiguous K vectors; the concat pair of the FF2 . proj_out fold is two sources), channel
// counts on the 64 grid, fp16, no split-K / statistics / time embedding (those launches keep the kernels above).
template <bool LNF>
__global__ void __launch_bounds__(256, 2) k_gemm_c4(const GemmP p) {
  constexpr int BM = 128, BN = 128, MJ = 4, NI = 4;
  constexpr int STAGE = (BM + BN) * 128;                  // 32 KiB
  constexpr int PATCH = 32 * 144;                         // per-wave transpose patch: 32 rows x (128 + 16) bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid & 1, wn = wid >> 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int sub = lane >> 3;
  const int cs = (lane & 7) ^ ((4 * (wid & 1) + (sub >> 1)) & 7);       // source chunk: pieces of a wave are 4 apart, so 8 g's parity is the wave's
  const unsigned lds0 = lds_off(smem);
  const int ntm = p.ntm, ntn = p.ntn, ntiles = ntm * ntn;
  const int nt = p.ktiles;
  const int gstep = gridDim.x;
  const int C1_ = p.C1, K_ = p.K, M_ = p.M, N_ = p.N;
  const i4v rs_x1 = raw_rsrc(p.x, p.x_bytes), rs_x2 = raw_rsrc(p.x2 ? p.x2 : p.x, p.x2_bytes), rs_w = raw_rsrc(p.w, p.w_bytes);
  const int C2_ = p.C2;
  const int fo = lr * 128 + ((lg ^ ((lr >> 1) & 7)) << 4);
  const int xo = wm * 64 * 128 + fo, wo_ = (BM + wn * 64) * 128 + fo;
  char* const patch = smem + STAGE + wid * PATCH;         // inside ring slot 1 (the next tile's first K tile lands in slot 0)
  f2* const stats = reinterpret_cast<f2*>(smem + 2 * STAGE);   // [4 waves][64 rows] halves of the LayerNorm row sums (behind the ring)

  // this wave's staging rows of a tile: activation pieces wid + 4 i (i < 4: rows 8 (wid + 4 i) + sub), weight pieces likewise
  int am[4];
  unsigned gw[4];
  auto setup = [&](int tile, int& m0, int& n0) {
    int tm, tn;
    if (p.order == 0) { tm = tile / ntn; tn = tile - tm * ntn; } else { tn = tile / ntm; tm = tile - tn * ntm; }
    m0 = tm * BM; n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 8 * (wid + 4 * i) + sub;
      am[i] = m < M_ ? m : -1;
      const int n = n0 + 8 * (wid + 4 * i) + sub;
      gw[i] = n < N_ ? (unsigned)(n * K_ + cs * 8) * 2u : TF_OOB;
    }
  };
  auto stage = [&](int slot, int kt) {                    // K tile kt of the tile whose rows are in (am, gw)
    const int c = kt * 64;
    const bool second = c >= C1_;
    const int ld = second ? C2_ : C1_;
    const int cc = (second ? c - C1_ : c) + cs * 8;
    const i4v rs = second ? rs_x2 : rs_x1;
    const unsigned base = lds0 + (unsigned)slot * STAGE + (unsigned)wid * 1024u;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(rs, am[i] >= 0 ? (unsigned)(am[i] * ld + cc) * 2u : TF_OOB, base + (unsigned)i * 4096u);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(rs_w, gw[i] != TF_OOB ? gw[i] + (unsigned)kt * 128u : TF_OOB, base + 16384u + (unsigned)i * 4096u);
  };
  auto barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // this block's tiles: chunks of `chunk` consecutive tiles of the list, the chunks strided by the grid.  Consecutive tiles (n-fastest order)
  // share their 128 rows -- L1 / L2 lines, and with the LayerNorm fold the row statistics, computed for the first tile of a run only --
  // while the blocks running at the same time stay next to each other in the list (whole runs per block, each block on rows of its own, cost
  // the wide-N shapes 5-15 %)
  const int chunk = p.c4_chunk;
  int cq_ = blockIdx.x, ce_ = 0;                          // chunk index, tile inside the chunk
  int tile = cq_ * chunk;
  if (tile >= ntiles) return;
  auto next_tile = [&](int& q, int& e) {                  // -> tile index or -1
    if (e + 1 < chunk && q * chunk + e + 1 < ntiles) { ++e; return q * chunk + e; }
    q += gstep; e = 0;
    return q * chunk < ntiles ? q * chunk : -1;
  };
  int m0, n0;
  setup(tile, m0, n0);
  stage(0, 0);
  float ln_mean[MJ], ln_rstd[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j) { ln_mean[j] = 0.f; ln_rstd[j] = 0.f; }
  int stat_m0 = -1;
  int pend = 0;                                           // stores issued behind the prefetch of this tile's K tile 0 (0: unknown -> full wait)
  while (tile >= 0) {
    const bool need_stats = LNF && m0 != stat_m0;
    f4 acc[NI][MJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    float ls[MJ], lq[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j) { ls[j] = 0.f; lq[j] = 0.f; }
    // bias (and LayerNorm column sums) of this lane's columns: requested now, consumed behind the K loop -- and BEFORE the next tile's
    // prefetch is issued: the compiler counts only its own loads, so a wait for them placed behind the asm LDS-DMA would wait for the DMA too
    const int nb = n0 + wn * 64;                           // first (packed) column of the wave tile
    h4 braw[NI];                                           // (kept as loaded: a conversion here would put the compiler's vmcnt(0) here)
    f4 cq[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      braw[i] = (h4){(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f}; cq[i] = (f4){0.f, 0.f, 0.f, 0.f};
      int n = nb + i * 16 + lg * 4;
      n = n + 3 < N_ ? n : 0;                              // columns beyond N are never stored: any readable address will do (no masked load)
      if (p.bias) braw[i] = *reinterpret_cast<const h4*>(p.bias + n);
      if constexpr (LNF) cq[i] = *reinterpret_cast<const f4*>(p.ln_colsum + n);
    }
    const int young = pend > 0 ? pend + (p.bias ? NI : 0) + (LNF ? NI : 0) : 0;
    // ---- K loop: tile t in slot t & 1; the wait + barrier at the top make tile t visible and slot (t + 1) & 1 free
    for (int t = 0; t < nt; ++t) {
      // K tile t has landed.  For t = 0 it was issued in front of the previous tile's epilogue: where that epilogue's vector-memory
      // instructions are known to be `pend` stores, followed by this tile's bias / column-sum loads and nothing else, those `young`
      // ones stay in flight (the counter retires in issue order)
      if (t == 0 && young == 4) wait_vm<4>();
      else if (t == 0 && young == 8) wait_vm<8>();
      else if (t == 0 && young == 12) wait_vm<12>();
      else if (t == 0 && young == 16) wait_vm<16>();
      else wait_vm<0>();
      barrier();
      if (t + 1 < nt) stage((t + 1) & 1, t + 1);
      const char* sb = smem + (t & 1) * STAGE;
      h8 wf[2][NI], xf[2][MJ];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) xf[f][j] = *reinterpret_cast<const h8*>(sb + ((xo + j * 2048) ^ (f * 64)));
#pragma unroll
        for (int i = 0; i < NI; ++i) wf[f][i] = *reinterpret_cast<const h8*>(sb + ((wo_ + i * 2048) ^ (f * 64)));
      }
      wait_lds_reads();
      __builtin_amdgcn_sched_barrier(0);
      if (LNF && need_stats) {
        // row statistics from the fragments: the two waves that share these 64 rows (wn = 0, 1) take one 32-deep k-step each
        typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
        const hh2 one2 = {(_Float16)1.0f, (_Float16)1.0f};
        auto acc_stats = [&](const h8 (&x)[MJ]) {
#pragma unroll
          for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              hh2 v = {x[j][2 * e], x[j][2 * e + 1]};
              ls[j] = __builtin_amdgcn_fdot2(v, one2, ls[j], false);
              lq[j] = __builtin_amdgcn_fdot2(v, v, lq[j], false);
            }
        };
        if (wn == 0) acc_stats(xf[0]); else acc_stats(xf[1]);
      }
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[f][i], xf[f][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (LNF && need_stats) {                               // this wave's half of the row sums -> LDS, the partner's half comes back behind the barrier
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        float s_ = ls[j], q_ = lq[j];
        s_ += __shfl_xor(s_, 16, 64); q_ += __shfl_xor(q_, 16, 64);
        s_ += __shfl_xor(s_, 32, 64); q_ += __shfl_xor(q_, 32, 64);
        ls[j] = s_; lq[j] = q_;
        if (lg == 0) stats[wid * 64 + j * 16 + lr] = (f2){s_, q_};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a raw s_barrier does not wait for LDS stores)
    }
    barrier();                                             // every wave is done with the ring
    // ---- LayerNorm fold and bias on the accumulators (registers)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      asm volatile("" : "+v"(braw[i]));                    // (the values are used from here on: nothing of this moves in front of the K loop)
      if constexpr (LNF) asm volatile("" : "+v"(cq[i]));
    }
    if constexpr (LNF) {
      if (need_stats) {
        const float invK = 1.0f / (float)K_;
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          const f2 o_ = stats[(wid ^ 2) * 64 + j * 16 + lr];
          const float s_ = ls[j] + o_[0], q_ = lq[j] + o_[1];
          ln_mean[j] = s_ * invK;
          ln_rstd[j] = rsqrtf(fmaxf(q_ * invK - ln_mean[j] * ln_mean[j], 0.f) + p.ln_eps);
        }
        stat_m0 = m0;
      }
#pragma unroll
      for (int j = 0; j < MJ; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = ln_rstd[j] * (acc[i][j] - ln_mean[j] * cq[i]);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[i][j] += (f4){(float)braw[i][0], (float)braw[i][1], (float)braw[i][2], (float)braw[i][3]};
    asm volatile("" ::: "memory");
    // ---- the next tile's rows and its first K tile (slot 0), in flight during the rest of this tile's epilogue
    const int cm0 = m0, cn0 = n0;
    const int next = next_tile(cq_, ce_);
    if (next >= 0) { setup(next, m0, n0); stage(0, 0); }
    const bool geglu = p.act == 1;
    const int No = geglu ? N_ >> 1 : N_;
    // an interior tile without a residual stores 2 halves x 32 rows x cpr chunks / 64 lanes = 8 (GEGLU: 4) times per wave, every lane active
    pend = (cm0 + BM <= M_ && cn0 + BN <= N_ && !p.residual) ? (geglu ? 4 : 8) : 0;
    const unsigned pa = lds_off(patch);
    // two halves of the wave tile (pixel tiles j = 2 h, 2 h + 1: 32 rows) through the private patch: rows of 64 (32 with GEGLU) fp16
    const int ocols = geglu ? 32 : 64;                     // output columns of the wave tile
    const int nbc = cn0 + wn * 64;
    const int ocol0 = geglu ? (nbc >> 1) : nbc;            // packed column -> output column (n >> 5) * 16 + (n & 15) = n / 2 for n a multiple of 32
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * h + jj;
        const unsigned rowa = pa + (unsigned)(jj * 16 + lr) * 144u;
        if (geglu) {
#pragma unroll
          for (int i = 0; i < NI; i += 2) {
            h4 o;
            for (int e = 0; e < 4; ++e) o[e] = (half_t)(acc[i][j][e] * gelu_f(acc[i + 1][j][e]));
            asm volatile("ds_write_b64 %0, %1" ::"v"(rowa + (unsigned)((i >> 1) * 32 + lg * 8)), "v"(o) : "memory");
          }
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            h4 o;
            for (int e = 0; e < 4; ++e) o[e] = (half_t)acc[i][j][e];
            asm volatile("ds_write_b64 %0, %1" ::"v"(rowa + (unsigned)(i * 32 + lg * 8)), "v"(o) : "memory");
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // read back as rows: 32 rows x (ocols / 8) 16-byte chunks
      const int cpr = ocols >> 3;                          // 8 or 4 chunks per row
      for (int idx = lane; idx < 32 * cpr; idx += 64) {
        const int row = idx / cpr, c8 = idx - row * cpr;
        const int m = cm0 + wm * 64 + h * 32 + row, no = ocol0 + c8 * 8;
        h8 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(pa + (unsigned)row * 144u + (unsigned)c8 * 16u) : "memory");
        if (m < M_ && no < No) {
          const long long o = (long long)m * No + no;
          if (p.residual) { h8 r = *reinterpret_cast<const h8*>(p.residual + o); for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)r[e]); }
          *reinterpret_cast<h8*>(p.y + o) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    tile = next;
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
      f4 v = {0.f, 0.f, 0.f, 0.f};
      for (int z0 = 0; z0 < splitk; z0 += 8) {           // 8 independent loads in flight, added in split order
        f4 u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = z0 + i < splitk ? *reinterpret_cast<const f4*>(partial + (long long)(z0 + i) * total + e0) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) v += u[i];
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

// split-K reduce + epilogue + GroupNorm statistics of the output (tf_conv2d_fused_f16 on a split-K shape): a block owns R
// whole output rows (R = HoWo / chunks); thread t owns column quad t % nq and the rows r = t / nq (mod RL), so the
// per-channel sums stay in its registers; row lanes and channels -> groups meet through LDS in a fixed order.
// The split partials of an element are fetched 8 at a time (independent loads) and added in split order.
__global__ void __launch_bounds__(1024) k_splitk_reduce_gn(half_t* __restrict__ y, const float* __restrict__ partial, const half_t* __restrict__ bias,
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
    if (bias) { h4 b = *reinterpret_cast<const h4*>(bias + n); for (int e = 0; e < 4; ++e) bv[e] = (float)b[e]; }
    for (int r = rl; r < R; r += RL) {
      const int m = m_first + r;
      const long long e0 = (long long)m * N + n;
      h4 bnc = {0, 0, 0, 0}, res = {0, 0, 0, 0};
      if (bias_nc) bnc = *reinterpret_cast<const h4*>(bias_nc + (long long)(m / HoWo) * bnc_stride + n);
      if (residual) res = *reinterpret_cast<const h4*>(residual + e0);
      f4 v = {0.f, 0.f, 0.f, 0.f};
      for (int z0 = 0; z0 < splitk; z0 += 8) {
        f4 u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = z0 + i < splitk ? *reinterpret_cast<const f4*>(partial + (long long)(z0 + i) * total + e0) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) v += u[i];
      }
      v += bv;
      for (int e = 0; e < 4; ++e) v[e] += (float)bnc[e];
      for (int e = 0; e < 4; ++e) v[e] += (float)res[e];
      h4 o;
      for (int e = 0; e < 4; ++e) { o[e] = (half_t)v[e]; float f = (float)o[e]; cs[e] += f; cq[e] += f * f; }
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
__global__ void __launch_bounds__(1024) k_splitk_reduce_gn_apply(half_t* __restrict__ y, half_t* __restrict__ z, const float* __restrict__ partial,
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
  if (act) {
    if (bias) { h4 b = *reinterpret_cast<const h4*>(bias + n); for (int e = 0; e < 4; ++e) bv[e] = (float)b[e]; }
    if (bias_nc) bnc = *reinterpret_cast<const h4*>(bias_nc + (long long)img * bnc_stride + n);
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
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int z0 = 0; z0 < splitk; z0 += 8) {
        f4 u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = z0 + i < splitk ? *reinterpret_cast<const f4*>(partial + (long long)(z0 + i) * total + e0) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += u[i];
      }
      acc += bv;
      for (int e = 0; e < 4; ++e) acc[e] += (float)bnc[e];
      for (int e = 0; e < 4; ++e) acc[e] += (float)res[e];
      h4 o;
      for (int e = 0; e < 4; ++e) { o[e] = (half_t)acc[e]; float f = (float)o[e]; cs[e] += f; cq[e] += f * f; }
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
    h4 gm = {1, 1, 1, 1}, bt = {0, 0, 0, 0};
    if (gamma) { gm = *reinterpret_cast<const h4*>(gamma + n); bt = *reinterpret_cast<const h4*>(beta + n); }
    for (int e = 0; e < 4; ++e) {
      const int g = (v * 4 + e) / cpg;
      a[e] = st[2 * g + 1] * (float)gm[e];
      b[e] = (float)bt[e] - st[2 * g] * a[e];
    }
  }
#pragma unroll
  for (int k = 0; k < RGA_MAXR; ++k) {
    const int r = rl + k * RPS;
    if (r < HoWo) {
      h4 o;
      for (int e = 0; e < 4; ++e) { float f = (float)out[k][e] * a[e] + b[e]; o[e] = (half_t)(do_silu ? silu_f(f) : f); }
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

// LayerNorm fold of a Linear weight (one wave per output row n):
//   w'[n,k] = fp16(w[n,k] * gamma[k]);  colsum[n] = sum_k float(w'[n,k]);  bias'[n] = sum_k beta[k] * w[n,k] + bias[n]
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
      o[j] = (half_t)((float)v[j] * (float)g[j]);
      cs += (float)o[j];
      bs += (float)b[j] * (float)v[j];
    }
    *reinterpret_cast<h8*>(wo + (long long)n * K + k) = o;
  }
  cs = wave_sum(cs); bs = wave_sum(bs);
  if (l == 0) { colsum[n] = cs; bo[n] = (half_t)(bs + (bias ? (float)bias[n] : 0.f)); }
}

// ------------------------------------------------------------------------------------------------
// per-launch event profiling of this kernel family (bench.py roofline leg)
static bool g_prof = false;
static double g_prof_ms = 0.0, g_prof_ms_full = 0.0, g_prof_flops = 0.0;
static long long g_prof_launches = 0;
struct ProfRec { hipEvent_t a, b, c; bool has_reduce; double flops; int M, N, K, taps, bm, bn, splitk, variant; };   // a .. b: the GEMM kernel alone; a .. c: with the split-K reduce that finishes it
#include <map>
#include <array>
static std::map<std::array<int, 8>, std::pair<long long, double>> g_prof_shapes;
static std::vector<ProfRec> g_prof_pending;

static int g_dbg = 0, g_force_wide = -1, g_force_order = -1;
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
#define TF_GN_MAX_CHUNKS 128
static int gn_reduce_chunks(int HoWo) { int R = (HoWo + TF_GN_MAX_CHUNKS - 1) / TF_GN_MAX_CHUNKS; while (HoWo % R) ++R; return HoWo / R; }
static int gn_pieces(const GemmP& p, int bn) { return bn % p.gn_cpg == 0 ? 1 : 2; }   // chunks per m-tile (igemm_gn_stats)
static int gn_chunks_for(const GemmP& p, TileCfg c, int splitk) {
  return splitk > 1 ? gn_reduce_chunks(p.HoWo) : gn_pieces(p, c.bn) * (p.HoWo / c.bm);
}
static bool gn_tile_ok(const GemmP& p, int bm, int bn) { return p.HoWo % bm == 0 && gn_pieces(p, bn) * (p.HoWo / bm) <= TF_GN_MAX_CHUNKS; }

static int gi_table_bytes(const GemmP& p) { return p.gi_part ? (p.gi_G + p.C) * 8 : 0; }
// LDS of a k_igemm<bm, bn> launch without the gi table (ring or epilogue scratch, whichever is larger)
static int igemm_lds_bytes(int bm, int bn, bool wide) {
  const int ring = (wide ? 2 : ring_slots(bm, bn)) * (bm + bn) * 128;
  const int scratch = 4 * (bm / 2) * (bn / 2 + 4) * 4, tail = bm * 8 + 4 * bn * 8;
  return ring > scratch + tail ? ring : scratch + tail;
}

template <int BM, int BN, bool GENERIC, bool WIDE, bool ALL8 = false>
static int launch_cfg3(const GemmP& p, hipStream_t st) {
  constexpr int TM = BM / 2, TN = BN / 2;
  constexpr int ring = (WIDE ? 2 : ring_slots(BM, BN)) * (BM + BN) * 128;
  constexpr int scratch = 4 * TM * (TN + 4) * 4;         // epilogue transpose scratch overlays the ring
  constexpr int tail = BM * 8 + 4 * BN * 8;              // LayerNorm (mean, rstd) table + GroupNorm column-sum table behind the scratch
  constexpr int smem = ring > scratch + tail ? ring : scratch + tail;
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm<BM, BN, GENERIC, WIDE, ALL8>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  if (p.gi_part) {                                       // the (mean, rstd) + (a, b) table of the input's GroupNorm sits behind everything else
    GemmP q = p;
    q.gi_off = (smem + 15) & ~15;
    const int total = q.gi_off + gi_table_bytes(p);
    if constexpr (GENERIC || (BM == 128 && BN == 160) || BM == 256) {
      tf_set_error("k_igemm<%d,%d>: this instance cannot carry the input GroupNorm", BM, BN); return TF_E_UNSUPPORTED;
    } else {
      if (total > 163840) { tf_set_error("k_igemm<%d,%d>: no room for the GroupNorm table (%d B)", BM, BN, total); return TF_E_UNSUPPORTED; }
      static bool attr_gi = false;
      if (!attr_gi) {
        TF_HIP(hipFuncSetAttribute((const void*)k_igemm<BM, BN, GENERIC, WIDE, ALL8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        attr_gi = true;
      }
      hipLaunchKernelGGL((k_igemm<BM, BN, GENERIC, WIDE, ALL8, true>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), total, st, q);
      TF_LAUNCH_CHECK();
      return TF_OK;
    }
  }
  hipLaunchKernelGGL((k_igemm<BM, BN, GENERIC, WIDE, ALL8>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
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
template <int BM, int BN>
static int launch_patch(const GemmP& p, hipStream_t st) {
  constexpr int TM = BM / 2, TN = BN / 2;
  constexpr int scratch = 4 * TM * (TN + 4) * 4, tail = BM * 8 + 4 * BN * 8;
  const int ring = 2 * p.pt_ppc * 1024 + p.pt_ns * p.pt_stage + gi_table_bytes(p);
  const int smem = ring > scratch + tail ? ring : scratch + tail;
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm_patch<BM, BN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm_patch<BM, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  if (p.gi_part) hipLaunchKernelGGL((k_igemm_patch<BM, BN, true>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), smem, st, p);
  else hipLaunchKernelGGL((k_igemm_patch<BM, BN, false>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

template <int BM, int BN>
static int launch8(const GemmP& p, hipStream_t st) {
  constexpr int TM = BM / 2, TN = BN / 2;
  constexpr int ring = ring_slots8(BM, BN) * (BM + BN) * 64;
  constexpr int scratch = 4 * TM * (TN + 4) * 4, tail = BM * 8 + 4 * BN * 8;
  constexpr int smem = ring > scratch + tail ? ring : scratch + tail;
  static_assert(smem <= 163840, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm8<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_igemm8<BM, BN>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
// k_igemm_pp (variant 4): 256 x BN tiles, every channel count on the 64 grid, no LayerNorm fold, no input GroupNorm, fp16 only
static bool gemm_generic(const GemmP& p);
static bool pp_ok(const GemmP& p, int bn, int bm = 256) {
  if (bn != 128 && bn != 160 && bn != 256) return false;
  if (bm != 256 && !(bm == 192 && bn != 256)) return false;
  if (p.bf16 || p.gi_part || gemm_generic(p)) return false;
  if (p.ln_colsum && (p.fp8 || p.S != 1 || p.stride != 1 || p.ups)) return false;   // the LayerNorm fold: linears, fp16
  if (p.fp8 && bn == 256) return false;                  // the e4m3 form holds a whole K tile's fragments: needs the three-slot ring
  if (p.bias_nc && p.HoWo < bm) return false;            // the epilogue's time-embedding table holds two images per tile
  return p.act != 1 || bn % 64 == 0;                     // GEGLU pairs 16-row value | gate blocks inside a wave tile
}
template <int BN, int NP, bool FASTA, bool F8 = false, bool H2 = false, int BM = 256, bool LNF = false>
static int launch_pp2(const GemmP& p, hipStream_t st) {
  constexpr int STAGE = (BM + BN) * 128, NS = (163840 / STAGE) >= 3 ? 3 : 2;
  constexpr int ring = NS * STAGE, scratch = 4 * (BM / 4) * (BN / 2 + 4) * 4, tail = (BM / 2) * 8 + 4 * BN * 8 + 3 * BN * 4;   // (+ the bias / time-embedding table)
  constexpr int smem = ring > scratch + tail ? ring : scratch + tail;
  static_assert(smem <= 163840, "LDS budget");
  if constexpr (!F8 && FASTA && BM == 256 && !LNF) {       // ablation build (tools/pp_dbg.py): the lean-addressing fp16 instances only
    if (p.dbg) {
      static bool attr_dbg = false;
      if (!attr_dbg) {
        TF_HIP(hipFuncSetAttribute((const void*)k_igemm_pp<BN, NP, FASTA, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        attr_dbg = true;
      }
      hipLaunchKernelGGL((k_igemm_pp<BN, NP, FASTA, true>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), smem, st, p);
      TF_LAUNCH_CHECK();
      return TF_OK;
    }
  }
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm_pp<BN, NP, FASTA, false, F8, H2, BM, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_igemm_pp<BN, NP, FASTA, false, F8, H2, BM, LNF>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
static int g_pp_np = 0;                                    // test / tuning hook: 0 = default phases per K tile, 1 / 2 = forced where admissible
template <int BN, int BM = 256>
static int launch_pp(const GemmP& p, hipStream_t st) {
  // lean addressing where the gather is a fixed pixel shift per tap: stride 1, no up-sampling, at most 31 taps
  const bool fast = p.stride == 1 && !p.ups && p.S * p.S <= 31;
  constexpr bool CAN1 = (163840 / ((BM + BN) * 128)) >= 3;
  const bool np1 = CAN1 && g_pp_np != 2;
  if constexpr (CAN1) {
    if (p.fp8) {
      const bool h2 = (p.C1 % 128) || (p.C2 % 128) || (p.C3 % 128) || (p.C4 % 128);
      if (fast) return h2 ? launch_pp2<BN, 1, true, true, true, BM>(p, st) : launch_pp2<BN, 1, true, true, false, BM>(p, st);
      return h2 ? launch_pp2<BN, 1, false, true, true, BM>(p, st) : launch_pp2<BN, 1, false, true, false, BM>(p, st);
    }
    if (p.ln_colsum) return launch_pp2<BN, 1, true, false, false, BM, true>(p, st);    // (pp_ok admits linears only: the lean addressing)
    if (np1) return fast ? launch_pp2<BN, 1, true, false, false, BM>(p, st) : launch_pp2<BN, 1, false, false, false, BM>(p, st);
  }
  if (p.fp8) { tf_set_error("k_igemm_pp: no e4m3 instance for a %d-wide tile", BN); return TF_E_UNSUPPORTED; }
  if constexpr (BM == 256) {
    if (p.ln_colsum) return launch_pp2<BN, 2, true, false, false, 256, true>(p, st);
    return fast ? launch_pp2<BN, 2, true>(p, st) : launch_pp2<BN, 2, false>(p, st);
  }
  else { tf_set_error("k_igemm_pp: the 192-row tile has the one-phase form only"); return TF_E_UNSUPPORTED; }
}
// rows of a tile as the GroupNorm-statistics code sees them: the ping-pong kernel's epilogue works in 128-row sub-blocks
static int stats_bm(int bm, int variant) { return variant == 4 ? bm / 2 : bm; }
// k_gemm_c4 (variant 5): the persistent short-K kernel -- linears / 1x1 stride-1 convolutions of fp16 operands whose channel counts sit on
// the 64 grid, one launch (no split-K), no statistics, no time-embedding bias; bias, residual, GEGLU and the LayerNorm fold ride along
static bool c4_ok(const GemmP& p) {
  if (p.fp8 || p.bf16 || p.gi_part || p.gn_part || p.bias_nc || p.out32 || p.out8 || p.on_z) return false;
  if (p.S != 1 || p.stride != 1 || p.pad != 0 || p.ups || p.C3 || p.C4 || p.K != p.Kc) return false;
  if ((p.C1 % 64) || (p.C2 % 64) || (p.N % 8) || p.M < 1) return false;
  return p.act == 0 || (p.act == 1 && p.N % 64 == 0);
}
static int c4_num_cus() {
  static int n = 0;
  if (!n) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256; }
  return n;
}
static int g_c4_chunk = getenv("TF_C4_CHUNK") ? atoi(getenv("TF_C4_CHUNK")) : 0;   // A/B: tiles per chunk of k_gemm_c4's walk (0 = per-shape choice)
static int launch_c4(const GemmP& p, hipStream_t st) {
  constexpr int smem = 2 * (128 + 128) * 128 + 4 * 64 * 8;   // the two-slot ring (the epilogue's patches live in slot 1) + the LayerNorm row-sum exchange
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_gemm_c4<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    TF_HIP(hipFuncSetAttribute((const void*)k_gemm_c4<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  const int tiles = p.ntm * p.ntn;
  GemmP q = p;
  // consecutive tiles per block.  Without the LayerNorm fold: one (chunks of 2-8 were 2-8 % faster on three narrow-N shapes and up to 6x slower
  // wherever they left fewer chunks than blocks).  With it (n-fastest order): 4 or 2 while every block still gets >= 4 chunks -- the statistics
  // of a row block are computed once per chunk
  int chunk = 1;
  if (p.ln_colsum && p.order == 0) chunk = tiles / 4 >= 8 * c4_num_cus() ? 4 : tiles / 2 >= 8 * c4_num_cus() ? 2 : 1;
  q.c4_chunk = g_c4_chunk > 0 ? g_c4_chunk : chunk;
  const int chunks = (tiles + q.c4_chunk - 1) / q.c4_chunk;
  const int grid = chunks < 2 * c4_num_cus() ? chunks : 2 * c4_num_cus();   // two resident blocks per CU walk the tile list
  if (p.ln_colsum) hipLaunchKernelGGL(k_gemm_c4<true>, dim3(grid), dim3(256), smem, st, q);
  else hipLaunchKernelGGL(k_gemm_c4<false>, dim3(grid), dim3(256), smem, st, q);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
// (a 256x128 tile spills: the compiler keeps two copies of the accumulator set to issue the two k halves independently)
static const int kTiles8[][2] = {{128, 128}, {64, 128}, {128, 64}, {256, 64}, {64, 64}};
static const int kNumTiles8 = 5;

static bool gemm_generic(const GemmP& p) { return (p.C1 % 64) != 0 || (p.C2 % 64) != 0 || (p.C3 % 64) != 0 || (p.C4 % 64) != 0; }
// bfloat16 instances: plain deep ring, one launch (no split-K: the reduce kernels are fp16), no statistics, no input GroupNorm
template <int BM, int BN, bool GENERIC>
static int launch_bf(const GemmP& p, hipStream_t st) {
  const int smem = igemm_lds_bytes(BM, BN, false);
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_igemm<BM, BN, GENERIC, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_igemm<BM, BN, GENERIC, false, false, false, true>), dim3(p.ntm * p.ntn * p.splitk), dim3(512), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
template <int BM, int BN, bool WIDE_OK>
static int launch_cfg(const GemmP& p, hipStream_t st, bool wide, bool all8 = false) {
  bool generic = gemm_generic(p);
  if (all8 && !generic) return launch_cfg3<BM, BN, false, false, true>(p, st);
  if (WIDE_OK && wide) return generic ? launch_cfg3<BM, BN, true, WIDE_OK>(p, st) : launch_cfg3<BM, BN, false, WIDE_OK>(p, st);
  return generic ? launch_cfg3<BM, BN, true, false>(p, st) : launch_cfg3<BM, BN, false, false>(p, st);
}

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
  if (variant == 4 && p.fp8) p.ktiles = (p.K + 127) / 128;   // the e4m3 ping-pong kernel's K tile is 128 elements (128 bytes of a row)
  p.ktiles_per_split = (p.ktiles + c.splitk - 1) / c.splitk;
  p.splitk = (p.ktiles + p.ktiles_per_split - 1) / p.ktiles_per_split;
  p.partial = (float*)workspace;
  p.ntm = (p.M + c.bm - 1) / c.bm;
  p.ntn = (p.N + c.bn - 1) / c.bn;
  float* gn_part = p.gn_part;
  if (gn_part) {
    // chunk geometry of the statistics partials: in-kernel (2 pieces per m-tile) or in the split-K reduce (row stripes)
    if (p.splitk > 1) { p.gn_chunks = gn_reduce_chunks(p.HoWo); p.gn_part = nullptr; }
    else p.gn_chunks = gn_pieces(p, c.bn) * (p.HoWo / stats_bm(c.bm, variant));
  }
  if (p.bf16) {
    const bool g = gemm_generic(p);
    if (c.bm == 128 && c.bn == 128) rc = g ? launch_bf<128, 128, true>(p, st) : launch_bf<128, 128, false>(p, st);
    else if (c.bm == 64 && c.bn == 64) rc = g ? launch_bf<64, 64, true>(p, st) : launch_bf<64, 64, false>(p, st);
    else { tf_set_error("run_gemm: no bfloat16 kernel for tile %dx%d", c.bm, c.bn); return TF_E_UNSUPPORTED; }
  }
  else if (p.fp8 && variant != 4) {
    if (c.bm == 128 && c.bn == 128) rc = launch8<128, 128>(p, st);
    else if (c.bm == 64 && c.bn == 128) rc = launch8<64, 128>(p, st);
    else if (c.bm == 128 && c.bn == 64) rc = launch8<128, 64>(p, st);
    else if (c.bm == 256 && c.bn == 64) rc = launch8<256, 64>(p, st);
    else if (c.bm == 64 && c.bn == 64) rc = launch8<64, 64>(p, st);
    else { tf_set_error("run_gemm: no fp8 kernel for tile %dx%d", c.bm, c.bn); return TF_E_UNSUPPORTED; }
  }
  else if (variant == 4) {
    if (!pp_ok(p, c.bn, c.bm)) { tf_set_error("run_gemm: the ping-pong kernel cannot run tile %dx%d of this launch", c.bm, c.bn); return TF_E_UNSUPPORTED; }
    if (c.bm == 192) rc = c.bn == 128 ? launch_pp<128, 192>(p, st) : launch_pp<160, 192>(p, st);
    else rc = c.bn == 128 ? launch_pp<128>(p, st) : c.bn == 160 ? launch_pp<160>(p, st) : launch_pp<256>(p, st);
  }
  else if (variant == 5) {
    if (!c4_ok(p) || c.bm != 128 || c.bn != 128 || p.splitk != 1) { tf_set_error("run_gemm: the persistent short-K kernel cannot run this launch (tile %dx%d, split %d)", c.bm, c.bn, p.splitk); return TF_E_UNSUPPORTED; }
    rc = launch_c4(p, st);
  }
  else if (variant == 2 && patch_setup(p, c.bm, c.bn)) {
    if (c.bm == 128 && c.bn == 160) rc = launch_patch<128, 160>(p, st);
    else if (c.bm == 64 && c.bn == 160) rc = launch_patch<64, 160>(p, st);
    else if (c.bm == 128 && c.bn == 128) rc = launch_patch<128, 128>(p, st);
    else rc = launch_patch<64, 128>(p, st);
  }
  else if (c.bm == 256 && c.bn == 128) {
    if (gemm_generic(p) || p.gi_part) { tf_set_error("run_gemm: the 256x128 tile needs channel counts on the 64 grid and no input GroupNorm"); return TF_E_UNSUPPORTED; }
    rc = launch_cfg3<256, 128, false, false>(p, st);
  }
  else if (c.bm == 128 && c.bn == 160) rc = launch_cfg<128, 160, false>(p, st, wide, all8);     // scratch 86 KB: one block per CU only
  else if (c.bm == 64 && c.bn == 160) rc = launch_cfg<64, 160, true>(p, st, wide, all8);
  else if (c.bm == 128 && c.bn == 128) rc = launch_cfg<128, 128, true>(p, st, wide, all8);
  else if (c.bm == 64 && c.bn == 128) rc = launch_cfg<64, 128, true>(p, st, wide, all8);
  else if (c.bm == 128 && c.bn == 64) rc = launch_cfg<128, 64, true>(p, st, wide, all8);
  else if (c.bm == 64 && c.bn == 64) rc = launch_cfg<64, 64, true>(p, st, wide, all8);
  else { tf_set_error("run_gemm: no kernel for tile %dx%d", c.bm, c.bn); return TF_E_UNSUPPORTED; }
  if (rc) return rc;
  if (g_prof_end) { TF_HIP(hipEventRecord(g_prof_end, st)); g_prof_end = nullptr; }   // the bracket holds k_igemm* alone (what rocprofv3 lists under that name)
  p.gn_part = gn_part;
  if (p.on_applied) *p.on_applied = 0;
  int rg_gpb = 0, rg_cv = 0, rg_rps = 0;
  size_t rg_lds = 0;
  if (p.splitk > 1 && p.on_z && p.gn_part && rga_geometry(p.HoWo, p.N, p.gn_G, &rg_gpb, &rg_cv, &rg_rps, &rg_lds)) {

    static bool attr_set = false;
    if (!attr_set) { TF_HIP(hipFuncSetAttribute((const void*)k_splitk_reduce_gn_apply, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
    const int nimg = p.M / p.HoWo;
    hipLaunchKernelGGL(k_splitk_reduce_gn_apply, dim3(nimg * (p.gn_G / rg_gpb)), dim3(1024), rg_lds, st, p.y, p.on_z, (const float*)p.partial, p.bias, p.bias_nc,
                       p.residual, p.M, p.N, p.HoWo, p.splitk, p.bias_nc_stride, p.gn_part, p.gn_G, p.gn_cpg, rg_gpb, p.on_gamma, p.on_beta, p.on_eps, p.on_silu,
                       rg_rps, rg_cv);
    TF_LAUNCH_CHECK();
    if (p.on_applied) *p.on_applied = 1;
  } else if (p.splitk > 1 && p.gn_part) {
    const int R = p.HoWo / p.gn_chunks, nq = p.N >> 2;
    int RL = 1024 / nq;
    if (RL > R) RL = R;
    int threads = (RL * nq + 63) & ~63;
    hipLaunchKernelGGL(k_splitk_reduce_gn, dim3(p.M / R), dim3(threads), (size_t)RL * p.N * 2 * sizeof(float), st, p.y, (const float*)p.partial,
                       p.bias, p.bias_nc, p.residual, p.M, p.N, p.HoWo, p.splitk, p.bias_nc_stride, p.gn_part, p.gn_G, p.gn_cpg, p.gn_chunks, R, RL);
    TF_LAUNCH_CHECK();
  } else if (p.splitk > 1) {
    long long nv = ((long long)p.M * p.N) >> 2;
    int grid = (int)((nv + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_splitk_reduce, dim3(grid), dim3(256), 0, st, p.y, (const float*)p.partial, p.bias, p.bias_nc, p.residual, p.M, p.N,
                       p.HoWo, p.splitk, p.bias_nc_stride);
    TF_LAUNCH_CHECK();
  }
  return TF_OK;
}

// ---- per-shape autotuner ("measure, don't guess"): the first eager call of a shape times every admissible
// (tile, split-K, ring variant) on the caller's own buffers with HIP events and caches the winner.  Never runs
// inside a stream capture (a captured shape that was never seen eagerly falls back to the cost model).
#define TF_SPLITK_WS_CAP ((size_t)64 << 20)
static bool g_autotune = true;
struct TunedCfg { TileCfg c; int variant; int order; };   // variant: see launch_one
static std::map<std::array<int, 10>, TunedCfg> g_tuned;

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
  for (int ci = 0; ci < (p.fp8 ? kNumTiles8 : 7); ++ci) {
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
    for (int sk = 1; sk <= 8; sk *= 2) {
      if (sk > 1 && (p.act == 1 || p.out32 || p.ln_colsum || p.ktiles / sk < 4 || !workspace || (size_t)sk * p.M * p.N * 4 > workspace_bytes)) break;
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
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  *out = bc;
  return TF_OK;
}

static int run_gemm(GemmP p, void* workspace, size_t workspace_bytes, int force_bm, int force_bn, int force_split, hipStream_t st, int* gn_chunks = nullptr) {
  p.ktiles = (p.K + 63) / 64;
  p.dbg = g_dbg;
  fast_div_magic((unsigned)p.HoWo, &p.dv_howo_mul, &p.dv_howo_shr);
  fast_div_magic((unsigned)p.Wo, &p.dv_wo_mul, &p.dv_wo_shr);
  TunedCfg t = {choose_tiles(p.M, p.N, p.K, p.act, true), 0, 0};
  bool tuned = false;
  if (p.gi_part) { t = gi_default(p); tuned = true; }    // (tuned: keep gi_default's variant unless the tuner knows better)
  if (p.fp8) {                                            // untuned fp8 default: the widest tile that still gives every CU a block
    long long b128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    t.c = {p.M >= 128 ? 128 : 64, p.act == 1 || p.N >= 128 ? 128 : 64, b128 >= 128 ? 1 : t.c.splitk};
    t.variant = 0; tuned = true;
  }
  if (p.bf16) {                                           // two tiles, no tuner: 128 x 128 once that gives every CU a block, else 64 x 64
    long long b128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    t.c = b128 >= 256 ? TileCfg{128, 128, 1} : TileCfg{64, 64, 1};
    t.variant = 0; t.order = 0; tuned = true;
  }
  else if (force_bm) {
    t.c = {force_bm, force_bn, force_split > 0 ? force_split : 1};
    t.order = g_force_order > 0 ? 1 : 0;
    if (p.gi_part) t.variant = p.S == 3 ? 2 : 0;
  } else if (g_autotune && !g_dbg) {
    std::array<int, 10> key = {p.M, p.N, p.K, p.C1, p.C2, p.S, p.stride, p.ups, p.act,
                               (p.bias ? 1 : 0) | (p.residual ? 2 : 0) | (p.bias_nc ? 4 : 0) | (p.ln_colsum ? 8 : 0) | (p.gi_part ? 16 : 0) | (p.fp8 ? 64 : 0) | (p.out8 ? 128 : 0) | (p.out32 ? 256 : 0)};   // (on_z shares the plain key: same tile, another reduce kernel)
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) {
      t = it->second; tuned = true;
      // a table row (shipped, or loaded from a user's file) whose tile cannot carry this launch's input GroupNorm -- the key holds
      // only a gi flag, not HoWo / H / W -- falls back to the first admissible tile instead of failing the forward
      if (p.gi_part && !gi_tile_ok(p, t.c.bm, t.c.bn, t.variant)) t = gi_default(p);
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
    int kps = (p.ktiles + t.c.splitk - 1) / t.c.splitk, eff = (p.ktiles + kps - 1) / kps;
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
    const int kps = (p.ktiles + t.c.splitk - 1) / t.c.splitk;
    rec.has_reduce = (p.ktiles + kps - 1) / kps > 1;
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
  g_dbg = (flags & 7) | ((flags & 4096) ? 8 : 0);         // 4096: no fragment reads (k_igemm_pp ablation build only)
  g_pp_np = (flags & 8192) ? 2 : 0;                       // 8192: k_igemm_pp with one phase per k-step even where the 3-slot ring allows one per K tile
  g_force_wide = (flags & 1024) ? 5 : (flags & 512) ? 4 : (flags & 256) ? 3 : (flags & 128) ? 2 : (flags & 16) ? 1 : (flags & 8) ? 0 : -1;   // 128 / 256 / 512 / 1024: the PATCH / ALL8 / ping-pong / persistent short-K variants where eligible
  g_force_order = (flags & 64) ? 1 : (flags & 32) ? 0 : -1;
  return TF_OK;
}
int tf_gemm_autotune(int on) { g_autotune = on != 0; if (!on) g_tuned.clear(); return TF_OK; }
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
    if (((f8 && wide != 4) || (bm == 256 && wide != 4)) && wide != 0) ok = false;
    if (wide == 4) ok = ((bm == 256 && (bn == 128 || bn == 160 || (!f8 && bn == 256))) || (bm == 192 && (bn == 128 || bn == 160))) && sk >= 1 && sk <= 32;
    // rows the tuner itself never emits: GEGLU (act = 1) pairs 16-row value|gate blocks inside a wave tile (bn % 64 == 0), and
    // neither GEGLU nor the LayerNorm fold (flag bit 8) can be split along K
    const int act = k[8], ln = k[9] & 8;
    if (act == 1 && (bn % 64) != 0) ok = false;
    if ((act == 1 || ln || (k[9] & 256)) && sk > 1) ok = false;
    if (wide == 5) ok = bm == 128 && bn == 128 && sk == 1 && !f8;
    if (ok) g_tuned[k] = {{bm, bn, sk}, wide < 0 || wide > 5 ? 0 : wide, order != 0 ? 1 : 0};
  }
  fclose(f);
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
  fprintf(f, "M,N,K,taps,bm,bn,splitk,variant,launches,total_ms,avg_us,tflops\n");   // variant: 0 deep ring, 1 wide, 2 patch, 3 all8, 4 ping-pong; times include the split-K reduce
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

int tf_conv2d_fused_norm_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                             const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                             void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                             size_t gn_partial_bytes, int gn_groups, int* gn_chunks, void* z, const void* z_gamma, const void* z_beta, float z_eps,
                             int z_silu, int* z_written, tfStream_t s) {
  TF_REQUIRE(gn_partial && gn_chunks && z && z_written, "tf_conv2d_fused_norm_f16: gn_partial, gn_chunks, z and z_written must be given");
  TF_REQUIRE((z_gamma == nullptr) == (z_beta == nullptr), "tf_conv2d_fused_norm_f16: gamma and beta must both be given or both NULL");
  *z_written = 0;
  GemmP ex = {};
  ex.on_z = (half_t*)z; ex.on_gamma = (const half_t*)z_gamma; ex.on_beta = (const half_t*)z_beta; ex.on_eps = z_eps; ex.on_silu = z_silu ? 1 : 0;
  ex.on_applied = z_written;
  return conv2d_impl(y, x, x2, w, bias, bias_nc, bias_nc_stride, residual, N, H, W, C1, C2, Cout, R, S, stride, pad, upsample, workspace,
                     workspace_bytes, (float*)gn_partial, gn_partial_bytes, gn_groups, gn_chunks, x3, x4, C3, C4, s, &ex);
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

int tf_conv2d_gn_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                     const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                     void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                     size_t gn_partial_bytes, int gn_groups, int* gn_chunks, const void* in_gamma, const void* in_beta, const void* in_partial,
                     int in_chunks, int in_groups1, const void* in_partial2, int in_chunks2, int in_groups2, int in_groups, float in_eps, int in_silu,
                     tfStream_t s) {
  TF_REQUIRE(!gn_partial || gn_chunks, "tf_conv2d_gn_f16: gn_chunks must be given with gn_partial");
  TF_REQUIRE(in_partial && in_chunks >= 1 && in_chunks <= 4096 && in_groups >= 1 && (C1 + C2) % in_groups == 0, "tf_conv2d_gn_f16: input statistics missing (chunks=%d groups=%d)", in_chunks, in_groups);
  TF_REQUIRE((in_gamma == nullptr) == (in_beta == nullptr), "tf_conv2d_gn_f16: gamma and beta must both be given or both NULL");
  GemmP gi = {};
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

int tf_conv2d_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                  const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                  void* workspace, size_t workspace_bytes, tfStream_t s) {
  return conv2d_impl(y, x, x2, w, bias, bias_nc, bias_nc_stride, residual, N, H, W, C1, C2, Cout, R, S, stride, pad, upsample, workspace,
                     workspace_bytes, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, s);
}

int tf_conv2d_fused_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                        const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                        void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                        size_t gn_partial_bytes, int gn_groups, int* gn_chunks, tfStream_t s) {
  TF_REQUIRE(!gn_partial || gn_chunks, "tf_conv2d_fused_f16: gn_chunks must be given with gn_partial");
  return conv2d_impl(y, x, x2, w, bias, bias_nc, bias_nc_stride, residual, N, H, W, C1, C2, Cout, R, S, stride, pad, upsample, workspace,
                     workspace_bytes, (float*)gn_partial, gn_partial_bytes, gn_groups, gn_chunks, x3, x4, C3, C4, s);
}

size_t tf_conv2d_fused_workspace(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample, int C3, int C4) {
  int Ho, Wo;
  if (stride < 1 || conv_geometry(H, W, R, S, stride, pad, upsample ? 1 : 0, &Ho, &Wo)) return 0;
  return gemm_workspace(N * Ho * Wo, Cout, R * S * (C1 + C2) + C3 + C4, 0);
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
  p.M = M; p.N = act == 1 ? 2 * N : N; p.K = K; p.Kc = K; p.C1 = K; p.C2 = 0; p.C = K;
  p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.HoWo = M; p.S = 1; p.stride = 1; p.pad = 0; p.ups = 0; p.act = act;
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

int tf_ln_fold_weights_f16(void* w_out, void* bias_out, void* colsum_out, const void* w, const void* bias, const void* gamma, const void* beta,
                           int N, int K, tfStream_t s) {
  TF_REQUIRE(w_out && bias_out && colsum_out && w && gamma && beta && N >= 1 && K % 8 == 0, "tf_ln_fold_weights_f16: bad arguments (K=%d)", K);
  hipLaunchKernelGGL(k_ln_fold, dim3(ceil_div(N, 4)), dim3(256), 0, tf_hs(s), (half_t*)w_out, (half_t*)bias_out, (float*)colsum_out, (const half_t*)w,
                     (const half_t*)bias, (const half_t*)gamma, (const half_t*)beta, N, K);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

int tf_linear_ln_f16(void* y, const void* x, const void* w_folded, const void* bias_folded, const void* colsum, const void* residual, int M, int N,
                     int K, int act, float eps, tfStream_t s) {
  TF_REQUIRE(y && x && w_folded && bias_folded && colsum, "tf_linear_ln_f16: null tensor");
  TF_REQUIRE(M >= 0 && N >= 1 && K >= 64 && K % 64 == 0, "tf_linear_ln_f16: K=%d must be a positive multiple of 64", K);
  TF_REQUIRE(act == 0 || act == 1, "tf_linear_ln_f16: act=%d", act);
  TF_REQUIRE((act == 1 ? N % 16 == 0 : N % 4 == 0), "tf_linear_ln_f16: N=%d must be a multiple of 4 (16 for GEGLU)", N);
  if (M == 0) return TF_OK;
  GemmP p = {};
  p.x = (const half_t*)x; p.w = (const half_t*)w_folded; p.y = (half_t*)y; p.bias = (const half_t*)bias_folded; p.residual = (const half_t*)residual;
  p.ln_colsum = (const float*)colsum; p.ln_eps = eps;
  p.M = M; p.N = act == 1 ? 2 * N : N; p.K = K; p.Kc = K; p.C1 = K; p.C2 = 0; p.C = K;
  p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.HoWo = M; p.S = 1; p.stride = 1; p.pad = 0; p.ups = 0; p.act = act;
  {
    long long xb = (long long)M * K * 2, wb = (long long)p.N * K * 2;
    TF_REQUIRE(xb < (1LL << 31) && wb < (1LL << 31), "tf_linear_ln_f16: tensors must be < 2 GiB each");
    p.x_bytes = (unsigned)xb; p.x2_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
  }
  return run_gemm(p, nullptr, 0, g_force_bm, g_force_bn, g_force_split, tf_hs(s));
}

int tf_gemv_f16(void* y, const void* x, const void* w, const void* bias, int M, int N, int K, int silu_input, tfStream_t s) {
  TF_REQUIRE(y && x && w && M >= 1 && M <= 8 && N >= 1 && K % 8 == 0, "tf_gemv_f16: needs 1 <= M <= 8 (M=%d) and K %% 8 == 0 (K=%d)", M, K);
  hipLaunchKernelGGL(k_gemv, dim3(ceil_div(N, 4)), dim3(256), 0, tf_hs(s), (half_t*)y, (const half_t*)x, (const half_t*)w, (const half_t*)bias, M, N, K, silu_input);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

}  // extern "C"
