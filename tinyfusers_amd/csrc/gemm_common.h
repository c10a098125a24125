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
#pragma once
#include "common.h"
#include <type_traits>
#include "../../include/tinyfusers_hip.h"
#include <vector>

#ifndef TF_IGEMM_STAMP
#define TF_IGEMM_STAMP 0
#endif
// The launch seam of k_igemm / k_igemm_patch (round 5; A/B in profiles/r05_ab.txt, tagged builds -DTF_IGEMM_PRE=2 / -DTF_IGEMM_EPRE=0):
//   TF_IGEMM_PRE  = ring stages issued in FRONT of barrier P.  99 = the whole ring (shipped; the round-4 behaviour).  2 was tried on the strength of the phase stamps -- the consumers reach K tile 0
//                   0.6-1.2 us earlier -- and shipped for part of the round: in the STEP it is 0.2 % SLOWER, four rounds of four on one box (the stages pushed behind the barrier land later, and the
//                   loaders' issue stream is the K loop's critical path): reverted.
//   TF_IGEMM_EPRE = epilogue loads (bias / residual / time embedding) requested when the K loop ends, under barriers X / Y (1, shipped; 0 = inside the epilogue): neutral within noise in the step.
#ifndef TF_IGEMM_PRE
#define TF_IGEMM_PRE 99
#endif
#ifndef TF_IGEMM_EPRE
#define TF_IGEMM_EPRE 1
#endif
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
  int dbg;              // ablation library only (-DTF_ABLATION, tools/gemm_dbg.py): 1 no stores, 2 no MFMA, 4 no staging; the shipped kernels never read it
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
  // mx: the e4m3 ACTIVATIONS are block scaled (common.h: mx_quant8) -- x / x2 hold their E8M0 bytes behind their codes (x_bytes / x2_bytes stay
  // the codes' byte counts) -- and so is an e4m3 output (out8; GEGLU only).  Such launches run on k_igemm_pp<F8> only; without it (fixed scale 1:
  // the round-2 API) only on k_igemm8
  int mx;
  int part16;           // the split-K partial slabs hold fp16 (p.partial then points at halves; N % 8 == 0): half the bytes of the seam, fp32 accumulation in the reducer
  // bfloat16 operands, bias, residual and output (tf_linear_bf16 / tf_conv2d_bf16): the plain deep ring with the bf16 MFMA, no split-K
  int bf16;
  // tf_linear_f32out_f16: the raw fp32 accumulators go to out32[m, n] (no bias / residual / activation, never split along K) -- the
  // q k^T scores of the unfused attention path, which must not be rounded to fp16 before the softmax; NULL = off
  float* out32;
  int c4_chunk;         // k_gemm_c4: consecutive tiles a block takes before it strides on by gridDim chunks (launch_c4)
#if TF_IGEMM_STAMP
  unsigned long long* stamp;   // diagnostic build only (tools/igemm_stamp.py): [block][8] s_memrealtime phase stamps of k_igemm, never in the shipped library
#endif
};

// TF_IGEMM_STAMP (defined above GemmP, default 0): python -m tinyfusers_amd.build --tag stamp16 -DTF_IGEMM_STAMP=1 -- k_igemm records where a
// launch's time goes (phase stamps of wave 0 and wave 4 of every block on the 100 MHz constant clock; tools/igemm_stamp.py)
// Ablation switches (kernels that skip work and return WRONG results by design, for tools/*_dbg.py) exist only in the second library
// built with -DTF_ABLATION (python -m tinyfusers_amd.build --ablation -> lib/libtinyfusers_hip_ablation.so, loaded through TF_LIB_PATH);
// in the shipped library every TF_ABL(...) is the constant 0 and the DBG template instances are not compiled.
#ifdef TF_ABLATION
#define TF_ABL(x) (x)
#else
#define TF_ABL(x) 0
#endif
typedef __amdgpu_buffer_rsrc_t rsrc_t;   // 128-bit buffer resource

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

// the once-read WEIGHT stream of a launch: default cache policy (shipped) or, in the tagged experiment build -DTF_W_NT=1, non-temporal (aux = 2)
#ifndef TF_W_NT
#define TF_W_NT 0
#endif
__device__ __forceinline__ void bload_lds16_w(rsrc_t rsrc, unsigned voffset_bytes, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset_bytes, 0, 0, TF_W_NT ? 2 : 0);
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
template <bool BF = false>
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
    float gm = p.gi_gamma ? e2f<BF>(p.gi_gamma[c]) : 1.0f, bt = p.gi_beta ? e2f<BF>(p.gi_beta[c]) : 0.0f;
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
template <bool BF = false>
__device__ __forceinline__ h8 gi_apply(h8 x, const float (&a)[8], const float (&b)[8], int do_silu, bool keep) {
  h8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float f = e2f<BF>(x[j]) * a[j] + b[j];
    o[j] = f2e<BF>(do_silu ? silu_f(f) : f);
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

// The epilogue's first batch of global loads (bias, time embedding, residual of the common 16-byte path), issued EARLY by k_igemm: the phase
// stamps (profiles/r05_small_gemm_stamps.txt) show 0.8-2.8 us between "K loop done" and "stores issued", most of it one global round trip that
// started only behind barrier X + scratch write + barrier Y.  A wave requests them when its K loop is over (the loader waves well before the
// consumers' last MFMAs) and igemm_epilogue consumes the registers instead of loading.  Same values, same arithmetic: results are bit-identical.
template <int BM, int BN, int KBMAX = 4>
struct EpiPre {
  static constexpr int ROWS = BM / 4, CPR = BN / 16, ITEMS = ROWS * CPR, ITER = (ITEMS + 63) / 64, KB = ITER < KBMAX ? ITER : KBMAX;
  h8 bv[KB], cv[KB], rv[KB];
  bool on;
};
template <int BM, int BN, int KBMAX = 4>
__device__ __forceinline__ void igemm_epilogue_prefetch(const GemmP& p, int m0, int n0, int w4, int half, int lane, EpiPre<BM, BN, KBMAX>& r) {
  typedef EpiPre<BM, BN, KBMAX> P;
  r.on = p.act != 1 && (p.N & 7) == 0 && p.splitk <= 1 && !p.out32;
  if (!r.on) return;
  constexpr int TM = BM / 2, TN = BN / 2;
  const int mb = m0 + (w4 & 1) * TM + half * P::ROWS, nb = n0 + (w4 >> 1) * TN;
#pragma unroll
  for (int k = 0; k < P::KB; ++k) {
    const int idx = lane + 64 * k;
    const int row = idx / P::CPR, c8 = idx - row * P::CPR;
    const int m = mb + row, n = nb + c8 * 8;
    if (idx < P::ITEMS && m < p.M && n < p.N) {
      if (p.bias) r.bv[k] = *reinterpret_cast<const h8*>(p.bias + n);
      if (p.bias_nc) r.cv[k] = *reinterpret_cast<const h8*>(p.bias_nc + (long long)(m / p.HoWo) * p.bias_nc_stride + n);
      if (p.residual) r.rv[k] = *reinterpret_cast<const h8*>(p.residual + (long long)m * p.N + n);
    }
  }
}

// all 8 waves: wave (w4, half) stores rows [half*TM/2, (half+1)*TM/2) of consumer w4's tile
// LB: bias and the time embedding (bias_nc) come from an fp32 LDS table `lb` the kernel filled for its tile ([0][BN]: bias, [1 + i][BN]:
//   bias_nc of image lb_img0 + i, i < 2) instead of per-item global loads -- the only loads left in the epilogue are the residual's.
template <int BM, int BN, int OUT8 = 0, bool BF = false, int KBMAX = 4, bool LB = false, bool PRE = false>     // PRE: the first batch of loads arrives in `pre` (igemm_epilogue_prefetch); OUT8: the output is stored as e4m3 -- 1: at scale 1 (k_igemm8), 2: block scaled (k_igemm_pp, GEGLU only: codes, then the E8M0 bytes behind the M x N/2 codes) -- a template parameter keeps it out of the fp16 kernels; BF: bias / residual / output are bfloat16; KBMAX: items whose loads are in flight together
__device__ __forceinline__ void igemm_epilogue(const GemmP& p, char* smem, int m0, int n0, int split, int w4, int half, int lane, const float* lb = nullptr, int lb_n0 = 0, int lb_m1 = 0,
                                               const EpiPre<BM, BN, KBMAX>& pre = EpiPre<BM, BN, KBMAX>()) {
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
      if (OUT8 != 2 && !ok[k]) continue;                  // (the block-scaled output shuffles across lanes: every lane stays, dead items contribute zeros)
      const int idx = lane + 64 * (k0 + k);
      const int row = idx / CPR, c8 = idx - row * CPR;
      const int m = mb + row;
      const int pc = 32 * (c8 >> 1) + 8 * (c8 & 1);      // packed column of the value chunk inside the wave tile
      const int n = nb + pc;
      const int no = (n >> 5) * 16 + (n & 15);
      const float* r = sc + (ok[k] ? row * RS + pc : 0);
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
      if (p.residual && ok[k]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (E)((float)o[e] + (float)rv[k][e]);
      }
      if constexpr (OUT8 == 2) {
        // CPR = 4 (a 64-wide wave tile): the 4 lanes of a row hold 32 consecutive output channels = one block
        static_assert(OUT8 != 2 || CPR == 4, "block-scaled GEGLU output: 128-wide tiles");
        f4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
        if (ok[k]) { for (int e = 0; e < 4; ++e) { q0[e] = (float)o[e]; q1[e] = (float)o[4 + e]; } }
        unsigned sb;
        const uint2 code = mx_quant8(q0, q1, sb);
        if (ok[k]) {
          unsigned char* yb = reinterpret_cast<unsigned char*>(p.y);
          *reinterpret_cast<uint2*>(yb + (long long)m * No + no) = code;
          if (c8 == 0) yb[(long long)p.M * No + (long long)m * (No >> 5) + (no >> 5)] = (unsigned char)sb;
        }
      } else if constexpr (OUT8 == 1) {
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
      if (PRE && k0 == 0) {                               // (k_igemm: this batch was requested when the wave's K loop ended; pre.on is implied by this branch)
        bv[k] = __builtin_bit_cast(E8, pre.bv[k]); cv[k] = __builtin_bit_cast(E8, pre.cv[k]); rv[k] = __builtin_bit_cast(E8, pre.rv[k]);
      } else if (ok[k]) {
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
      if constexpr (OUT8 != 0) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.y) + o) = pack8_fp8(v0, v1);
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
    if (part && p.part16 && p.splitk > 1) {                 // fp16 slab (N % 8 == 0: one 16-byte store per item)
      // (a partial is a SUB-sum: it may exceed fp16's range where the full sum does not, and (half_t) does not saturate -- clamp, so that an
      // out-of-range partial costs accuracy, not an inf / NaN in y; the bfloat16 launches keep fp32 slabs: launch_one)
      h8 hv;
      for (int e = 0; e < 4; ++e) { hv[e] = (half_t)__builtin_fminf(__builtin_fmaxf(v0[e], -65504.0f), 65504.0f); hv[4 + e] = (half_t)__builtin_fminf(__builtin_fmaxf(v1[e], -65504.0f), 65504.0f); }
      *reinterpret_cast<h8*>(reinterpret_cast<half_t*>(p.partial) + (long long)split * p.M * p.N + o) = hv;
      continue;
    }
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
      if constexpr (OUT8 != 0) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.y) + o) = pack8_fp8(v0, v1);
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

// raw buffer descriptor + LDS-DMA from inline asm (k_igemm_pp, k_gemm_c4): the compiler sees no LDS write, so it puts no vmcnt(0) in front of fragment reads
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
__device__ __forceinline__ void dma16_w(i4v rsrc, unsigned voffset_bytes, unsigned lds_base) {      // (weights: see TF_W_NT)
#if TF_W_NT
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds"
               :: "s"(__builtin_amdgcn_readfirstlane((int)lds_base)), "v"(voffset_bytes), "s"(rsrc) : "memory");
#else
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane((int)lds_base)), "v"(voffset_bytes), "s"(rsrc) : "memory");
#endif
}
__device__ __forceinline__ void dma16(i4v rsrc, unsigned voffset_bytes, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane((int)lds_base)), "v"(voffset_bytes), "s"(rsrc) : "memory");   // M0 has no other user in this kernel
}


// ---- shared by the launchers (one translation unit per kernel family) and the host code of gemm.hip -------------------------------
#define TF_PATCH_PPW 9     // patch pieces per loader wave at most (33 pieces: BM = 128, W = 64)
static inline bool gemm_generic(const GemmP& p) { return (p.C1 % 64) != 0 || (p.C2 % 64) != 0 || (p.C3 % 64) != 0 || (p.C4 % 64) != 0; }
static inline int gi_table_bytes(const GemmP& p) { return p.gi_part ? (p.gi_G + p.C) * 8 : 0; }
// LDS of a k_igemm<bm, bn> launch without the gi table (ring or epilogue scratch, whichever is larger)
static inline int igemm_lds_bytes(int bm, int bn, bool wide) {
  const int ring = (wide ? 2 : ring_slots(bm, bn)) * (bm + bn) * 128;
  const int scratch = 4 * (bm / 2) * (bn / 2 + 4) * 4, tail = bm * 8 + 4 * bn * 8;
  return ring > scratch + tail ? ring : scratch + tail;
}
// Launchers: each returns TF_OK or an error code; an instance that does not exist is TF_E_UNSUPPORTED with tf_last_error set.
//   k_igemm (gemm_k_igemm_{160,128,64}.hip by tile width; the 64 file also holds the 256 x 128 tile and the bfloat16 instances)
int tfk_launch_igemm_160(const GemmP& p, hipStream_t st, int bm, bool wide, bool all8);
int tfk_launch_igemm_128(const GemmP& p, hipStream_t st, int bm, bool wide, bool all8);
int tfk_launch_igemm_64(const GemmP& p, hipStream_t st, int bm, bool wide, bool all8);
int tfk_launch_igemm_256x128(const GemmP& p, hipStream_t st);
//   k_igemm_patch (gemm_k_patch.hip); the caller has run patch_setup
int tfk_launch_patch(const GemmP& p, hipStream_t st, int bm, int bn);
//   k_igemm8 (gemm_k_igemm8.hip)
int tfk_launch_igemm8(const GemmP& p, hipStream_t st, int bm, int bn);
//   k_igemm_pp: fp16 instances (gemm_k_pp16.hip; np_force = 2: one phase per k-step where the tile also has the one-phase form) and e4m3 (gemm_k_pp8.hip)
int tfk_launch_pp16(const GemmP& p, hipStream_t st, int bm, int bn, int np_force);
int tfk_launch_pp8(const GemmP& p, hipStream_t st, int bm, int bn);
//   k_gemm_c4 (gemm_k_c4.hip)
int tfk_launch_c4(const GemmP& p, hipStream_t st);
int tfk_launch_pp3(const GemmP& p, hipStream_t st, int bn);
//   k_gemm_c8 (gemm_k_c8.hip)
int tfk_launch_c8(const GemmP& p, hipStream_t st);
int tfk_launch_c8_bf16(const GemmP& p, hipStream_t st);
//   k_gemm_ar (gemm_k_ar.hip)
int tfk_launch_ar(const GemmP& p, hipStream_t st);
int tfk_launch_ar_bf16(const GemmP& p, hipStream_t st);
//   the bfloat16 instances of the same kernels (gemm_k_*_bf16.hip)
int tfk_launch_igemm_160_bf16(const GemmP& p, hipStream_t st, int bm, bool wide, bool all8);
int tfk_launch_igemm_128_bf16(const GemmP& p, hipStream_t st, int bm, bool wide, bool all8);
int tfk_launch_igemm_64_bf16(const GemmP& p, hipStream_t st, int bm, bool wide, bool all8);
int tfk_launch_igemm_256x128_bf16(const GemmP& p, hipStream_t st);
int tfk_launch_patch_bf16(const GemmP& p, hipStream_t st, int bm, int bn);
int tfk_launch_pp16_bf16(const GemmP& p, hipStream_t st, int bm, int bn, int np_force);
int tfk_launch_c4_bf16(const GemmP& p, hipStream_t st);
int tfk_launch_pp3_bf16(const GemmP& p, hipStream_t st, int bn);
