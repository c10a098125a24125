// Part of the implicit-GEMM family of csrc/gemm.hip (see its head comment); split into translation units so that the
// instances compile in parallel.
#pragma once
#include "gemm_common.h"

// =====================================================================================================================
// ACTIVATION-RESIDENT short-K kernel (round 5, variant 8): Linear / 1x1 convolution with K = 256 or 320 and no residual -- at SD's first level the GEGLU projection,
// q|k|v and to_q (ff/nn.py:5-12, attention/attention.py:35-41, ff/linear.py:112-121 of the reference).
//
// What bounds k_gemm_c4 on these shapes is the number of vector-memory wave-instructions a CU can retire: every LDS-DMA piece and every 16-byte-per-lane store
// is one 1-KiB instruction, and a CU gets through one per ~45-50 cycles whatever mix of waves issues them (profiles/r05_ar_stamps.txt: k_gemm_c4 spends 160 loads +
// 16 / 32 stores = 176 / 192 instructions and 8.5k / 10.1k cycles per 128 x 128 x 320 tile and CU; the matrix pipe needs 2.6k).  So the lever is instructions per tile:
//   * with K this short the whole K extent of a 128-row activation panel fits in LDS (5 x 16 KiB): ONE 8-wave block per CU keeps its panel RESIDENT and walks the
//     panel's n-tiles, staging only the 80 KiB of weights per tile -- half the load instructions -- through a 4-slot ring of 16-KiB K steps that runs on ACROSS
//     tiles;
//   * waves 4-7 own the vector memory: the LDS-DMA stream (counted vmcnt, one raw s_barrier per K step as in k_igemm) AND the output stores; waves 0-3 (2 x 2 wave
//     tiles of 64 x 64) only multiply: fragments of the other 32-deep half of a step are read under the MFMAs of this half;
//   * the epilogue is a pipeline, not a phase: at a tile's last step the consumers fold LayerNorm / bias / GEGLU into the accumulators in registers and round them
//     to 16 bits (32 registers); during the NEXT tile's first four K steps they drop one 16-row quarter per step into a per-wave LDS patch (two patches, swizzled
//     128-byte rows, conflict-free both ways), and behind that step's barrier the matching loader wave reads the quarter back as 16-byte row segments and stores
//     it -- the stores ride in the loader's instruction stream between two stages, the consumers never touch vector memory after the bias load and never wait for
//     the memory system; only the run's last tile is stored by the consumers themselves;
//   * a block owns a CONTIGUOUS run of the n-fastest tile list (every block the same count: no tail round); where the run crosses into the next panel, that
//     panel's K tile k replaces the old one right behind the last step that read it (ring depth 4 <= K tiles 4 / 5) -- no drain at the seam;
//   * the row statistics of the LayerNorm fold come from the fragments of the panel's first tile, each wave for its own 64 rows (no exchange).
// LDS: panel 80 KiB + ring 64 KiB + patches 16 KiB = all 160 KiB.  Launches: k_gemm_c4's (c4_ok) with K = 256 / 320 and no residual.
//
// Diagnostic build 3 (python -m tinyfusers_amd.build --tag stamp3 -DTF_IGEMM_STAMP=3; tools/ar_stamp.py): cycle sums (s_memtime) of consumer wave 0 and loader wave 4.
#ifndef TF_AR_EXP
#define TF_AR_EXP 0      // timing experiments (tagged builds only, WRONG results), a bit mask: 1 no stores by the loaders, 2 no MFMAs, 4 no fragment reads, 8 no LDS-DMA
#endif
template <bool LNF, bool GG, bool BF = false>   // GG: GEGLU (value / gate column tiles alternate: ff/nn.py:10-12) -- a template parameter, so that the epilogue is straight-line code
__global__ void __launch_bounds__(512, 1) k_gemm_ar(const GemmP p) {
  constexpr int BM = 128, BN = 128, MJ = 4, NI = 4, NS = 4;
  constexpr int IMG = 128 * 128;                          // one K step of 128 rows: 16 KiB
  constexpr int AMAX = 5;                                 // K tiles of the resident panel at most
  constexpr int RING = AMAX * IMG;                        // byte offset of the weight ring
  constexpr int PATCH = 16 * 128;                         // one quarter of a wave tile: 16 rows x 128 bytes, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7)
  constexpr int PATCH0 = RING + NS * IMG;                 // [consumer wave][2] patches
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = lds_off(smem);
  const int ntn = p.ntn, ntiles = p.ntm * ntn;
  const int nt = p.ktiles;
  const int M_ = p.M, N_ = p.N, K_ = p.K;
  constexpr bool geglu = GG;
  const int No = geglu ? N_ >> 1 : N_;
  const int t0 = (int)blockIdx.x * p.c4_chunk;            // this block's run of the tile list [t0, t1)
  if (t0 >= ntiles) return;
  const int t1 = t0 + p.c4_chunk < ntiles ? t0 + p.c4_chunk : ntiles;
  const int G = (t1 - t0) * nt;                           // K steps of the run: step g = K tile g % nt of tile t0 + g / nt, weights in ring slot g % NS
  const int tm0 = t0 / ntn, tn0 = t0 - tm0 * ntn;
  auto barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#if TF_IGEMM_STAMP == 3
#define AR_T(acc, ...) do { const unsigned long long t_a = __builtin_amdgcn_s_memtime(); __VA_ARGS__; const unsigned long long t_b = __builtin_amdgcn_s_memtime(); acc += t_b - t_a; } while (0)
#else
#define AR_T(acc, ...) do { __VA_ARGS__; } while (0)
#endif
  // the stored form of one patch quarter: 16 rows x cpr 16-byte chunks (8; GEGLU 4) -> one chunk per lane and pass, 2 (1) passes
  constexpr int csh = geglu ? 2 : 3, cpr = 1 << csh, npass = geglu ? 1 : 2;

  if (wid >= 4) {
    // =========================== loader waves: the weight stream, the panel images where the run enters a panel, and the stores ===========================
    const int lw = wid - 4;
    const int sub = lane >> 3;
    const int cs = (lane & 7) ^ ((4 * (lw & 1) + (sub >> 1)) & 7);        // source chunk of this lane's 16 bytes (the LDS image is lane-linear; pieces of a wave are 4 apart)
    const i4v rs_x1 = raw_rsrc(p.x, p.x_bytes), rs_x2 = raw_rsrc(p.x2 ? p.x2 : p.x, p.x2_bytes), rs_w = raw_rsrc(p.w, p.w_bytes);
    const int C1_ = p.C1, C2_ = p.C2;
    int am[4];
    unsigned gw[4];
    auto rows_a = [&](int m0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int m = m0 + 8 * (lw + 4 * i) + sub; am[i] = m < M_ ? m : -1; }
    };
    auto rows_w = [&](int n0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int n = n0 + 8 * (lw + 4 * i) + sub; gw[i] = n < N_ ? (unsigned)(n * K_ + cs * 8) * 2u : TF_OOB; }
    };
    int l_tm = tm0, l_tn = tn0, l_kt = 0, l_g = 0;
    bool l_new = true;                                    // the cursor's tile is the first one of its panel in this run: its stages carry the panel images
    rows_a(l_tm * BM); rows_w(l_tn * BN);
    auto issue = [&]() -> int {                           // stage l_g (if the run has one); returns the pieces this wave issued
      if (l_g >= G || (TF_AR_EXP & 8)) return 0;
      const unsigned wbase = lds0 + RING + (unsigned)(l_g & (NS - 1)) * IMG + (unsigned)lw * 1024u;
      int q = 4;
      if (l_new) {
        const int c = l_kt * 64;
        const bool second = c >= C1_;
        const int ld = second ? C2_ : C1_;
        const int cc = (second ? c - C1_ : c) + cs * 8;
        const i4v rs = second ? rs_x2 : rs_x1;
        const unsigned abase = lds0 + (unsigned)l_kt * IMG + (unsigned)lw * 1024u;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rs, am[i] >= 0 ? (unsigned)(am[i] * ld + cc) * 2u : TF_OOB, abase + (unsigned)i * 4096u);
        q = 8;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) dma16_w(rs_w, gw[i] != TF_OOB ? gw[i] + (unsigned)l_kt * 128u : TF_OOB, wbase + (unsigned)i * 4096u);
      ++l_g;
      if (++l_kt == nt) {
        l_kt = 0;
        if (++l_tn == ntn) { l_tn = 0; ++l_tm; rows_a(l_tm * BM); l_new = true; } else l_new = false;
        rows_w(l_tn * BN);
      }
      return q;
    };
    // at most a of this wave's vector-memory operations in flight (a = what it issued behind the stage waited for: loads and stores count together, in order)
    auto wait_dyn = [&](int a) {
      switch (a) {
#define AR_W(n) case n: wait_vm<n>(); break;
        AR_W(1) AR_W(2) AR_W(3) AR_W(4) AR_W(5) AR_W(6) AR_W(7) AR_W(8) AR_W(9) AR_W(10) AR_W(11) AR_W(12) AR_W(13) AR_W(14) AR_W(15) AR_W(16) AR_W(17) AR_W(18) AR_W(19) AR_W(20) AR_W(21) AR_W(22)
#undef AR_W
        default: if (a > 22) wait_vm<22>(); else wait_vm<0>();
      }
    };
    // the stores: consumer wave lw's patch (quarter j of its 64 x 64 tile of output tile (s_tm, s_tn)) -> 16-byte row segments
    const int wm = lw & 1, wn = lw >> 1;
    const unsigned pbase = lds0 + PATCH0 + (unsigned)lw * 2u * PATCH;
    unsigned pra[2];                                      // this lane's chunk of each pass: LDS offset inside a patch
    int prow[2], pcol[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = lane + 64 * it;
      const int row = idx >> csh, c8 = idx & (cpr - 1);
      prow[it] = row; pcol[it] = c8 * 8;
      pra[it] = (unsigned)(row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4));
    }
    h8 fv0, fv1 = {};                                     // the quarter being stored: read from the patch in front of the stage issue (its LDS latency hides under the DMA issue), stored behind it
    auto flush_read = [&](int j) {
      const unsigned pb = pbase + (unsigned)(j & 1) * PATCH;
      asm volatile("ds_read_b128 %0, %1" : "=v"(fv0) : "v"(pb + pra[0]) : "memory");
      if (npass == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(fv1) : "v"(pb + pra[1]) : "memory");
    };
    auto flush_store = [&](int s_tm, int s_tn, int j) -> int {   // returns the stores issued
      const int mb = s_tm * BM + wm * 64 + j * 16;
      const int nbc = s_tn * BN + wn * 64;
      const int ocol0 = geglu ? (nbc >> 1) : nbc;          // packed column -> output column (n >> 5) * 16 + (n & 15) = n / 2 for n a multiple of 32
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fv0), "+v"(fv1) :: "memory");
      // a pass is issued iff its first lane stores (rows and columns ascend with the lane): a wave-uniform test, so the count below is exact -- a wait that
      // allowed one operation more than are in flight could return with a piece of the awaited stage still on its way
      int d = 0;
      if (mb < M_ && ocol0 < No) {
        const int m = mb + prow[0], no = ocol0 + pcol[0];
        if (m < M_ && no < No) *reinterpret_cast<h8*>(p.y + (long long)m * No + no) = fv0;
        ++d;
      }
      if (npass == 2 && mb + 8 < M_ && ocol0 < No) {
        const int m = mb + prow[1], no = ocol0 + pcol[1];
        if (m < M_ && no < No) *reinterpret_cast<h8*>(p.y + (long long)m * No + no) = fv1;
        ++d;
      }
      return d;
    };
#if TF_IGEMM_STAMP == 3
    unsigned long long a_wait = 0, a_bar = 0, a_iss = 0, a_fl = 0;
    const unsigned long long a_t0 = __builtin_amdgcn_s_memtime();
#endif
    // Ring protocol (k_igemm's): barrier P hands step 0 to the consumers; barrier(g) guarantees step g + 1 has landed (the consumers read its fragments while
    // multiplying step g) and hands slot g % NS -- and, at a panel seam, panel image (g + NS) % nt -- back: the consumers drained their reads of step g before
    // arriving.  Behind barrier(g) the consumers' quarter g % nt of the PREVIOUS tile sits in patch (g % nt) & 1; it is rewritten two steps later.
    issue();
    const int qb = issue();
    wait_dyn(qb);                                         // stage 0 landed, stage 1 in flight
    barrier();                                            // barrier P
    int q2 = issue(), q1 = issue();                       // stages 2, 3
    int d1 = 0, d2 = 0, d3 = 0;                           // stores issued behind the last three stages
    int u_tm = tm0, u_tn = tn0, s_tm = -1, s_tn = 0, u_kt = 0;   // the consumers' tile, the tile before it (-1: none), the K tile of step g
    for (int g = 0; g < G; ++g) {
      AR_T(a_wait, if (g + 1 < G) wait_dyn(d3 + q2 + q1));   // stage g + 1 landed; behind it in flight: the stores of iteration g - 3, stages g + 2, g + 3 and their stores
      AR_T(a_bar, barrier());                             // barrier(g)
      int q, d = 0;
      const bool fl = !(TF_AR_EXP & 1) && s_tm >= 0 && u_kt < 4;
      if (fl) flush_read(u_kt);
      AR_T(a_iss, q = issue());                           // stage g + NS
      AR_T(a_fl, if (fl) d = flush_store(s_tm, s_tn, u_kt));
      d3 = d2; d2 = d1; d1 = d;
      q2 = q1; q1 = q + d;
      if (++u_kt == nt) { u_kt = 0; s_tm = u_tm; s_tn = u_tn; if (++u_tn == ntn) { u_tn = 0; ++u_tm; } }
    }
#if TF_IGEMM_STAMP == 3
    if (wid == 4 && lane == 0) {
      unsigned long long* o = p.stamp + (size_t)blockIdx.x * 8;
      o[4] = __builtin_amdgcn_s_memtime() - a_t0; o[5] = a_wait; o[6] = a_bar; o[7] = a_iss;
      p.stamp[(size_t)(4096 + blockIdx.x) * 8 + 6] = a_fl;
    }
#endif
    return;
  }

  // =========================== consumer waves ===========================
  const int wm = wid & 1, wn = wid >> 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int fo = lr * 128 + ((lg ^ ((lr >> 1) & 7)) << 4);
  const int xo = wm * 64 * 128 + fo, wo_ = RING + wn * 64 * 128 + fo;
  // this lane's 8 bytes of a patch row: row lr, logical 16-byte chunk 2 i + (lg >> 1) (GEGLU: 2 (i >> 1) + (lg >> 1)) -> physical chunk ^ ((lr >> 1) & 7)
  const unsigned pw0 = lds0 + PATCH0 + (unsigned)wid * 2u * PATCH + (unsigned)(lr * 128 + (lg & 1) * 8);
  const unsigned pwx = (unsigned)(((lg >> 1) ^ ((lr >> 1) & 7)) << 4);
  int c_tm = tm0, c_tn = tn0;
  float ln_mean[MJ], ln_rstd[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j) { ln_mean[j] = 0.f; ln_rstd[j] = 0.f; }
  int stat_tm = -1;
  bool need_stats = false;
  f4 acc[NI][MJ];
  float ls[MJ], lq[MJ];
  h4 braw[NI];
  f4 cq[NI];
  constexpr int NP = geglu ? NI / 2 : NI;
  h4 pend[NP][MJ];                                        // the previous tile's outputs, rounded, on their way out one quarter (j) per K step (GEGLU: pend[0 / 1][j])
#pragma unroll
  for (int i = 0; i < NP; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) pend[i][j] = (h4){(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
  bool have_pend = false;
  auto tile_begin = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MJ; ++j) { ls[j] = 0.f; lq[j] = 0.f; }
    need_stats = LNF && c_tm != stat_tm;
    // bias (and LayerNorm column sums) of this lane's columns: requested now, consumed behind the K loop -- the only vector-memory instructions of these waves
    // (through the scalar cache instead -- s_load into SGPRs, v_cndmask per lane -- the tile end took 1000 cycles longer: measured, profiles/r05_ar_stamps.txt)
    const int nb = c_tn * BN + wn * 64;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      braw[i] = (h4){(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f}; cq[i] = (f4){0.f, 0.f, 0.f, 0.f};
      int n = nb + i * 16 + lg * 4;
      n = n + 3 < N_ ? n : 0;                              // columns beyond N are never stored: any readable address will do
      if (p.bias) braw[i] = *reinterpret_cast<const h4*>(p.bias + n);
      if constexpr (LNF) cq[i] = *reinterpret_cast<const f4*>(p.ln_colsum + n);
    }
  };
  // one 32-deep half (f) of a K step's fragments: the two halves are the software pipeline -- half f = 1 of step g is read while half 0 multiplies, half 0 of
  // step g + 1 while half 1 multiplies (64 fragment registers; whole steps double-buffered would be 128 next to the 64 accumulators: scratch)
  auto read_half = [&](int kt, int slot, int f, h8 (&wf)[NI], h8 (&xf)[MJ]) {
    if (TF_AR_EXP & 4) {
#pragma unroll
      for (int i = 0; i < NI; ++i) { asm volatile("" : "+v"(wf[i])); asm volatile("" : "+v"(xf[i])); }
      return;
    }
    const char* sa = smem + kt * IMG;
    const char* sw = smem + slot * IMG;
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const h8*>(sw + ((wo_ + i * 2048) ^ (f * 64)));
#pragma unroll
    for (int j = 0; j < MJ; ++j) xf[j] = *reinterpret_cast<const h8*>(sa + ((xo + j * 2048) ^ (f * 64)));
  };
  auto stats_half = [&](h8 (&xf)[MJ]) {                   // row statistics of the panel from the fragments of its first tile: this wave's own 64 rows
#pragma unroll
    for (int j = 0; j < MJ; ++j) dot2_stats<BF>(xf[j], ls[j], lq[j]);
  };
  auto mma_half = [&](h8 (&wf)[NI], h8 (&xf)[MJ]) {
    if (TF_AR_EXP & 2) {
#pragma unroll
      for (int i = 0; i < NI; ++i) { asm volatile("" ::"v"(wf[i])); asm volatile("" ::"v"(xf[i])); }
      return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[i][j] = mfma16<BF>(wf[i], xf[j], acc[i][j]);
  };
  // one half step = 16 MFMAs on one fragment half + the 8 ds_read_b128 of the other, interleaved (the reads in the shadow of the first eight MFMAs)
#define AR_INTERLEAVE() do { \
    _Pragma("unroll") for (int z_ = 0; z_ < 8; ++z_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0); } while (0)     // (reads spread over twelve MFMAs, or issued in pairs: the same times within 2 %, profiles/r05_ar_stamps.txt)
  // one quarter (16 rows) of the pending tile -> patch b
  auto drop = [&](int b, const h4 (&q)[NP]) {
    const unsigned a = pw0 + (unsigned)b * PATCH;
#pragma unroll
    for (int i = 0; i < NP; ++i) asm volatile("ds_write_b64 %0, %1" ::"v"(a + (pwx ^ (unsigned)(i * 32))), "v"(q[i]) : "memory");
  };
#define AR_QUARTER(q, j) h4 q[NP]; _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) q[i_] = pend[i_][j]
  auto drop_j = [&](int j) {                              // (static register indices: one copy per quarter)
    if (j == 0) { AR_QUARTER(q, 0); drop(0, q); }
    else if (j == 1) { AR_QUARTER(q, 1); drop(1, q); }
    else if (j == 2) { AR_QUARTER(q, 2); drop(0, q); }
    else { AR_QUARTER(q, 3); drop(1, q); }
  };
  // the end of a tile: LayerNorm fold / bias / GEGLU on the accumulators, rounded into pend
  auto tile_end = [&](bool more) {
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
          float s_ = ls[j], q_ = lq[j];
          s_ += __shfl_xor(s_, 16, 64); q_ += __shfl_xor(q_, 16, 64);
          s_ += __shfl_xor(s_, 32, 64); q_ += __shfl_xor(q_, 32, 64);
          ln_mean[j] = s_ * invK;
          ln_rstd[j] = rsqrtf(fmaxf(q_ * invK - ln_mean[j] * ln_mean[j], 0.f) + p.ln_eps);
        }
        stat_tm = c_tm;
      }
#pragma unroll
      for (int j = 0; j < MJ; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i][j] = ln_rstd[j] * (acc[i][j] - ln_mean[j] * cq[i]);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[i][j] += (f4){e2f<BF>(braw[i][0]), e2f<BF>(braw[i][1]), e2f<BF>(braw[i][2]), e2f<BF>(braw[i][3])};
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      if constexpr (geglu) {
#pragma unroll
        for (int i = 0; i < NI; i += 2) {
          h4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2e<BF>(acc[i][j][e] * gelu_f(acc[i + 1][j][e]));
          pend[i >> 1][j] = o;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          h4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2e<BF>(acc[i][j][e]);
          pend[i][j] = o;
        }
      }
    }
    have_pend = true;
    if (more) { if (++c_tn == ntn) { c_tn = 0; ++c_tm; } tile_begin(); }
  };

  h8 wfA[NI], xfA[MJ], wfB[NI], xfB[MJ];
#if TF_IGEMM_STAMP == 3
  // p.stamp[block][0..3]: whole run | until barrier P | barriers | tile ends (consumer wave 0)
  unsigned long long a_bar = 0, a_epi = 0, a_pro = 0, a_tail = 0;
  const unsigned long long a_t0 = __builtin_amdgcn_s_memtime();
#endif
  barrier();                                              // barrier P: step 0 landed
#if TF_IGEMM_STAMP == 3
  a_pro = __builtin_amdgcn_s_memtime() - a_t0;
#endif
  read_half(0, 0, 0, wfA, xfA);
  tile_begin();
  int kt = 0;
  wait_lds_reads();
  for (int g = 0; g < G; ++g) {
    if (LNF && need_stats) stats_half(xfA);
    if (have_pend && kt < 4) drop_j(kt);                  // quarter kt of the previous tile: in its patch before barrier(g), stored by the loader wave behind it
    __builtin_amdgcn_sched_barrier(0);
    read_half(kt, g & (NS - 1), 1, wfB, xfB);             // half 1 of this step, read under the MFMAs of half 0
    mma_half(wfA, xfA);
    AR_INTERLEAVE();
    __builtin_amdgcn_sched_barrier(0);
    wait_lds_reads();                                     // every fragment of step g is in registers (and the patch quarter written): the slot may be refilled
    AR_T(a_bar, barrier());                               // barrier(g): step g + 1 landed
    const int kn = kt + 1 == nt ? 0 : kt + 1;
    if (LNF && need_stats) stats_half(xfB);
    __builtin_amdgcn_sched_barrier(0);
    read_half(kn, (g + 1) & (NS - 1), 0, wfA, xfA);       // half 0 of the next step (behind the run's last step: a slot nobody writes any more; never used)
    mma_half(wfB, xfB);
    AR_INTERLEAVE();
    __builtin_amdgcn_sched_barrier(0);
    wait_lds_reads();
    if (kn == 0) AR_T(a_epi, tile_end(g + 1 < G));
    kt = kn;
  }
  // the run's last tile: no further step carries its quarters -- through patch 0 (the loader's last read was of patch 1 at most), stored from here
  {
#if TF_IGEMM_STAMP == 3
    const unsigned long long t_a = __builtin_amdgcn_s_memtime();
#endif
    const int mb = c_tm * BM + wm * 64;
    const int nbc = c_tn * BN + wn * 64;
    const int ocol0 = geglu ? (nbc >> 1) : nbc;
    const unsigned prd = lds0 + PATCH0 + (unsigned)wid * 2u * PATCH;
    auto own = [&](int j, const h4 (&q)[NP]) {
      drop(0, q);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        if (it < npass) {
          const int idx = lane + 64 * it;
          const int row = idx >> csh, c8 = idx & (cpr - 1);
          const int m = mb + j * 16 + row, no = ocol0 + c8 * 8;
          h8 v;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(prd + (unsigned)(row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4))) : "memory");
          if (m < M_ && no < No) *reinterpret_cast<h8*>(p.y + (long long)m * No + no) = v;
        }
      }
    };
    { AR_QUARTER(q, 0); own(0, q); }
    { AR_QUARTER(q, 1); own(1, q); }
    { AR_QUARTER(q, 2); own(2, q); }
    { AR_QUARTER(q, 3); own(3, q); }
#if TF_IGEMM_STAMP == 3
    a_tail = __builtin_amdgcn_s_memtime() - t_a;
#endif
  }
#if TF_IGEMM_STAMP == 3
  if (wid == 0 && lane == 0) {
    unsigned long long* o = p.stamp + (size_t)blockIdx.x * 8;
    o[0] = __builtin_amdgcn_s_memtime() - a_t0; o[1] = a_pro; o[2] = a_bar; o[3] = a_epi;
    p.stamp[(size_t)(4096 + blockIdx.x) * 8 + 7] = a_tail;
  }
#endif
}
