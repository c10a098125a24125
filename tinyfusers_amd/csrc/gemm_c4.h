// Part of the implicit-GEMM family of csrc/gemm.hip (see its head comment); split into translation units so that the
// instances compile in parallel.
#pragma once
#include "gemm_common.h"

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
#ifndef TF_C4_TWO
#define TF_C4_TWO 0      // 1 (tagged build -DTF_C4_TWO=1): two K tiles in flight on the two-slot ring through a second barrier behind the fragment reads.  MEASURED: no change
                         // on any shape (profiles/r05_c4_two.txt: 182.0 vs 181.4 us on 73728 x 2560 x 320 ...) -- the K step is not waiting for ONE tile's latency but for the
                         // wave's own LDS-DMA issue stream (8 pieces per wave and K tile at 100-185 cycles each); off
#endif
template <bool LNF, bool BF = false>   // BF: bfloat16 operands / outputs (gemm_k_c4_bf16.hip)
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
    for (int i = 0; i < 4; ++i) dma16_w(rs_w, gw[i] != TF_OOB ? gw[i] + (unsigned)kt * 128u : TF_OOB, base + 16384u + (unsigned)i * 4096u);
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
    // ---- K loop: tile t in slot t & 1.  Round-5 experiment (TF_C4_TWO = 1, off): TWO K tiles in flight on the two-slot ring -- a wave holds a whole K tile's
    // fragments in registers while it multiplies, so slot t & 1 is free as soon as EVERY wave has read tile t; a second barrier behind the fragment reads says
    // so and tile t + 2 is issued there, in front of tile t's MFMAs, instead of one step later.  Parity-green, and no faster (see TF_C4_TWO above).
    for (int t = 0; t < nt; ++t) {
      // K tile t has landed.  For t = 0 it was issued in front of the previous tile's epilogue: where that epilogue's vector-memory
      // instructions are known to be `pend` stores, followed by this tile's bias / column-sum loads and nothing else, those `young`
      // ones stay in flight (the counter retires in issue order)
      if (t == 0) {
        if (young == 4) wait_vm<4>();
        else if (young == 8) wait_vm<8>();
        else if (young == 12) wait_vm<12>();
        else if (young == 16) wait_vm<16>();
        else wait_vm<0>();
      }
#if TF_C4_TWO
      else if (t + 1 < nt) wait_vm<8>();                   // tile t landed; tile t + 1 (issued one step ago, 8 pieces per wave) stays in flight
#endif
      else wait_vm<0>();
      barrier();
#if TF_C4_TWO
      if (t == 0 && nt > 1) stage(1, 1);                   // (slot 1 held the previous epilogue's patches: free behind this barrier)
#else
      if (t + 1 < nt) stage((t + 1) & 1, t + 1);
#endif
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
#if TF_C4_TWO
      if (t + 2 < nt) {
        barrier();                                         // every wave holds tile t in registers: its slot takes tile t + 2 now
        stage(t & 1, t + 2);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      if (LNF && need_stats) {
        // row statistics from the fragments: the two waves that share these 64 rows (wn = 0, 1) take one 32-deep k-step each
        auto acc_stats = [&](const h8 (&x)[MJ]) {
#pragma unroll
          for (int j = 0; j < MJ; ++j) dot2_stats<BF>(x[j], ls[j], lq[j]);
        };
        if (wn == 0) acc_stats(xf[0]); else acc_stats(xf[1]);
      }
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j) acc[i][j] = mfma16<BF>(wf[f][i], xf[f][j], acc[i][j]);
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
      for (int j = 0; j < MJ; ++j) acc[i][j] += (f4){e2f<BF>(braw[i][0]), e2f<BF>(braw[i][1]), e2f<BF>(braw[i][2]), e2f<BF>(braw[i][3])};
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
            for (int e = 0; e < 4; ++e) o[e] = f2e<BF>(acc[i][j][e] * gelu_f(acc[i + 1][j][e]));
            asm volatile("ds_write_b64 %0, %1" ::"v"(rowa + (unsigned)((i >> 1) * 32 + lg * 8)), "v"(o) : "memory");
          }
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            h4 o;
            for (int e = 0; e < 4; ++e) o[e] = f2e<BF>(acc[i][j][e]);
            asm volatile("ds_write_b64 %0, %1" ::"v"(rowa + (unsigned)(i * 32 + lg * 8)), "v"(o) : "memory");
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // read back as rows: 32 rows x (ocols / 8) 16-byte chunks
      const int csh = geglu ? 2 : 3, cpr = 1 << csh;       // 8 or 4 chunks per row (a shift, not a divide: the runtime quotient cost ~40 VALU instructions per use)
      // (round 5) the residual rows of this half are requested up front -- as one loop the residual load of every iteration sat behind the previous
      // iteration's store and in front of its own use: four serial global round trips per half
      h8 rres[4];
      if (!LNF && p.residual) {                            // (no LayerNorm-folded launch of the step carries a residual: that instance keeps its registers)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int idx = lane + 64 * it;
          const int row = idx >> csh, c8 = idx & (cpr - 1);
          const int m = cm0 + wm * 64 + h * 32 + row, no = ocol0 + c8 * 8;
          if (idx < 32 * cpr && m < M_ && no < No) rres[it] = *reinterpret_cast<const h8*>(p.residual + (long long)m * No + no);
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = lane + 64 * it;
        if (idx >= 32 * cpr) break;                        // (GEGLU: two iterations)
        const int row = idx >> csh, c8 = idx & (cpr - 1);
        const int m = cm0 + wm * 64 + h * 32 + row, no = ocol0 + c8 * 8;
        h8 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(pa + (unsigned)row * 144u + (unsigned)c8 * 16u) : "memory");
        if (m < M_ && no < No) {
          const long long o = (long long)m * No + no;
          if (p.residual) { const h8 r = LNF ? *reinterpret_cast<const h8*>(p.residual + o) : rres[it]; for (int e = 0; e < 8; ++e) v[e] = f2e<BF>(e2f<BF>(v[e]) + e2f<BF>(r[e])); }
          *reinterpret_cast<h8*>(p.y + o) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    tile = next;
  }
}
