// Part of the implicit-GEMM family of csrc/gemm.hip (see its head comment); split into translation units so that the
// instances compile in parallel.
#pragma once
#include "gemm_common.h"

// =====================================================================================================================
// k_gemm_c8 (round 5): the WIDE form of the persistent short-K kernel k_gemm_c4 for launches that fill the chip with 256-row tiles --
// BASELINE config 5's linears and 1 x 1 convolutions with K <= 6400 (ff/linear.py:112-121, ff/nn.py:5-23, attention/attention.py:35-41 of the
// reference): q|k|v, to_out, the GEGLU projection, FF2, proj_in / proj_out.  Those were the largest block of config 5's GEMM time (7.05 of 15.1 ms at
// 0.35-0.88 PFLOP/s, profiles/r04_gemm_shapes_images4_latent96_fp16.txt) on two kernels that each lacked half of what the shapes need: k_gemm_c4 has the
// register epilogue and the cross-tile prefetch but 128 x 128 tiles (64 FLOP per ingested byte) and ONE 32-KiB K tile in flight per block; the ping-pong
// kernel has the 256-row tile but a shared-scratch epilogue with two block barriers per half and no overlap across tiles.  This kernel is both:
//   * ONE block of 8 waves per CU for the whole launch, walking its list of 256 x 128 tiles (4 x 2 wave tiles of 64 x 64: k_gemm_c4's fragment code);
//     85 FLOP per ingested byte, a third more than the 128 x 128 tile;
//   * a THREE-slot LDS-DMA ring of 48-KiB K tiles that runs straight across tile boundaries (slot = running K-tile index mod 3): two K tiles in
//     flight at any time, and the first TWO K tiles of the next output tile are issued before this tile's epilogue starts;
//   * k_gemm_c4's epilogue: LayerNorm fold / bias / GEGLU on the accumulators in registers, transposed half a wave tile at a time through a private
//     per-wave LDS patch -- in the ring slot the next two K tiles do not land in -- residual added on the row side, 16-byte stores that nothing waits
//     for (counted vmcnt in front of the next tile's K tiles).
// Eligibility = k_gemm_c4's (c4_ok: 1 x 1 / stride 1, channel counts on the 64 grid, no split-K / statistics / time embedding).
template <bool LNF, bool BF = false>
__global__ void __launch_bounds__(512, 2) k_gemm_c8(const GemmP p) {
  constexpr int BM = 256, BN = 128, MJ = 4, NI = 4;
  constexpr int STAGE = (BM + BN) * 128;                  // 48 KiB
  constexpr int PATCH = 32 * 144;                         // per-wave transpose patch: 32 rows x (128 + 16) bytes
  constexpr int LPT = 6;                                   // LDS-DMA pieces per wave and K tile: 4 activation + 2 weight
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid & 3, wn = wid >> 2;                   // 4 (rows) x 2 (columns) wave tiles of 64 x 64
  const int lr = lane & 15, lg = lane >> 4;
  const int sub = lane >> 3;
  const int cs = (lane & 7) ^ ((4 * (wid & 1) + (sub >> 1)) & 7);       // source chunk: pieces of a wave are 8 apart, so 8 g's parity is the wave's
  const unsigned lds0 = lds_off(smem);
  const int ntm = p.ntm, ntn = p.ntn, ntiles = ntm * ntn;
  const int nt = p.ktiles;
  const int gstep = gridDim.x;
  const int C1_ = p.C1, K_ = p.K, M_ = p.M, N_ = p.N;
  const i4v rs_x1 = raw_rsrc(p.x, p.x_bytes), rs_x2 = raw_rsrc(p.x2 ? p.x2 : p.x, p.x2_bytes), rs_w = raw_rsrc(p.w, p.w_bytes);
  const int C2_ = p.C2;
  const int fo = lr * 128 + ((lg ^ ((lr >> 1) & 7)) << 4);
  const int xo = wm * 64 * 128 + fo, wo_ = (BM + wn * 64) * 128 + fo;
  f2* const stats = reinterpret_cast<f2*>(smem + 3 * STAGE);   // [8 waves][64 rows] halves of the LayerNorm row sums (behind the ring)

  // this wave's staging rows of a tile: activation pieces wid + 8 i (i < 4: rows 8 (wid + 8 i) + sub), weight pieces wid + 8 i (i < 2)
  int am[4];
  unsigned gw[2];
  auto setup = [&](int tile, int& m0, int& n0) {
    int tm, tn;
    if (p.order == 0) { tm = tile / ntn; tn = tile - tm * ntn; } else { tn = tile / ntm; tm = tile - tn * ntm; }
    m0 = tm * BM; n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 8 * (wid + 8 * i) + sub;
      am[i] = m < M_ ? m : -1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = n0 + 8 * (wid + 8 * i) + sub;
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
    for (int i = 0; i < 4; ++i) dma16(rs, am[i] >= 0 ? (unsigned)(am[i] * ld + cc) * 2u : TF_OOB, base + (unsigned)i * 8192u);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16_w(rs_w, gw[i] != TF_OOB ? gw[i] + (unsigned)kt * 128u : TF_OOB, base + (unsigned)(BM * 128) + (unsigned)i * 8192u);
  };
  auto barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

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
  int s0 = 0;                                             // ring slot of this tile's K tile 0 (the ring runs across tiles: slot = (s0 + t) % 3)
  stage(0, 0);
  if (nt > 1) stage(1, 1);
  float ln_mean[MJ], ln_rstd[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j) { ln_mean[j] = 0.f; ln_rstd[j] = 0.f; }
  int stat_m0 = -1;
  int pend = -1;                                          // vector-memory instructions of the previous epilogue behind this tile's prefetched K tiles (-1: unknown -> full wait)
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
    // bias (and LayerNorm column sums) of this lane's columns: requested now, consumed behind the K loop (as in k_gemm_c4: the values stay raw until
    // then, so that the compiler's own wait for them does not land in front of the K loop)
    const int nb = n0 + wn * 64;                           // first (packed) column of the wave tile
    h4 braw[NI];
    f4 cq[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      braw[i] = (h4){(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f}; cq[i] = (f4){0.f, 0.f, 0.f, 0.f};
      int n = nb + i * 16 + lg * 4;
      n = n + 3 < N_ ? n : 0;                              // columns beyond N are never stored: any readable address will do (no masked load)
      if (p.bias) braw[i] = *reinterpret_cast<const h4*>(p.bias + n);
      if constexpr (LNF) cq[i] = *reinterpret_cast<const f4*>(p.ln_colsum + n);
    }
    // what is younger than this tile's K tile 0 in the wave's vector-memory queue: K tile 1 (LPT, if any), the previous epilogue's stores (pend), these loads
    const int nload = (p.bias ? NI : 0) + (LNF ? NI : 0);
    const int young0 = pend >= 0 ? (nt > 1 ? LPT : 0) + pend + nload : -1;
    // ---- K loop: K tile t in slot (s0 + t) % 3; at the top of step t tile t must have landed (tile t + 1 may be in flight); the barrier then frees
    // the slot of tile t - 1 for tile t + 2
    int sl = s0;                                           // slot of K tile t
    for (int t = 0; t < nt; ++t) {
      if (t == 0) {
        if (young0 == 6) wait_vm<6>(); else if (young0 == 10) wait_vm<10>(); else if (young0 == 14) wait_vm<14>(); else if (young0 == 18) wait_vm<18>();
        else if (young0 == 22) wait_vm<22>(); else if (young0 == 4) wait_vm<4>(); else if (young0 == 8) wait_vm<8>(); else if (young0 == 12) wait_vm<12>();
        else if (young0 == 16) wait_vm<16>(); else wait_vm<0>();
      } else if (t == 1) {
        // K tile 1 was issued in front of the previous epilogue too: younger than it are that epilogue's stores, this tile's bias loads and K tile 2
        const int y1 = pend >= 0 ? pend + nload + (nt > 2 ? LPT : 0) : -1;
        if (y1 == 6) wait_vm<6>(); else if (y1 == 10) wait_vm<10>(); else if (y1 == 14) wait_vm<14>(); else if (y1 == 18) wait_vm<18>(); else if (y1 == 22) wait_vm<22>();
        else if (y1 == 4) wait_vm<4>(); else if (y1 == 8) wait_vm<8>(); else if (y1 == 12) wait_vm<12>(); else if (y1 == 16) wait_vm<16>(); else wait_vm<0>();
      } else {
        if (t + 1 < nt) wait_vm<LPT>(); else wait_vm<0>();   // tile t landed, tile t + 1 in flight (everything else in the queue is older than tile t)
      }
      barrier();
      if (t + 2 < nt) { int s2 = sl + 2; s2 = s2 >= 3 ? s2 - 3 : s2; stage(s2, t + 2); }
      const char* sb = smem + sl * STAGE;
      sl = sl == 2 ? 0 : sl + 1;
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
    barrier();                                             // every wave is done with the ring (all three slots are free)
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
          const f2 o_ = stats[(wid ^ 4) * 64 + j * 16 + lr];       // the partner along n: same rows, the other k-step
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
    // ---- the next tile's rows and its first TWO K tiles, in flight during this tile's epilogue: they land in slots s1, s1 + 1 (s1 = the slot behind this
    // tile's last K tile); the epilogue's patches live in the third slot -- the one this tile's LAST K tile occupied, free since the barrier above
    const int cm0 = m0, cn0 = n0;
    int s1 = s0 + nt % 3; s1 = s1 >= 3 ? s1 - 3 : s1;
    int sp = s1 + 2; sp = sp >= 3 ? sp - 3 : sp;            // = slot of K tile nt - 1
    const int next = next_tile(cq_, ce_);
    if (next >= 0) {
      setup(next, m0, n0);
      stage(s1, 0);
      if (nt > 1) { int s11 = s1 + 1; s11 = s11 >= 3 ? s11 - 3 : s11; stage(s11, 1); }
    }
    s0 = s1;
    const bool geglu = p.act == 1;
    const int No = geglu ? N_ >> 1 : N_;
    // an interior tile without a residual stores 2 halves x 32 rows x cpr chunks / 64 lanes = 8 (GEGLU: 4) times per wave, every lane active
    pend = (cm0 + BM <= M_ && cn0 + BN <= N_ && !p.residual) ? (geglu ? 4 : 8) : -1;
    char* const patch = smem + sp * STAGE + wid * PATCH;
    const unsigned pa = lds_off(patch);
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
