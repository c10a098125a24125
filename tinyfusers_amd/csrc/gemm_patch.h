// Part of the implicit-GEMM family of csrc/gemm.hip (see its head comment); split into translation units so that the
// instances compile in parallel.
#pragma once
#include "gemm_common.h"

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
template <int BM, int BN, bool GI = false, bool BF = false>   // BF: bfloat16 operands / outputs (gemm_k_patch_bf16.hip)
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
      lds_write16(a, gi_apply<BF>(lds_read16(a), na, nb, p.gi_silu, v >= 0));
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
        bload_lds16_w(rs_w, off, base + (w4 + 4 * i) * 1024);
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
      // TF_IGEMM_PRE stages in front of barrier P, the rest of the ring right behind it (gemm_common.h: 99 = the whole ring, shipped; 2 was measured slower in the step)
      int s_ = 0;
      for (; s_ < TF_IGEMM_PRE && s_ < NS && s_ < nt; ++s_) { int n = stage(s_); if (s_ > 0) W += n; }
      if (gi_on) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }   // barriers A, B of gi_prologue (consumer waves)
      wait_vm_dyn(W);                                      // tile 0 (and everything issued before it) landed
      if (gi_on && pro_g >= 0) {
        gi_range(pro_g, 0, TF_PATCH_PPW);
        if (pro_next > 0) gi_range(pro_g + 1, 0, pro_next);
      }
      __builtin_amdgcn_s_barrier();                       // barrier P
      asm volatile("" ::: "memory");
      for (; s_ < NS && s_ < nt; ++s_) W += stage(s_);
    }
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
    EpiPre<BM, BN> epre;
    igemm_epilogue_prefetch<BM, BN>(p, m0, n0, w4, 1, lane, epre);     // the epilogue's first loads, requested now (no counted vmcnt wait follows in this wave)
    __builtin_amdgcn_s_barrier();                         // barrier X
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // barrier Y: the consumers' tiles are in the LDS scratch
    asm volatile("" ::: "memory");
    igemm_epilogue<BM, BN, 0, BF, 4, false, true>(p, smem, m0, n0, split, w4, 1, lane, nullptr, 0, 0, epre);
    if (p.gn_part) igemm_gn_stats<BM, BN>(p, smem, m0, n0, w4, 1, lane);
    return;
  }

  // ================================= CONSUMER WAVES ===============================================
  if constexpr (GI) gi_prologue<BF>(p, smem, m0 / p.HoWo, w4 * 64 + lane);
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
        for (int j = 0; j < MJ; ++j) acc[i][j] = mfma16<BF>(wf[k2][i], xf[k2][j], acc[i][j]);
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
  EpiPre<BM, BN> epre;
  igemm_epilogue_prefetch<BM, BN>(p, m0, n0, w4, 0, lane, epre);       // the K loop is over: the epilogue's loads fly under barriers X / Y and the scratch write
  __builtin_amdgcn_s_barrier();                           // barrier X: every consumer is done with the ring and the patches
  asm volatile("" ::: "memory");
  igemm_scratch_write<BM, BN>(p, acc, csum, smem, w4, lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                           // barrier Y
  asm volatile("" ::: "memory");
  igemm_epilogue<BM, BN, 0, BF, 4, false, true>(p, smem, m0, n0, split, w4, 0, lane, nullptr, 0, 0, epre);
  if (p.gn_part) igemm_gn_stats<BM, BN>(p, smem, m0, n0, w4, 0, lane);
}
