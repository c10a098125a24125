// Part of the implicit-GEMM family of csrc/gemm.hip (see its head comment).
#pragma once
#include "gemm_common.h"

// =====================================================================================================================
// PATCH form of the ping-pong kernel (round 4): 3 x 3 / stride 1 / pad 1 convolutions whose 192-row tile is a whole number of image rows
// (vision/conv2d.py:9-28 of the reference at BASELINE config 5's sizes: 96 / 48 / 24 pixels per image row).
//
// Why: k_igemm_pp is bound by what a CU pulls through its L1 / TA per FLOP, and for a 3 x 3 conv most of those bytes are the SAME
// pixels fetched nine times -- every filter tap stages its own shifted copy of the tile's activation rows.  Here the K loop runs channel-slab major
// (for each 64-channel slab its nine taps) and the slab's activation PATCH -- the (192 / W + 2) x (W + 2) pixels the nine taps touch, zero where
// the image ends -- is brought into LDS ONCE, double buffered; a tap's fragments are read straight out of the patch at pixel offset
// dy (W + 2) + dx.  Per K tile a block then ingests the weight tile (BN x 128 B) plus a ninth of a patch instead of weight tile + 192 x 128 B:
// 25.6 KB instead of 44.6 KB at 192 x 160 and W = 96, and 3.2 LDS-DMA instructions per wave instead of 5.5.
//
// Everything else is k_igemm_pp's one-phase form: all eight waves load and compute, two wave groups one barrier apart, a three-slot LDS-DMA ring
// (weights only) two tiles ahead, counted vmcnt waits (run-time counts through a computed jump: the tiles of a slab carry different numbers of
// loads), LDS image rows XOR-swizzled on the source side and again on the read, the shared epilogue in two passes of 96 rows.
//   LDS: patch buffer 0 | patch buffer 1 | weight slots 0 .. 2.  patch(g) lives in buffer g & 1; its pieces ride on the tiles of slab g - 1, one per wave
//   and tap (the buffer was last read for slab g - 2: behind every barrier those tiles are issued after); slab 0's patch is issued whole in the prologue.
//   Patch row pr = py (W + 2) + px holds pixel (y0 - 1 + py, px - 1) of the tile's image (y0 = its first image row); output row m of the tile
//   (image row yl = m / W, column x) reads tap (dy, dx) at patch row (yl + dy)(W + 2) + x + dx.
// Channel counts on the 64 grid (the concat pair: a slab lies in one source), no extra 1x1 segment, no split-K, fp16.
template <int BN>
__global__ void __launch_bounds__(512, 2) k_igemm_pp3(const GemmP p) {
  constexpr int BM = 192, TN = BN / 2, MJ = 3, NI = TN / 16, NS = 3;
  constexpr int WST = BN * 128;                            // bytes of a weight ring slot
  constexpr int NWG = BN / 8, WPW = (NWG + 7) / 8, WREM = NWG % 8;
  constexpr int PPW = 7;                                   // patch pieces per wave at most (56 pieces = 448 patch rows)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wm = wid & 3, wn = wid >> 2;
  const int ntiles = p.ntm * p.ntn;
  int bid = blockIdx.x;
  {
    int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;      // XCD-aware order, as in k_igemm
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_m, tile_n;
  if (p.order == 0) { tile_m = bid / p.ntn; tile_n = bid - tile_m * p.ntn; }
  else { tile_n = bid / p.ntm; tile_m = bid - tile_n * p.ntm; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int PW = p.W + 2, PR = p.pt_ppix, NPP = p.pt_ppc, PB = p.pt_stage;
  const int img = m0 / p.HoWo, y0 = fast_div(m0 - img * p.HoWo, p.dv_wo_mul, p.dv_wo_shr);
  const int G = p.C >> 6, nt = G * 9;                      // 64-channel slabs, K tiles
  const unsigned lds0 = lds_off(smem);
  const unsigned lds_w = lds0 + 2u * (unsigned)PB;

  // ---- staging geometry.  Patch piece q = wid + 8 i covers patch rows 8 q .. 8 q + 7; lane -> row 8 q + sub, source chunk cs (swizzled)
  const int sub = lane >> 3;
  const int cs = (lane & 7) ^ ((4 * (wid & 1) + (sub >> 1)) & 7);
  int pp_pix[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = wid + 8 * i, pr = 8 * q + sub;
    pp_pix[i] = -1;
    if (q < NPP && pr < PR) {
      const int py = pr / PW, px = pr - py * PW;
      const int y = y0 - 1 + py, x = px - 1;
      if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) pp_pix[i] = (img * p.H + y) * p.W + x;
    }
  }
  const i4v rs_w = raw_rsrc(p.w, p.w_bytes);
  unsigned gw[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int g = wid + 8 * i, n = n0 + 8 * g + sub;
    gw[i] = (g < NWG && n < p.N) ? (unsigned)(n * p.K) * 2u + cs * 16u : TF_OOB;
  }
  const int C1_ = p.C1, C2_ = p.C2, Cc_ = p.C;
  const unsigned long long px1 = (unsigned long long)p.x, px2 = (unsigned long long)(p.x2 ? p.x2 : p.x);
  const int nb1 = (int)p.x_bytes, nb2 = (int)p.x2_bytes;
  // pieces [i0, i1) of this wave's share of patch(g) into buffer g & 1; returns the number of loads issued
  auto stage_patch = [&](int g, int i0, int i1) -> int {
    const int c = g * 64;
    const bool second = c >= C1_;
    const unsigned long long px = second ? px2 : px1;
    i4v rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)px); rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(px >> 32) & 0xffffu));
    rs[2] = __builtin_amdgcn_readfirstlane(second ? nb2 : nb1); rs[3] = 0x00020000;
    const int ld2 = __builtin_amdgcn_readfirstlane((second ? C2_ : C1_) * 2);
    const int cb = __builtin_amdgcn_readfirstlane((second ? c - C1_ : c) * 2) + cs * 16;
    const unsigned base = lds0 + (unsigned)(g & 1) * (unsigned)PB + (unsigned)wid * 1024u;
    int n = 0;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      if (i < i0 || i >= i1 || wid + 8 * i >= NPP) continue;
      dma16(rs, pp_pix[i] >= 0 ? (unsigned)(pp_pix[i] * ld2 + cb) : TF_OOB, base + (unsigned)i * 8192u);
      ++n;
    }
    return n;
  };
  // the weight tile of (slab g, tap) into ring slot `slot`; returns the number of loads issued
  auto stage_w = [&](int slot, int g, int tap) -> int {
    const unsigned base = lds_w + (unsigned)slot * WST + (unsigned)wid * 1024u;
    const unsigned kb = (unsigned)(tap * Cc_ + g * 64) * 2u;
    int n = 0;
#pragma unroll
    for (int i = 0; i < WPW; ++i)
      if (WREM == 0 || i < WPW - 1 || wid < WREM) { dma16_w(rs_w, gw[i] != TF_OOB ? gw[i] + kb : TF_OOB, base + (unsigned)i * 8192u); ++n; }
    return n;
  };

  // ---- fragment addressing
  const int lr = lane & 15, lg = lane >> 4;
  int pp0[MJ];                                             // patch row of tap (0, 0) for row lr of pixel tile j
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int ml = wm * (BM / 4) + j * 16 + lr;
    const int yl = fast_div(ml, p.dv_wo_mul, p.dv_wo_shr);
    pp0[j] = yl * PW + (ml - yl * p.W);
  }
  const int wo_ = (wn * TN) * 128 + lr * 128 + ((lg ^ ((lr >> 1) & 7)) << 4);      // + i * 2048
  f4 acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  h8 wf[2][NI], xf[2][MJ];
  auto read_frags = [&](int buf, int slot, int tap) {
    const char* pb = smem + buf * PB;
    const char* wb = smem + 2 * PB + slot * WST;
    const int dlt = (tap / 3) * PW + (tap % 3);
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const int pr = pp0[j] + dlt;
      const int a = pr * 128 + ((lg ^ ((pr >> 1) & 7)) << 4);
      xf[0][j] = *reinterpret_cast<const h8*>(pb + a);
      xf[1][j] = *reinterpret_cast<const h8*>(pb + (a ^ 64));
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      wf[0][i] = *reinterpret_cast<const h8*>(wb + (wo_ + i * 2048));
      wf[1][i] = *reinterpret_cast<const h8*>(wb + ((wo_ + i * 2048) ^ 64));
    }
  };
  auto mma = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[f][i], xf[f][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // bias and time-embedding values of this tile's columns (as in k_igemm_pp)
  const int lb_img0 = img;
  float lb_b = 0.f, lb_c0 = 0.f, lb_c1 = 0.f;
  if (tid < BN && n0 + tid < p.N) {
    if (p.bias) lb_b = (float)p.bias[n0 + tid];
    if (p.bias_nc) {
      lb_c0 = (float)p.bias_nc[(long long)lb_img0 * p.bias_nc_stride + n0 + tid];
      if ((lb_img0 + 1) * p.HoWo < p.M) lb_c1 = (float)p.bias_nc[(long long)(lb_img0 + 1) * p.bias_nc_stride + n0 + tid];
    }
  }

  // ---- prologue: patch(0) whole, weight tiles 0 and 1; tile 0's loads landed, tile 1's in flight
  stage_patch(0, 0, PPW);
  stage_w(0, 0, 0);
  const int n1 = stage_w(1, 0, 1);                         // (nt >= 9)
  wait_vm_dyn(n1);
  barrier();                                               // P: patch(0) and weight tile 0 are visible to every wave
  if (grp == 1) barrier();                                 // the second half falls one barrier behind

  int rs = 0, ws = 2;                                      // ring slot of tile t / of tile t + 2
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int t = g * 9 + tap;
      read_frags(g & 1, rs, tap);
      int nl = 0;                                          // loads this wave issues during this tile (the newest ones: they stay in flight)
      if (t + 2 < nt) nl += tap < 7 ? stage_w(ws, g, tap + 2) : stage_w(ws, g + 1, tap - 7);
      if (g + 1 < G && tap < PPW) nl += stage_patch(g + 1, tap, tap + 1);
      const bool next = t + 1 < nt;
      if (next && grp == 1) wait_vm_dyn(nl);               // tile t + 1 (and every patch piece issued so far) landed; what this tile issued stays in flight
      wait_lds_reads();
      barrier();
      mma();
      if (next && grp == 0) wait_vm_dyn(nl);
      barrier();
      if (++rs == NS) rs = 0;
      if (++ws == NS) ws = 0;
    }
  }
  if (grp == 0) barrier();                                 // the first half waits for the second: every wave is done with the ring and the patches

  // ---- epilogue: two passes of 96 rows through the shared scratch (as k_igemm_pp)
  f4 csum[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) csum[i] = (f4){0.f, 0.f, 0.f, 0.f};
  constexpr int BS = BM / 2;
  float* const lbt = reinterpret_cast<float*>(smem + 4 * (BS / 2) * (TN + 4) * 4 + BS * 8 + 4 * BN * 8);
  if (tid < BN) { lbt[tid] = lb_b; lbt[BN + tid] = lb_c0; lbt[2 * BN + tid] = lb_c1; }
  const int lb_m1 = (lb_img0 + 1) * p.HoWo;
#pragma unroll
  for (int sm = 0; sm < 2; ++sm) {
    if ((wm >> 1) == sm) igemm_scratch_write<BS, BN>(p, acc, csum, smem, (wm & 1) | (wn << 1), lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
    igemm_epilogue<BS, BN, 0, false, 2, true>(p, smem, m0 + sm * BS, n0, 0, wid & 3, wid >> 2, lane, lbt, n0, lb_m1);
    if (p.gn_part && m0 + sm * BS < p.M) igemm_gn_stats<BS, BN>(p, smem, m0 + sm * BS, n0, wid & 3, wid >> 2, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
  }
}
