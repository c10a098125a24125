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
// dy PW + dx (PW = W + 4: the patch's row pitch).  Per K tile a block then ingests the weight tile (BN x 128 B) plus a ninth of a patch instead of weight tile + 192 x 128 B:
// 25.6 KB instead of 44.6 KB at 192 x 160 and W = 96, and 3.2 LDS-DMA instructions per wave instead of 5.5.
//
// Everything else is k_igemm_pp's one-phase form: all eight waves load and compute, two wave groups one barrier apart, an LDS-DMA ring of three
// slots (weights only) two tiles ahead -- nine taps = three turns of the ring, so a tap's slot is a compile-time constant; a fourth slot where the
// patches leave room for it measured 3-5 % SLOWER (profiles/r04_pp3.txt): latency is not what this loop waits for -- counted vmcnt waits (run-time counts through a computed jump: the tiles of a slab carry different numbers
// of loads), LDS image rows XOR-swizzled on the source side and again on the read, the shared epilogue in two passes of 96 rows.
//   LDS: patch buffer 0 | patch buffer 1 | weight slots 0 .. 2.  patch(g) lives in buffer g & 1; its pieces ride on the first tiles of slab g - 1, one
//   per wave and tap (the buffer was last read for slab g - 2: behind every barrier those tiles are issued after); slab 0's patch is issued whole in the
//   prologue.  Patch row pr = py PW + px holds pixel (y0 - 1 + py, px - 1) of the tile's image (y0 = its first image row; px > W + 1 is padding);
//   output row m of the tile (image row yl = m / W, column x) reads tap (dy, dx) at patch row (yl + dy) PW + x + dx.
// Channel counts on the 64 grid (the concat pair: a slab lies in one source), no split-K.  The folded 1x1 skip projection of a ResBlock (extra K
// columns behind the nine taps, their activations the output pixels of x3 | x4: vision/resnet.py:23-31 of the reference) follows as one K tile per
// 64-channel slab in k_igemm_pp's form -- 192 activation rows per tile through a three-slot ring of its own, laid over the patch buffers (the one the
// last slab does not read holds slots 0 and 1 -- slot 1 in the spare LDS behind the weight ring where a buffer is too small for two -- the other one slot 2).  W -- the OUTPUT row length -- is a
// template parameter: every patch offset is then an instruction immediate.  The nearest-2x up-sampling of the Upsample convs (vision/unet.py:79-86 of the
// reference) folds into the patch gather: patch pixel (y, x) comes from source pixel (y >> 1, x >> 1), fetched once per slab instead of once per tap and 2 x 2 copy.
// The patch image's swizzle: a tap reads 16 consecutive patch rows starting ANYWHERE, so the XOR term of the aligned tiles ((row >> 1) & 7) would put two
// rows of a ds_read_b128 lane group on the same banks for every shift but 0.  chunk ^ (row & 6) is conflict-free for every start row: the hardware
// serves lanes {0-3, 12-15} of one k-chunk together with lanes {4-11} of the next chunk, i.e. per row parity four rows of one cyclic window of (row >> 1)
// with chunk c and the other four with chunk c ^ 1, and 2 ((row >> 1) & 3) separates them for all eight windows (brute force over all 8^8 maps).
// With PW = 4 mod 8 the swizzle term of row pr + PW is that of pr with bit 2 flipped -- the other 64-byte half of the row -- and that of pr + 2 PW
// is the same: the addresses of tap (dy, dx) are those of tap (0, dx) plus the immediate dy PW 128, with the two k-steps' registers swapped for dy = 1.
// So a wave keeps 3 (dx) x 3 (pixel tiles) x 2 (k-steps) fragment addresses and the K loop has no address arithmetic besides the buffer flip per slab.
#ifndef TF_PP3_STAMP
#define TF_PP3_STAMP 0    // diagnostic builds (tools/pp3_stamp.py), stamps into the workspace: 1 = s_memtime of waves 0 and 4 of block 0 around the phases of the tiles of slab 1 + the clock pair; 2 = only the (s_memtime, s_memrealtime) pair around block 0's K loop
#endif
// (Issuing part of a tile's loads between its MFMAs instead of in front of the barrier -- balancing the two phases of the ping-pong -- measured
// 0 ... -5 %, the more the more loads moved: profiles/r04_pp3.txt.  The loop runs at 1.83-1.94 GHz with the matrix pipe busy in ~72 % of its
// cycles: the chip holds its clock down under this load, and what raises throughput from here is less energy per MFMA, not a tighter issue stream.)
// F8 = OCP e4m3 operands on the block-scaled MFMA, as k_igemm_pp<F8>: a slab is 128 channels (the 128 bytes of a patch row), the fragment
//   reads and their addresses are the fp16 kernel's, one v_mfma_scale_f32_16x16x128_f8f6f4 takes both 64-byte halves.  The E8M0 bytes of a patch row's
//   four 32-channel blocks travel as a second, small patch (one dword per patch row, double buffered behind the weight ring): waves 0 .. 6 fetch it with
//   one buffer_load_dword ... lds each, on tap 7 of the previous slab (the patch pieces ride on taps 0 .. PPW - 1 <= 6), and a lane reads the byte of (its
//   patch row, block lg) with one ds_read_u8 per pixel tile at an immediate offset per tap.  Channel counts on the 128 grid; weights carry one fp32 scale
//   per output channel, applied to the accumulators in front of the epilogue.
// H2 (F8 only) = channel counts on the 64 grid (320, 960: one source tensor): the last slab holds 64 channels -- the lanes of its upper 64 bytes fetch out of range
//   (zero codes; the weight bytes they meet belong to the next tap or row, finite e4m3 values, and contribute nothing) -- and a patch row's four scale bytes sit at
//   an offset that is only 2-byte aligned (C / 32 = 10 bytes per pixel), so they travel as two buffer_load_ushort ... lds into two tables (blocks 0-1 | blocks 2-3; the
//   second one out of range for the half slab: E8M0 0 instead of a neighbour's byte, which could be the NaN code).
template <int BN, int W, bool F8 = false, bool H2 = false, bool BF = false>   // BF: bfloat16 operands / outputs (gemm_k_pp3_bf16.hip)
__global__ void __launch_bounds__(512, 2) k_igemm_pp3(const GemmP p) {
  static_assert(!(BF && F8), "the e4m3 form has fp16 bias / residual / outputs");
  static_assert(F8 || !H2, "the half-slab form is the e4m3 kernel's");
  constexpr int BM = 192, TN = BN / 2, MJ = 3, NI = TN / 16, NS = 3, D = NS - 1;
  constexpr int ES = F8 ? 1 : 2, SLAB = 128 / ES;          // bytes per element; channels of a slab (128 bytes of a patch row)
  constexpr int WST = BN * 128;                            // bytes of a weight ring slot
  constexpr int NWG = BN / 8, WPW = (NWG + 7) / 8, WREM = NWG % 8;
  constexpr int PW = W + 4, PROWS = (BM / W + 2) * PW, NPP = (PROWS + 7) / 8, PB = NPP * 1024;     // patch: row pitch, rows, 8-row pieces, bytes of a buffer
  constexpr int PPW = (NPP + 7) / 8;                       // patch pieces per wave at most: they ride on taps 0 .. PPW - 1 and are waited for D - 1 tiles later
  static_assert(W % 8 == 0 && BM % W == 0, "the tile is whole image rows; PW = 4 mod 8");
  static_assert(PPW - 1 + D - 1 <= 8, "a slab's patch must have landed when its first tile is read");
  constexpr int NSW = (PROWS + 63) / 64;                   // waves that fetch scales
  constexpr int SCT = F8 ? NSW * 256 : 0;                  // bytes of a scale table: one dword per patch row, whole 64-row wave loads
  constexpr int SCB = (H2 ? 2 : 1) * SCT;                  // bytes of a scale patch (H2: two tables)
  static_assert(2 * PB + NS * WST + 2 * SCB <= 163840, "LDS budget");
  static_assert(!F8 || (PPW <= 7 && NSW <= 8), "the scale loads ride on tap 7");
  constexpr int AST = BM * 128;                            // bytes of an activation tile of the extra 1x1 segment
  constexpr int XS1 = 2 * AST <= PB ? AST : -1;            // its slot 1: behind slot 0 in the free patch buffer, or (-1) in the spare LDS behind the weight ring
  static_assert(F8 || XS1 > 0 || 2 * PB + NS * WST + AST <= 163840, "LDS budget of the extra segment");
  static_assert(PPW <= 7, "taps 7 and 8 of the last slab carry the first two extra activation tiles");
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
  const int img = m0 / p.HoWo, y0 = (m0 - img * p.HoWo) / W;
  const int G = (p.C + SLAB - 1) / SLAB, nt = G * 9;       // channel slabs (H2: the last one half full), K tiles of the nine taps
  const bool half_last = H2 && (p.C & 64);
  const int GE = F8 ? 0 : (p.C3 + p.C4) >> 6;              // K tiles of the extra 1x1 segment
  const unsigned lds0 = lds_off(smem);
  const unsigned lds_w = lds0 + 2u * (unsigned)PB;

  const int ups = p.ups;                                   // nearest-2x up-sampling folded into the gather: the patch holds the UP-SAMPLED pixels (p.H x p.W is the source)
  // ---- staging geometry.  Patch piece q = wid + 8 i covers patch rows 8 q .. 8 q + 7; lane -> row 8 q + sub, source chunk swizzled
  const int sub = lane >> 3;
  const int cs = (lane & 7) ^ ((4 * (wid & 1) + (sub >> 1)) & 7);      // weight rows: the aligned tiles' swizzle
  const int csp = (lane & 7) ^ (sub & 6);                                // patch rows
  int pp_pix[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pr = 8 * (wid + 8 * i) + sub;
    const int py = pr / PW, px = pr - py * PW;
    const int y = y0 - 1 + py, x = px - 1;
    pp_pix[i] = (pr < PROWS && (unsigned)y < (unsigned)p.Ho && (unsigned)x < (unsigned)W) ? (img * p.H + (y >> ups)) * p.W + (x >> ups) : -1;
  }
  const i4v rs_w = raw_rsrc(p.w, p.w_bytes);
  unsigned gw[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int g = wid + 8 * i, n = n0 + 8 * g + sub;
    gw[i] = (g < NWG && n < p.N) ? (unsigned)(n * p.K) * ES + cs * 16u : TF_OOB;
  }
  const int C1_ = p.C1, C2_ = p.C2, Cc_ = p.C;
  const unsigned long long px1 = (unsigned long long)p.x, px2 = (unsigned long long)(p.x2 ? p.x2 : p.x);
  const int nb1 = (int)p.x_bytes, nb2 = (int)p.x2_bytes;
  // piece `i` of this wave's share of patch(g) into buffer g & 1; returns the number of loads issued
  auto stage_patch = [&](int g, int i) -> int {
    if (wid + 8 * i >= NPP) return 0;
    const int c = g * SLAB;
    const bool second = c >= C1_;
    const unsigned long long px = second ? px2 : px1;
    i4v rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)px); rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(px >> 32) & 0xffffu));
    rs[2] = __builtin_amdgcn_readfirstlane(second ? nb2 : nb1); rs[3] = 0x00020000;
    const int ld2 = __builtin_amdgcn_readfirstlane((second ? C2_ : C1_) * ES);
    const int cb = __builtin_amdgcn_readfirstlane((second ? c - C1_ : c) * ES) + csp * 16;
    const bool dead = H2 && half_last && g == G - 1 && csp >= 4;       // the upper 64 bytes of a half slab
    dma16(rs, (pp_pix[i] >= 0 && !dead) ? (unsigned)(pp_pix[i] * ld2 + cb) : TF_OOB, lds0 + (unsigned)(g & 1) * (unsigned)PB + (unsigned)wid * 1024u + (unsigned)i * 8192u);
    return 1;
  };
  // F8: the scale dwords of patch(g): lane L of wave w < NSW fetches those of patch row 64 w + L.  A source tensor holds its E8M0 bytes behind its
  // codes (offset = the codes' byte count), C / 32 per pixel; the slab starting at channel c has its four at c / 32.
  int sc_pix = -1;
  if constexpr (F8) {
    const int pr = 64 * wid + lane;
    const int py = pr / PW, px = pr - py * PW;
    const int y = y0 - 1 + py, x = px - 1;
    if (pr < PROWS && (unsigned)y < (unsigned)p.Ho && (unsigned)x < (unsigned)W) sc_pix = (img * p.H + (y >> ups)) * p.W + (x >> ups);
  }
  auto stage_sc = [&](int g) -> int {
    if constexpr (!F8) return 0;
    if (wid >= NSW) return 0;
    const int c = g * SLAB;
    const bool second = c >= C1_;
    const unsigned long long px = second ? px2 : px1;
    const int s_nb = __builtin_amdgcn_readfirstlane(second ? nb2 : nb1);
    i4v rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)px); rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(px >> 32) & 0xffffu));
    rs[2] = s_nb + (s_nb >> 5); rs[3] = 0x00020000;
    const int ld32 = __builtin_amdgcn_readfirstlane((second ? C2_ : C1_) >> 5), c32 = __builtin_amdgcn_readfirstlane((second ? c - C1_ : c) >> 5);
    const unsigned off = sc_pix >= 0 ? (unsigned)s_nb + (unsigned)(sc_pix * ld32 + c32) : TF_OOB;
    const unsigned ldsb = lds_w + (unsigned)NS * WST + (unsigned)(g & 1) * SCB + (unsigned)wid * 256u;
    if constexpr (H2) {
      const unsigned off_hi = (sc_pix >= 0 && !(half_last && g == G - 1)) ? off + 2u : TF_OOB;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ushort %1, %2, 0 offen lds" :: "s"(__builtin_amdgcn_readfirstlane((int)ldsb)), "v"(off), "s"(rs) : "memory");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ushort %1, %2, 0 offen lds" :: "s"(__builtin_amdgcn_readfirstlane((int)(ldsb + SCT))), "v"(off_hi), "s"(rs) : "memory");
      return 2;
    }
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" :: "s"(__builtin_amdgcn_readfirstlane((int)ldsb)), "v"(off), "s"(rs) : "memory");
    return 1;
  };
  // the weight tile at byte offset kb of every row into ring slot `slot`; returns the number of loads issued
  auto stage_wk = [&](int slot, unsigned kb) -> int {
    const unsigned base = lds_w + (unsigned)slot * WST + (unsigned)wid * 1024u;
    int n = 0;
#pragma unroll
    for (int i = 0; i < WPW; ++i)
      if (WREM == 0 || i < WPW - 1 || wid < WREM) { dma16_w(rs_w, gw[i] != TF_OOB ? gw[i] + kb : TF_OOB, base + (unsigned)i * 8192u); ++n; }
    return n;
  };
  auto stage_w = [&](int slot, int g, int tap) -> int { return stage_wk(slot, (unsigned)(tap * Cc_ + g * SLAB) * ES); };      // (slab g, tap)
  // extra segment: LDS offset of activation slot e % 3, and the activation tile of its 64-channel slab e (3 pieces per wave)
  const int xb_free = (G & 1) * PB;                        // the patch buffer the last slab does not read
  auto xslot = [&](int e) -> int { const int r = e % 3; return r == 0 ? xb_free : r == 1 ? (XS1 > 0 ? xb_free + AST : 2 * PB + NS * WST) : PB - xb_free; };
  auto stage_x = [&](int e) -> int {
    const int c = e * 64, C3_ = p.C3;
    const bool second = c >= C3_;
    const unsigned long long px = (unsigned long long)(second ? p.x4 : p.x3);
    i4v rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)px); rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(px >> 32) & 0xffffu));
    rs[2] = __builtin_amdgcn_readfirstlane((int)(second ? p.x4_bytes : p.x3_bytes)); rs[3] = 0x00020000;
    const int ld2 = __builtin_amdgcn_readfirstlane((second ? p.C4 : C3_) * 2);
    const int cb = __builtin_amdgcn_readfirstlane((second ? c - C3_ : c) * 2) + csp * 16;       // (rows 8 q + sub: the patch rows' swizzle)
    const unsigned base = lds0 + (unsigned)xslot(e) + (unsigned)wid * 1024u;
#pragma unroll
    for (int i = 0; i < BM / 64; ++i) dma16(rs, (unsigned)((m0 + 8 * (wid + 8 * i) + sub) * ld2 + cb), base + (unsigned)i * 8192u);
    return BM / 64;
  };

  // ---- fragment addressing
  const int lr = lane & 15, lg = lane >> 4;
  int ax[2][3][MJ];                                        // [k-step][dx][pixel tile]: byte address in the CURRENT patch buffer of tap (0, dx)
  int asx[MJ];                                             // F8: byte address in the CURRENT scale patch of (tap (0, 0) of row lr of pixel tile j, block lg)
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int ml = wm * (BM / 4) + j * 16 + lr;
    const int yl = ml / W;
    asx[j] = 2 * PB + NS * WST + (yl * PW + (ml - yl * W)) * 4 + (H2 ? (lg >> 1) * SCT + (lg & 1) : lg);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int pr = yl * PW + (ml - yl * W) + dx;
      ax[0][dx][j] = pr * 128 + ((lg ^ (pr & 6)) << 4);
      ax[1][dx][j] = ax[0][dx][j] ^ 64;
    }
  }
  const int wo0 = 2 * PB + (wn * TN) * 128 + lr * 128 + ((lg ^ ((lr >> 1) & 7)) << 4);      // + slot * WST + i * 2048
  const int wo1 = wo0 ^ 64;
  f4 acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  h8 wf[2][NI], xf[2][MJ];
  int sx[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j) sx[j] = 0x7F;
  auto read_frags = [&](int slot, int tap) {
    const int dy = tap / 3, dx = tap - 3 * dy, sw = dy & 1;      // (compile-time after unrolling)
    if constexpr (F8) {
#pragma unroll
      for (int j = 0; j < MJ; ++j) sx[j] = *(reinterpret_cast<const unsigned char*>(smem) + asx[j] + (dy * PW + dx) * 4);
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      xf[0][j] = *reinterpret_cast<const h8*>(smem + ax[sw][dx][j] + dy * PW * 128);
      xf[1][j] = *reinterpret_cast<const h8*>(smem + ax[sw ^ 1][dx][j] + dy * PW * 128);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      wf[0][i] = *reinterpret_cast<const h8*>(smem + wo0 + (slot * WST + i * 2048));
      wf[1][i] = *reinterpret_cast<const h8*>(smem + wo1 + (slot * WST + i * 2048));
    }
  };
  auto read_frags_x = [&](int slot, int abase) {          // a tile of the extra segment: rows of the activation slot at abase, weight ring slot `slot` (run-time)
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const int r = wm * (BM / 4) + j * 16 + lr;
      const int a = abase + r * 128 + ((lg ^ (r & 6)) << 4);
      xf[0][j] = *reinterpret_cast<const h8*>(smem + a);
      xf[1][j] = *reinterpret_cast<const h8*>(smem + (a ^ 64));
    }
    const char* wb = smem + slot * WST;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      wf[0][i] = *reinterpret_cast<const h8*>(wb + wo0 + i * 2048);
      wf[1][i] = *reinterpret_cast<const h8*>(wb + wo1 + i * 2048);
    }
  };
  auto mma = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
      typedef int v8i __attribute__((ext_vector_type(8)));
      typedef int v4i __attribute__((ext_vector_type(4)));
      v8i xv[MJ];
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        v4i lo = __builtin_bit_cast(v4i, xf[0][j]), hi = __builtin_bit_cast(v4i, xf[1][j]);
        xv[j] = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        v4i lo = __builtin_bit_cast(v4i, wf[0][i]), hi = __builtin_bit_cast(v4i, wf[1][i]);
        const v8i wv = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int j = 0; j < MJ; ++j)      // e4m3 x e4m3; weights at 2^0, activations with their block scales (block b's from lane group b)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, xv[j], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, sx[j]);
      }
      // (pin the results here: the intrinsic has no side effect, and without a use in this phase the compiler sinks a whole slab's MFMAs behind the
      // last barrier of the slab and parks the fragments in scratch)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) asm volatile("" :: "v"(acc[i][j]));
    } else {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j) acc[i][j] = mfma16<BF>(wf[f][i], xf[f][j], acc[i][j]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // bias and time-embedding values of this tile's columns (as in k_igemm_pp; a tile lies inside one image)
  float lb_b = 0.f, lb_c0 = 0.f;
  if (tid < BN && n0 + tid < p.N) {
    if (p.bias) lb_b = e2f<BF>(p.bias[n0 + tid]);
    if (p.bias_nc) lb_c0 = e2f<BF>(p.bias_nc[(long long)img * p.bias_nc_stride + n0 + tid]);
  }

  // ---- prologue: patch(0) whole, weight tiles 0 .. D - 1; tile 0's loads landed, the others in flight
#pragma unroll
  for (int i = 0; i < PPW; ++i) stage_patch(0, i);
  stage_sc(0);
  int nw = 0;
#pragma unroll
  for (int s_ = 0; s_ < D; ++s_) nw = stage_w(s_, 0, s_);  // (nt >= 9 > D; every tile of a wave carries the same number of weight loads)
  wait_vm_dyn(nw * (D - 1));
  barrier();                                               // P: patch(0) and weight tile 0 are visible to every wave
  if (grp == 1) barrier();                                 // the second half falls one barrier behind

#if TF_PP3_STAMP
  unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(p.partial) + (wid >> 2) * 64;
  const bool stamping = blockIdx.x == 0 && (wid & 3) == 0 && p.partial;
#define PP3_STAMP(k) do { if (TF_PP3_STAMP != 1) break; __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); if (stamping && g == 1 && lane == 0) stamps[tap * 6 + (k)] = t_; } while (0)
  if (stamping && wid == 0 && lane == 0) {                  // in-kernel clock: core-clock and 100 MHz counters around the K loop (MI355X_MICROARCH.md, DVFS give-back item 6)
    unsigned long long a_, b_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a_), "=s"(b_) :: "memory");
    stamps[56] = a_; stamps[57] = b_;
  }
#else
#define PP3_STAMP(k) do { } while (0)
#endif
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int t = g * 9 + tap;
      PP3_STAMP(0);
      read_frags(tap % NS, tap);                           // (tile t = 9 g + tap lives in slot t % 3 = tap % 3)
      int nl = 0;                                          // loads this wave issues during this tile: the newest ones, they stay in flight
      if (tap + D < 9) nl += stage_w((tap + D) % NS, g, tap + D);
      else if (g + 1 < G) nl += stage_w((tap + D) % NS, g + 1, tap + D - 9);
      else if (tap + D - 9 < GE) { nl += stage_wk((tap + D) % NS, (unsigned)(9 * Cc_ + (tap + D - 9) * 64) * 2u); nl += stage_x(tap + D - 9); }   // the first tiles of the extra segment
      if (tap < PPW && g + 1 < G) nl += stage_patch(g + 1, tap);
      if (F8 && tap == 7 && g + 1 < G) nl += stage_sc(g + 1);
      const bool next = t + 1 < nt + GE;
      PP3_STAMP(1);
      if (next && grp == 1) wait_vm_dyn(nl);               // tile t + 1 (and every patch piece of an earlier tile) landed
      wait_lds_reads();
      PP3_STAMP(2);
      barrier();
      PP3_STAMP(3);
      mma();
      PP3_STAMP(4);
      if (next && grp == 0) wait_vm_dyn(nl);
      PP3_STAMP(5);
      barrier();
    }
    // the next slab's patch is the other buffer
    const int flip = (g & 1) ? -PB : PB;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int j = 0; j < MJ; ++j) ax[f][dx][j] += flip;
    if constexpr (F8) {
#pragma unroll
      for (int j = 0; j < MJ; ++j) asx[j] += (g & 1) ? -SCB : SCB;
    }
  }
#if TF_PP3_STAMP
  if (stamping && wid == 0 && lane == 0) {
    unsigned long long a_, b_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a_), "=s"(b_) :: "memory");
    stamps[58] = a_; stamps[59] = b_;
  }
#endif
  // ---- the extra 1x1 segment: tile 9 G + e in weight slot e % 3 (9 G is a multiple of 3) and activation slot e % 3
  for (int e = 0; e < GE; ++e) {
    read_frags_x(e % NS, xslot(e));
    int nl = 0;
    if (e + D < GE) { nl += stage_wk((e + D) % NS, (unsigned)(9 * Cc_ + (e + D) * 64) * 2u); nl += stage_x(e + D); }
    const bool next = e + 1 < GE;
    if (next && grp == 1) wait_vm_dyn(nl);
    wait_lds_reads();
    barrier();
    mma();
    if (next && grp == 0) wait_vm_dyn(nl);
    barrier();
  }
  if (grp == 0) barrier();                                 // the first half waits for the second: every wave is done with the ring and the patches

  if constexpr (F8) {                                      // per-output-channel weight scales (this lane's 4 consecutive channels of every n-tile)
    if (p.wscale) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int n = n0 + wn * TN + i * 16 + lg * 4;
        f4 w = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n + e < p.N) w[e] = p.wscale[n + e];
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] *= w;
      }
    }
  }
  // ---- epilogue: two passes of 96 rows through the shared scratch (as k_igemm_pp)
  f4 csum[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) csum[i] = (f4){0.f, 0.f, 0.f, 0.f};
  constexpr int BS = BM / 2;
  float* const lbt = reinterpret_cast<float*>(smem + 4 * (BS / 2) * (TN + 4) * 4 + BS * 8 + 4 * BN * 8);
  if (tid < BN) { lbt[tid] = lb_b; lbt[BN + tid] = lb_c0; lbt[2 * BN + tid] = 0.f; }
  const int lb_m1 = (img + 1) * p.HoWo;
#pragma unroll
  for (int sm = 0; sm < 2; ++sm) {
    if ((wm >> 1) == sm) igemm_scratch_write<BS, BN>(p, acc, csum, smem, (wm & 1) | (wn << 1), lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
    igemm_epilogue<BS, BN, 0, BF, 2, true>(p, smem, m0 + sm * BS, n0, 0, wid & 3, wid >> 2, lane, lbt, n0, lb_m1);
    if (p.gn_part) igemm_gn_stats<BS, BN>(p, smem, m0 + sm * BS, n0, wid & 3, wid >> 2, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
  }
}
