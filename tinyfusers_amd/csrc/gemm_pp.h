// Part of the implicit-GEMM family of csrc/gemm.hip (see its head comment); split into translation units so that the
// instances compile in parallel.
#pragma once
#include "gemm_common.h"

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
// NP = half-phases per K tile and wave group: 2 = one per 32-deep k-step (fragments of one k-step in registers), 1 = the whole K tile per
//   phase (both k-steps' fragments in registers, half the barriers; needs the 3-slot ring: with two slots the second half would issue a
//   tile's pieces and wait for them in the same segment).
// FASTA = the lean activation addressing for stride-1 convolutions without up-sampling (and linears): per piece a pixel index and a
//   bit mask of the taps that fall inside the image, so a tile's source offset is one mad + one mask test instead of the bounds
//   arithmetic of the general gather (the load segments, not the MFMAs, set this kernel's pace: every VALU / SALU instruction in them counts).
// DBG: the ablation build (p.dbg: 1 no epilogue, 2 no MFMA, 4 no staging in the loop, 8 no fragment reads)
// F8 = OCP e4m3 operands (BASELINE config 5) on the block-scaled MFMA v_mfma_scale_f32_16x16x128_f8f6f4: 128-deep K per instruction at twice the
//   fp16 rate.  The LDS image is the fp16 kernel's byte for byte -- a K tile is 128 BYTES of every row, i.e. 128 e4m3 elements -- and so are the
//   fragment reads: lane group lg takes chunk lg and chunk lg + 4 of its row, which IS the instruction's register layout (registers 0-3 hold
//   k = 16 lg .. 16 lg + 15, registers 4-7 hold k = 64 + 16 lg ..; tools/probe_mx.hip).  ACTIVATIONS are block scaled ("MX", common.h: mx_quant8):
//   every 32 consecutive channels of a pixel share one E8M0 byte, stored behind the tensor's codes; the instruction takes the scale of K block b
//   (k = 32 b .. 32 b + 31 of the 128) from lane group b, byte 0 of the scale operand (the same probe).  The four bytes of a (row, K tile) travel
//   with the tile: waves 4-7 DMA them (buffer_load_dword / _ushort ... lds, one lane per row, 4 bytes of LDS per lane) into a small table
//   behind the ring, slot by slot like the tile itself, and a lane reads the byte of (its row, block lg) with one ds_read_u8 per fragment.
//   WEIGHTS carry one fp32 scale per output channel (tf_pack_weight_fp8), applied to the accumulators in front of the shared epilogue; their
//   hardware scale is 2^0.  A K tile is two 64-channel HALVES that may lie in different taps / source tensors (320 channels = 2.5 tiles): H2 =
//   true issues every activation piece as two half-masked loads with their own descriptor and offsets, and the scales as two 2-byte loads (same
//   counts every tile: the vmcnt bookkeeping stays static); H2 = false (every channel count a multiple of 128) one load each.  F8 instances
//   exist for the lean addressing only (FASTA: stride-1 convolutions without up-sampling, linears).
// BM = 256 or 192 rows: 192 (wave tiles of 48 rows) exists for the tile COUNT -- 96 x 96 latents give M = 9216 * images rows, and
//   e.g. 73728 x 320 is 576 tiles of 256 x 160 = 2.25 rounds on 256 CUs but 768 tiles of 192 x 160 = 3 rounds exactly.
// LNF = the LayerNorm fold (tf_linear_ln_f16: Linear(LN(x)) = rstd[m] (x . w'^T - mean[m] colsum[n]) + bias'[n]): the row statistics come from
//   the activation FRAGMENTS the wave multiplies anyway -- lane (lr, lg) holds the 8 k-values k = 8 lg .. of row lr of every fragment, so
//   8 v_dot2_f32_f16 per fragment (in the MFMA block's spare issue slots) keep (sum, sum of squares) of that row's share, two lane
//   shuffles at the end complete the row -- and they end up in exactly the lanes whose accumulators belong to that row.
#ifndef TF_PP_V2
#define TF_PP_V2 0        // experiment switch (tagged build): 1 = one barrier per K tile, loads issued between MFMA chunks (see the tile loop)
#endif
#ifndef TF_PP_H2_MERGE
#define TF_PP_H2_MERGE 1  // e4m3, channel counts off the 128 grid: one full-width load where the two slabs of a K tile are contiguous (0: always two half-masked loads)
#endif
#ifndef TF_PP_PRIO
#define TF_PP_PRIO 0      // experiment switch of tools' tagged builds: 0 = s_setprio 1 around every MFMA block (shipped), 1 = static priority for waves 4-7, 2 = none
#endif
template <int BN, int NP, bool FASTA, bool DBG = false, bool F8 = false, bool H2 = false, int BM = 256, bool LNF = false, bool BF = false>   // BF: bfloat16 operands / outputs (gemm_k_pp16_bf16.hip)
__global__ void __launch_bounds__(512, 2) k_igemm_pp(const GemmP p) {
  static_assert(!(BF && F8), "the e4m3 form has fp16 bias / residual / outputs");
  static_assert(!LNF || !F8, "the LayerNorm fold is an fp16 path");
  constexpr int TN = BN / 2, MJ = BM / 64, NI = TN / 16;
  constexpr int APW = BM / 64;                            // activation pieces (8 rows x 128 B) per wave and stage: BM / 8 pieces in front of the weight pieces
  constexpr int ES = F8 ? 1 : 2;                          // bytes per element
  constexpr int APL = (F8 && H2) ? 2 * APW : APW;         // activation loads per wave and K tile
  static_assert(BM == 256 || BM == 192, "block rows");
  static_assert(!F8 || NP == 1, "the 128-deep MFMA takes both 64-byte halves of a row at once");
  static_assert(F8 || !H2, "half-masked activation loads are the fp8 kernel's");
  static_assert(!F8 || FASTA, "the e4m3 instances use the lean addressing");
  constexpr int SCL = F8 ? (H2 ? 2 : 1) : 0;              // scale loads per K tile of a wave that stages scales (waves 4-7)
  constexpr int SCS = SCL * 1024;                         // bytes of a ring slot's scale table: one dword per row and half, 4 waves x 64 rows (the 192-row tile leaves the last 64 unused)
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
  // F8: lane L of wave 4 + w stages the scale bytes of tile row 64 w + L: (pixel index of the output position, tap-validity mask) as g_a / g_b
  int s_pix = 0, s_mask = 0;
  if constexpr (F8) {
    const int m = m0 + 64 * (wid - 4) + lane;
    if (wid >= 4 && 64 * (wid - 4) + lane < BM && m < p.M) {
      int img = fast_div(m, p.dv_howo_mul, p.dv_howo_shr), rem = m - img * p.HoWo;
      int ho = fast_div(rem, p.dv_wo_mul, p.dv_wo_shr), wo = rem - ho * p.Wo;
      s_pix = img * p.H * p.W + ho * p.W + wo;
      unsigned mask = 0x80000000u;
      for (int r = 0; r < p.S; ++r)
        for (int s_ = 0; s_ < p.S; ++s_)
          if ((unsigned)(ho - p.pad + r) < (unsigned)p.H && (unsigned)(wo - p.pad + s_) < (unsigned)p.W) mask |= 1u << (r * p.S + s_);
      s_mask = (int)mask;
    }
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
  // pieces [i0, i1) of this wave's APL activation loads of a tile (H2: the first APW are the first slab's half-masked loads, the next APW the second's)
  auto stage_act = [&](int slot, int i0 = 0, int i1 = 64) {
    const unsigned base = lds0 + (unsigned)slot * STAGE + (unsigned)wid * 1024u;
    // (the scalars are wave-uniform by construction; the readfirstlanes are no-ops that keep them in SGPRs whatever the compiler's
    // divergence analysis makes of the bookkeeping's control flow)
    auto one = [&](int lo, int hi, int nb, int r_, int s_, int c0_, int ld_, int cq, int hsel, int j0, int j1) {
      // cq: this lane's 16-byte chunk inside the slab; hsel < 0: every lane issues, else only the lanes of half hsel; pieces [j0, j1) of APW
      if (j0 >= j1) return;
      i4v rs;
      rs[0] = __builtin_amdgcn_readfirstlane(lo); rs[1] = __builtin_amdgcn_readfirstlane(hi);
      rs[2] = __builtin_amdgcn_readfirstlane(nb); rs[3] = 0x00020000;
      const int s_r = __builtin_amdgcn_readfirstlane(r_), s_c0 = __builtin_amdgcn_readfirstlane(c0_), s_ld = __builtin_amdgcn_readfirstlane(ld_);
      const bool mine = hsel < 0 || (cs >> 2) == hsel;
      if constexpr (FASTA) {
        const int vc = s_c0 + cq * 16;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
          if (i < j0 || i >= j1) continue;
          unsigned off = __umul24((unsigned)g_a[i], (unsigned)s_ld) + (unsigned)vc;
          if (mine) dma16(rs, (g_b[i] & s_r) ? off : TF_OOB, base + (unsigned)i * 8192u);
        }
      } else {
        const int s_s = __builtin_amdgcn_readfirstlane(s_);
        const int cc = s_c0 + cq * (16 / ES);
#pragma unroll
        for (int i = 0; i < APW; ++i) {
          if (i < j0 || i >= j1) continue;
          int hi_ = g_a[i] + s_r, wi = g_b[i] + s_s;
          bool ok = (unsigned)hi_ < (unsigned)Hl && (unsigned)wi < (unsigned)Wl;
          int pix = g_c[i] + (hi_ >> ups) * Wd + (wi >> ups);
          if (mine) dma16(rs, ok ? (unsigned)(pix * s_ld + cc) * ES : TF_OOB, base + (unsigned)i * 8192u);
        }
      }
    };
    const auto lim = [](int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; };
    if constexpr (!F8) one(a_lo, a_hi, a_nb, a_r, a_s, a_c0, a_ld, cs, -1, i0, i1);
    else if constexpr (H2) {
      // the two 64-channel slabs of this K tile are usually 128 CONTIGUOUS bytes of one (tap, tensor) -- with 320 channels 4 tiles of 5 -- and then
      // one full-width load does it: the load instructions of the load half, not bytes, set this kernel's pace (13 per wave and tile against 9).
      // (wave-uniform test on the prepared scalars; the counted waits assume the smaller number of loads: see wait_landed)
      const bool contig = TF_PP_H2_MERGE && FASTA && __builtin_amdgcn_readfirstlane((b_lo == a_lo) & (b_hi == a_hi) & (b_nb == a_nb) & (a_r == b_r) & (a_c0 == b_c0 + 64) & (a_ld == b_ld));
      if (contig) one(b_lo, b_hi, b_nb, b_r, b_s, b_c0, b_ld, cs, -1, lim(i0, 0, APW), lim(i1, 0, APW));
      else {
        one(b_lo, b_hi, b_nb, b_r, b_s, b_c0, b_ld, cs & 3, 0, lim(i0, 0, APW), lim(i1, 0, APW));
        one(a_lo, a_hi, a_nb, a_r, a_s, a_c0, a_ld, cs & 3, 1, lim(i0 - APW, 0, APW), lim(i1 - APW, 0, APW));
      }
    } else one(b_lo, b_hi, b_nb, b_r, b_s, b_c0, b_ld, cs, -1, i0, i1);       // channel counts on the 128 grid: the two slabs of a tile are 128 contiguous bytes
  };
  auto stage_w = [&](int slot, int kt, int i0 = 0, int i1 = 64) {
    const unsigned base = lds0 + (unsigned)slot * STAGE + (unsigned)(BM / 8 + wid) * 1024u;
    const unsigned kb = (unsigned)kt * 128u;
#pragma unroll
    for (int i = 0; i < WPW; ++i)
      if (i >= i0 && i < i1 && (WREM == 0 || i < WPW - 1 || wid < WREM)) dma16_w(rs_w, (gw[i] != TF_OOB && (!F8 || (int)kb < klim)) ? gw[i] + kb : TF_OOB, base + (unsigned)i * 8192u);
  };
  // F8: the E8M0 bytes of the tile's activation rows (waves 4-7; every one of them issues SCL loads so that the counts stay uniform -- the
  // rows beyond BM of the 192-row tile fetch out of range).  A source tensor holds its scale bytes behind its codes (offset = the codes'
  // byte count `nb`), C / 32 per pixel; a 64-channel slab starting at byte offset c0 of a pixel's codes has its 2 bytes at c0 / 32.
  auto stage_sc = [&](int slot, int i0 = 0, int i1 = 2) {
    if constexpr (F8) {
      if (wid < 4) return;
      const unsigned base = lds0 + (unsigned)NS * STAGE + (unsigned)slot * SCS + (unsigned)(wid - 4) * 256u;
      auto one = [&](int lo, int hi, int nb, int r_, int c0_, int ld_, unsigned ldsb, auto wide) {
        i4v rs;
        const int s_nb = __builtin_amdgcn_readfirstlane(nb);
        rs[0] = __builtin_amdgcn_readfirstlane(lo); rs[1] = __builtin_amdgcn_readfirstlane(hi);
        rs[2] = s_nb + (s_nb >> 5); rs[3] = 0x00020000;
        const int s_r = __builtin_amdgcn_readfirstlane(r_), s_c0 = __builtin_amdgcn_readfirstlane(c0_) >> 5, s_ld = __builtin_amdgcn_readfirstlane(ld_) >> 5;
        const unsigned off = (s_mask & s_r) ? (unsigned)s_nb + __umul24((unsigned)s_pix, (unsigned)s_ld) + (unsigned)s_c0 : TF_OOB;
        if constexpr (decltype(wide)::value)
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" :: "s"(__builtin_amdgcn_readfirstlane((int)ldsb)), "v"(off), "s"(rs) : "memory");
        else
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ushort %1, %2, 0 offen lds" :: "s"(__builtin_amdgcn_readfirstlane((int)ldsb)), "v"(off), "s"(rs) : "memory");
      };
      if constexpr (H2) {
        if (i0 <= 0 && i1 > 0) one(b_lo, b_hi, b_nb, b_r, b_c0, b_ld, base, std::false_type{});
        if (i0 <= 1 && i1 > 1) one(a_lo, a_hi, a_nb, a_r, a_c0, a_ld, base + 1024u, std::false_type{});
      } else if (i0 <= 0 && i1 > 0) one(b_lo, b_hi, b_nb, b_r, b_c0, b_ld, base, std::true_type{});
    }
  };
  // "this wave's pieces of every tile but the newest one (NEWEST) / of every tile (!NEWEST) have landed"
  auto wait_landed = [&](auto newest) {
    if constexpr (decltype(newest)::value && D >= 2) {
      // loads of one tile by this wave: APL activation pieces, WPW (or one fewer from wave WREM on) weight pieces, SCL scale loads on waves 4-7
      static_assert(WREM == 0 || WREM == 4, "the wave classes below");
      // (H2 with merging: a tile issues APW or 2 APW activation loads; the wait assumes APW -- with the newest tile's loads at least that many in
      // flight, "at most that many outstanding" still means every older load has landed; a split tile's extra loads are waited for a little early)
      constexpr int APLW = (F8 && H2 && TF_PP_H2_MERGE) ? APW : APL;
      if (wid < 4) wait_vm<APLW + WPW>(); else wait_vm<APLW + WPW - (WREM ? 1 : 0) + SCL>();
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
  // F8: the E8M0 byte of (row lr of pixel tile j, K block lg) of the tile being multiplied: the instruction takes block b's scale from lane group b
  int sx[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j) sx[j] = 0x7F;
  const int sc_lane = NS * STAGE + (H2 ? (lg >> 1) * 1024 + (lg & 1) : lg) + (wm * (BM / 4) + lr) * 4;
  auto read_sc = [&](int slot) {
    if constexpr (F8) {
      const unsigned char* sc = reinterpret_cast<const unsigned char*>(smem) + sc_lane + slot * SCS;
#pragma unroll
      for (int j = 0; j < MJ; ++j) sx[j] = sc[j * 64];
    }
  };
  float ls[MJ], lq[MJ];                                    // LNF: this lane's share of (sum x, sum x^2) of row lr of every pixel tile
#pragma unroll
  for (int j = 0; j < MJ; ++j) { ls[j] = 0.f; lq[j] = 0.f; }
  auto mma = [&]() {                                       // the MFMAs of every k-step held in registers
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LNF) {
#pragma unroll
      for (int f = 0; f < KF; ++f)
#pragma unroll
        for (int j = 0; j < MJ; ++j) dot2_stats<BF>(xf[f][j], ls[j], lq[j]);
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
    if (TF_PP_PRIO == 0) __builtin_amdgcn_s_setprio(1);
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
        for (int j = 0; j < MJ; ++j)      // e4m3 x e4m3; weights at 2^0 (their per-channel scale multiplies the accumulators later), activations with their block scales
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, xv[j], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, sx[j]);
      }
    } else {
#pragma unroll
      for (int f = 0; f < KF; ++f)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j) acc[i][j] = mfma16<BF>(wf[f][i], xf[f][j], acc[i][j]);
    }
    if (TF_PP_PRIO == 0) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // chunk c of the MFMA block of a K tile (the experiment's form): the MJ instructions of (k-step c / NI, channel tile c % NI)
  auto mma_chunk = [&](int c) {
    const int f = c / NI, i = c % NI;
    if constexpr (F8) {
      typedef int v8i __attribute__((ext_vector_type(8)));
      typedef int v4i __attribute__((ext_vector_type(4)));
      v4i lo = __builtin_bit_cast(v4i, wf[0][i]), hi = __builtin_bit_cast(v4i, wf[KF - 1][i]);
      const v8i wv = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (f == 0) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          v4i xl = __builtin_bit_cast(v4i, xf[0][j]), xh = __builtin_bit_cast(v4i, xf[KF - 1][j]);
          const v8i xv = (v8i){xl[0], xl[1], xl[2], xl[3], xh[0], xh[1], xh[2], xh[3]};
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, xv, acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, sx[j]);
        }
      }
    } else {
      if constexpr (LNF) {
        if (i == 0) {
#pragma unroll
          for (int j = 0; j < MJ; ++j) dot2_stats<BF>(xf[f][j], ls[j], lq[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[i][j] = mfma16<BF>(wf[f][i], xf[f][j], acc[i][j]);
    }
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
    if (p.bias) lb_b = e2f<BF>(p.bias[n0 + tid]);
    if (p.bias_nc) {
      lb_c0 = e2f<BF>(p.bias_nc[(long long)lb_img0 * p.bias_nc_stride + n0 + tid]);
      if ((lb_img0 + 1) * p.HoWo < p.M) lb_c1 = e2f<BF>(p.bias_nc[(long long)(lb_img0 + 1) * p.bias_nc_stride + n0 + tid]);
    }
  }
  // ---- prologue: the first D tiles, whole
#pragma unroll
  for (int s_ = 0; s_ < D; ++s_)
    if (s_ < nt) { prep_tile(); stage_act(s_); stage_w(s_, kt_begin + s_); stage_sc(s_); }
  if (D < nt) prep_tile();                                 // the scalars of tile D: its pieces ride on tile 0
  if (D >= 2 && nt >= 2) wait_landed(std::true_type{}); else wait_landed(std::false_type{});     // tile 0 landed
  barrier();                                               // P: tile 0 is visible to every wave
  if (grp == 1 && !(NP == 1 && TF_PP_V2 != 0)) barrier();  // the second half falls one barrier behind
  if (TF_PP_PRIO == 1 && grp == 1) __builtin_amdgcn_s_setprio(1);      // (experiment: static priority for the later-dispatched half, no per-segment flips)
  int rs = 0, ws = D % NS;                                 // ring slot of tile t / of tile t + D
  int ktw = kt_begin + D;                                  // K tile whose weight pieces are staged next
  // one K tile.  MORE: tile t + D exists (its pieces are issued during this tile, and the wait for tile t + 1 leaves them in flight);
  // NEXT: tile t + 1 exists (it must have landed before the barrier in front of the first half's next load segment).
  auto tile = [&](auto more_c, auto next_c) {
    constexpr bool MORE = decltype(more_c)::value, NEXT = decltype(next_c)::value;
    const char* sb = smem + rs * STAGE;
    bool more = MORE;
    if constexpr (DBG) { if (p.dbg & 4) more = false; }
    if constexpr (NP == 1 && TF_PP_V2 != 0) {
      // EXPERIMENT (tagged build -DTF_PP_V2=1): no ping-pong -- every wave runs the same phase, ONE barrier per K tile; a wave's loads of tile
      // t + D are issued piece by piece between the chunks of its MFMA block (an LDS-DMA issue costs ~60 cycles among bare MFMAs against
      // 100-185 inside a load segment that also carries the fragment reads: MI355X_MICROARCH.md), the partner wave of the SIMD covers the stalls
      read_k(sb, 0, 0);
      read_k(sb, 1, 1);
      read_sc(rs);
      constexpr int CH = F8 ? NI : KF * NI, PT = APL + WPW + SCL;      // MFMA chunks (MJ instructions each) and load instructions of a tile
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        mma_chunk(c);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
          const int q0 = c * PT / CH, q1 = (c + 1) * PT / CH;                 // pieces issued behind this chunk
          stage_act(ws, q0, q1 < APL ? q1 : APL);
          stage_w(ws, ktw, q0 - APL, q1 - APL);
          stage_sc(ws, q0 - APL - WPW, q1 - APL - WPW);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (MORE) prep_tile();
      if constexpr (NEXT) wait_landed(more_c);
      wait_lds_reads();
      barrier();
    } else if constexpr (NP == 1) {
      read_k(sb, 0, 0);
      read_k(sb, 1, 1);
      read_sc(rs);
      if (more) { stage_act(ws); stage_w(ws, ktw); stage_sc(ws); }
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
  if (grp == 0 && !(NP == 1 && TF_PP_V2 != 0)) barrier();  // the first half waits for the second: every wave is done with the ring

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
    if (F8 && BN == 128 && p.out8) igemm_epilogue<BS, BN, (F8 && BN == 128) ? 2 : 0, false, 2, true>(p, smem, m0 + sm * BS, n0, split, wid & 3, wid >> 2, lane, lbt, n0, lb_m1);   // (block-scaled GEGLU output: the host admits it for act = 1 and 128-wide tiles only)
    else igemm_epilogue<BS, BN, false, BF, 2, true>(p, smem, m0 + sm * BS, n0, split, wid & 3, wid >> 2, lane, lbt, n0, lb_m1);
    if (p.gn_part && m0 + sm * BS < p.M) igemm_gn_stats<BS, BN>(p, smem, m0 + sm * BS, n0, wid & 3, wid >> 2, lane);   // (block-uniform: the barrier inside is safe)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
  }
}
