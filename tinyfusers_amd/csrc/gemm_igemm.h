// Part of the implicit-GEMM family of csrc/gemm.hip (see its head comment); split into translation units so that the
// instances compile in parallel.
#pragma once
#include "gemm_common.h"

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
template <int BM, int BN, bool GENERIC, bool WIDE, bool ALL8 = false, bool GI = false, bool BF = false>   // BF: bfloat16 operands / outputs (gemm_common.h: mfma16 / e2f / f2e)
__global__ void __launch_bounds__(512, WIDE ? 4 : 2) k_igemm(const GemmP p) {
  constexpr int TM = BM / 2, TN = BN / 2, MJ = TM / 16, NI = TN / 16;
  constexpr bool EPRE = !WIDE && TF_IGEMM_EPRE;                            // epilogue loads requested when the K loop ends (igemm_epilogue_prefetch); not on the 128-VGPR two-blocks-per-CU form (up to 48 registers)
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
#if TF_IGEMM_STAMP
  // diagnostic build: wave 0 (a consumer) and wave 4 (a loader) of every block keep four stamps of the constant 100 MHz clock each and store them when
  // they leave: [0] entry [1] K loop done (behind barrier X) [2] epilogue stores issued [3] stores drained | [4] entry [5] K tile 0 landed
  // [6] last stage issued [7] left.  The fences idle nothing the shipped kernel overlaps across these points (they sit at barriers).
  unsigned long long stp0 = __builtin_amdgcn_s_memrealtime(), stp1 = 0, stp2 = 0, stp3 = 0;
  auto stamp_out = [&](int base) {
    if (p.stamp && lane == 0 && blockIdx.x < 8192) { unsigned long long* d = p.stamp + (size_t)blockIdx.x * 8 + base; d[0] = stp0; d[1] = stp1; d[2] = stp2; d[3] = stp3; }
  };
#define IG_STAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
  // TF_IGEMM_STAMP == 2: per-K-tile stamps (s_memtime, core clock) of block 0's wave 4 (loader: 0 top, 1 tile it + 1 landed, 2 behind barrier(it), 3 next stage
  // issued) and wave 0 (consumer: 4 fragments of tile it in registers, 5 behind barrier(it)) for K tiles 4 .. 4 + 255, kept in the LDS above the ring
  // (the launch gets all 160 KiB) and copied out when the waves leave.  Only where lgkmcnt is 0 anyway, so no overlap of the shipped loop is fenced off.
  const bool lp_on = TF_IGEMM_STAMP == 2 && blockIdx.x == 0 && lane == 0 && (wid == 0 || wid == 4) && p.stamp && !WIDE && BM != 256;
  const unsigned lp_base = lds_off(smem) + 163840u - 256u * 32u;
  auto lp = [&](int it, int k) {
    if (TF_IGEMM_STAMP != 2) return;
    if (!lp_on || it < 4 || it >= 260) return;
    unsigned long long t_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
    asm volatile("ds_write_b32 %0, %1" :: "v"(lp_base + (unsigned)((it - 4) * 8 + k) * 4u), "v"((unsigned)t_) : "memory");
  };
  auto lp_out = [&]() {
    if (TF_IGEMM_STAMP != 2 || !lp_on) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < 256 * 8; ++i) {
      if (((i & 7) < 4) != (wid == 4)) continue;           // each wave copies its own columns
      unsigned v_;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v_) : "v"(lp_base + (unsigned)i * 4u) : "memory");
      reinterpret_cast<unsigned*>(p.stamp + 8192 * 8)[i] = v_;
    }
  };
#else
#define IG_STAMP(v) do { } while (0)
#endif

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
      const char* base = smem + slot * STAGE;
#pragma unroll
      for (int i = 0; i < LPS; ++i) {
        const int g = w4 + 4 * i;
        if (g * 8 < BM) {
          h8 x = *reinterpret_cast<const h8*>(base + g * 1024 + lane * 16);
          dot2_stats<BF>(x, ls[i], lq[i]);
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
          lds_write16(base + g * 1024, gi_apply<BF>(x, ga, gb, p.gi_silu, true));
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
      if (TF_ABL(p.dbg & 4)) return;
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
          bload_lds16_w(rs_w, off, dst);
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
#if TF_IGEMM_STAMP
        if (it == 0) IG_STAMP(stp1);
#endif
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
      // (TF_IGEMM_PRE stages are issued in front of barrier P -- gemm_common.h; with 2, tried in round 5, the stamps showed the consumers no longer waiting 0.6-1.2 us for nothing but the
      // ISSUE of ring slots 2 .. NS-1, at 60-100 cycles per 1-KiB piece; the rest of the ring follows right behind the barrier)
      constexpr int PRE = NS < TF_IGEMM_PRE ? NS : TF_IGEMM_PRE;
#pragma unroll
      for (int s_ = 0; s_ < PRE; ++s_)
        if (s_ < nt) stage(s_, kt_begin + s_);
      if (gi_on) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }   // barriers A, B of gi_prologue (consumer waves)
      wait_stages<LPS, PRE - 1>(nt - 1);                   // tile 0 landed; up to PRE-1 newer stages in flight
#if TF_IGEMM_STAMP
      IG_STAMP(stp1);
#endif
      if (gi_on && nt > 0) gi_tile(0, kt_begin);
      __builtin_amdgcn_s_barrier();                       // barrier P
      asm volatile("" ::: "memory");
#pragma unroll
      for (int s_ = PRE; s_ < NS; ++s_)
        if (s_ < nt) stage(s_, kt_begin + s_);
      if (ln_on && nt > 0) ln_tile(0);                    // slot 0 is refilled only after barrier(0)
      asm volatile("" ::: "memory");
      for (int it = 0; it < nt; ++it) {
#if TF_IGEMM_STAMP == 2
        lp(it, 0);
#endif
        if (it + 1 < nt) wait_stages<LPS, NS - 2>(nt - 2 - it);   // tile it+1 landed (ring holds up to tile it+NS-1 here)
#if TF_IGEMM_STAMP == 2
        lp(it, 1);
#endif
        if (gi_on && it + 1 < nt) gi_tile((it + 1) % NS, kt_begin + it + 1);
        __builtin_amdgcn_s_barrier();                     // barrier(it)
        asm volatile("" ::: "memory");
#if TF_IGEMM_STAMP == 2
        lp(it, 2);
#endif
        if (it + NS < nt) stage(it % NS, kt_begin + it + NS);
#if TF_IGEMM_STAMP == 2
        lp(it, 3);
#endif
        if (ln_on && it + 1 < nt) ln_tile((it + 1) % NS);  // tile it+1 stays in its slot until barrier(it+1)
      }
    }
    EpiPre<BM, BN> epre;
    if constexpr (EPRE) igemm_epilogue_prefetch<BM, BN>(p, m0, n0, w4, 1, lane, epre);     // (no counted vmcnt wait follows in this wave)
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
#if TF_IGEMM_STAMP
    IG_STAMP(stp2);
#endif
    __builtin_amdgcn_s_barrier();                         // barrier X: matches the consumers' "ring is free" barrier
    asm volatile("" ::: "memory");
    if (TF_ABL(p.dbg & 1)) return;
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
    igemm_epilogue<BM, BN, false, BF, 4, false, EPRE>(p, smem, m0, n0, split, w4, 1, lane, nullptr, 0, 0, epre);
    if (p.gn_part) igemm_gn_stats<BM, BN>(p, smem, m0, n0, w4, 1, lane);
#if TF_IGEMM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IG_STAMP(stp3);
    if (wid == 4) stamp_out(4);
    lp_out();
#endif
    return;
  }

  // ================================= CONSUMER WAVES ===============================================
  if constexpr (GI) gi_prologue<BF>(p, smem, m0 / p.HoWo, w4 * 64 + lane);   // first: its global loads must not wait behind this wave's own DMA (ALL8)
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
        bload_lds16_w(rs_cw, cw[i] != TF_OOB ? cw[i] + (unsigned)kt * 128u : TF_OOB, base + (NG - 4 * LPC + w4 + 4 * i) * 1024);
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
        for (int j = 0; j < MJ; ++j) acc[i][j] = mfma16<BF>(wf[i], xf[j], acc[i][j]);
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
        if (TF_ABL(p.dbg & 2)) {
#pragma unroll
          for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(wf[i]));
#pragma unroll
          for (int j = 0; j < MJ; ++j) asm volatile("" ::"v"(xf[j]));
          continue;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j) acc[i][j] = mfma16<BF>(wf[i], xf[j], acc[i][j]);

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
    if (TF_ABL(p.dbg & 2)) {
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
        for (int j = 0; j < MJ; ++j) acc[i][j] = mfma16<BF>(wf[k2][i], xf[k2][j], acc[i][j]);
  };
  constexpr int PRE = NS < TF_IGEMM_PRE ? NS : TF_IGEMM_PRE;   // (as the loaders: two stages in front of barrier P, the rest of the ring behind it)
  if constexpr (ALL8) {
#pragma unroll
    for (int s_ = 0; s_ < PRE; ++s_)
      if (s_ < nt) cstage(s_, kt_begin + s_);
    wait_stages<LPC, PRE - 1>(nt - 1);                     // this wave's pieces of tile 0 landed
  }
  __builtin_amdgcn_s_barrier();                           // barrier P: tile 0 landed
  asm volatile("" ::: "memory");
  if constexpr (ALL8) {
#pragma unroll
    for (int s_ = PRE; s_ < NS; ++s_)
      if (s_ < nt) cstage(s_, kt_begin + s_);
  }
  if (nt > 0) read_frags(0, wfA, xfA);
  for (int it = 0; it < nt; it += 2) {
    wait_lds_reads();    // fragments of tile it are in registers: its slot may be refilled
#if TF_IGEMM_STAMP == 2
    lp(it, 4);
#endif
    if constexpr (ALL8) { if (it + 1 < nt) wait_stages<LPC, NS - 2>(nt - 2 - it); }   // ... and this wave's pieces of tile it+1 landed
    __builtin_amdgcn_s_barrier();                         // barrier(it): tile it+1 landed
    asm volatile("" ::: "memory");
#if TF_IGEMM_STAMP == 2
    lp(it, 5);
#endif
    if constexpr (ALL8) { if (it + NS < nt) cstage(it % NS, kt_begin + it + NS); }
    if (it + 1 < nt) read_frags((it + 1) % NS, wfB, xfB);
    __builtin_amdgcn_sched_barrier(0);
    mma(wfA, xfA);
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 >= nt) break;
    wait_lds_reads();
#if TF_IGEMM_STAMP == 2
    lp(it + 1, 4);
#endif
    if constexpr (ALL8) { if (it + 2 < nt) wait_stages<LPC, NS - 2>(nt - 3 - it); }
    __builtin_amdgcn_s_barrier();                         // barrier(it+1)
    asm volatile("" ::: "memory");
#if TF_IGEMM_STAMP == 2
    lp(it + 1, 5);
#endif
    if constexpr (ALL8) { if (it + 1 + NS < nt) cstage((it + 1) % NS, kt_begin + it + 1 + NS); }
    if (it + 2 < nt) read_frags((it + 2) % NS, wfA, xfA);
    __builtin_amdgcn_sched_barrier(0);
    mma(wfB, xfB);
    __builtin_amdgcn_sched_barrier(0);
  }
  }
  EpiPre<BM, BN> epre;
  if constexpr (EPRE) igemm_epilogue_prefetch<BM, BN>(p, m0, n0, w4, 0, lane, epre);       // the K loop is over (its fragments are dead): the epilogue's loads fly under barriers X / Y and the scratch write
  __builtin_amdgcn_s_barrier();                           // barrier X: every consumer is done with the ring
  asm volatile("" ::: "memory");
#if TF_IGEMM_STAMP
  IG_STAMP(stp1);
#endif
  if (TF_ABL(p.dbg & 1)) {
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
  igemm_epilogue<BM, BN, false, BF, 4, false, EPRE>(p, smem, m0, n0, split, w4, 0, lane, nullptr, 0, 0, epre);
  if (p.gn_part) igemm_gn_stats<BM, BN>(p, smem, m0, n0, w4, 0, lane);
#if TF_IGEMM_STAMP
  IG_STAMP(stp2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  IG_STAMP(stp3);
  if (wid == 0) stamp_out(0);
  lp_out();
#endif
}
#undef IG_STAMP
