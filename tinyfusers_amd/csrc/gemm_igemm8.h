// Part of the implicit-GEMM family of csrc/gemm.hip (see its head comment); split into translation units so that the
// instances compile in parallel.
#pragma once
#include "gemm_common.h"

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
