// Fused flash-style scaled-dot-product attention for gfx950 (wave64, MFMA 16x16x32 f16, online softmax).
// Replaces attention/sdpa.py:53-77 (cp.matmul -> softmax_kernel -> cp.matmul with the (B,NH,Tq,Tk) fp32
// score matrix materialised twice in HBM): scores never leave registers.
//
// Block = 4 waves, 128 query rows (32 per wave); K/V tiles of 64 keys staged through LDS (register-staged
// prefetch, two LDS stages).  Per wave and tile:
//   S^T (64 keys x 32 queries) = K_tile . Q^T      -- K rows are the MFMA A operand (ds_read_b128),
//                                                     Q fragments stay in registers for the whole kernel;
//   softmax along keys is lane-local (query = lane & 15) plus two xor-shuffles across the 4 lane groups.  The loop is
//   VALU-issue bound at d = 40 (v_exp 8 cycles, everything else 4), so the softmax is cut to max3 + exp2 + cvt per
//   score: Q is pre-scaled by log2(e)/sqrt(d), the running reference maximum enters as the INITIAL ACCUMULATOR of
//   the QK^T chain (S' = K Q^T - m needs no subtract), and the row sums come out of the PV product itself through
//   a column of ones in the padding of the V tile (column DV-1 >= HS), rescaled together with O for free;
//   O^T (d x 32 queries) += V^T . P^T              -- P^T is already in B-operand layout (accumulator ->
//                                                     operand, keys permuted so each lane group owns 8
//                                                     consecutive keys), V^T comes from the row-major V
//                                                     tile with ds_read_b64_tr_b16.
#include "common.h"
#include "../../include/tinyfusers_hip.h"

struct SdpaP {
  const half_t* q; const half_t* k; const half_t* v; half_t* o;
  int B, NH, Tq, Tk, HS;
  long long q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st;
  float scale_log2e;
  int causal;
  int dbg;              // ablation build only (TF_SDPA_DBG, tools/sdpa_dbg.py): 1 no exp, 2 no P.V MFMAs, 4 no Q.K MFMAs, 8 no barrier, 16 no DMA in the loop, 32 no V reads, 64 no max
};

// max over the four 16-lane groups of the wave (lanes l, l^16, l^32, l^48), result in every lane: gfx950's v_permlane16_swap /
// v_permlane32_swap exchange rows between two registers in the VALU -- swap(x, x) leaves (row0,row0,row2,row2) and (row1,row1,row3,row3),
// whose max is the xor-16 butterfly; the 32-lane swap finishes it.  The ds_bpermute form (__shfl_xor) put two dependent LDS round trips
// in front of every tile's exponentials (self-attention 64 x 64, d = 40: 88.4 -> 85.3 us).
__device__ __forceinline__ float max_over_lane_groups(float v) {
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  float m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  u2v q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

__device__ __forceinline__ s4v lds_tr16(const half_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p);
}

// 1-D grid, XCD-aware: blocks i and i+8 share an XCD (and its L2).  Every XCD gets a contiguous run of work items with
// the query block fastest, so all query blocks of one (batch, head) sit on one XCD and its K/V stream through that
// L2 once instead of through all eight (bijective remap, any grid size).
struct SdpaBlk { int qb, h, b; };
__device__ __forceinline__ SdpaBlk sdpa_block(const SdpaP& p, const int QB = 128) {
  const int nqb = (p.Tq + QB - 1) / QB, nblk = nqb * p.NH * p.B;
  int bid = blockIdx.x;
  int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  SdpaBlk o;
  int bh = bid / nqb;
  o.qb = bid - bh * nqb;
  o.b = bh / p.NH;
  o.h = bh - o.b * p.NH;
  return o;
}

template <int DQK, int DV, bool BF = false>   // BF: bfloat16 q / k / v / o (sdpa_bf16.hip)
__global__ void __launch_bounds__(256) k_sdpa(const SdpaP p) {
  constexpr int KS = DQK + 8;                 // K row stride (halves); 16-B multiple
  constexpr int VS = DV + 8;                  // V row stride (halves); 16-B multiple
  constexpr int NKS = DQK / 32;               // k-steps of QK^T
  constexpr int NDT = DV / 16;                // d tiles of PV (the last one holds the ones column: DV > HS)
  constexpr int CPR = DV / 8;                 // 16-B chunks per staged row (covers HS <= DV)
  constexpr int NCH = (64 * CPR + 255) / 256; // staging rounds per tensor
  constexpr int STAGE_H = 64 * KS + 64 * VS;  // halves per stage
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  half_t* smem = reinterpret_cast<half_t*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const SdpaBlk blk = sdpa_block(p);
  const int b = blk.b, h = blk.h;
  const int qblk = blk.qb * 128 + wid * 32;
  const half_t* qb = p.q + b * p.q_sb + h * p.q_sh;
  const half_t* kb = p.k + b * p.k_sb + h * p.k_sh;
  const half_t* vb = p.v + b * p.v_sb + h * p.v_sh;

  // zero both stages once: padding columns [HS, DQK) of K and [HS, DV) of V are never written afterwards
  for (int i = tid; i < 2 * STAGE_H / 8; i += 256) reinterpret_cast<h8*>(smem)[i] = (h8){0, 0, 0, 0, 0, 0, 0, 0};

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[query lr][d = 32 ks + 8 lg + j]
  h8 qf[2][NKS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    int qi = qblk + qt * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      int d0 = ks * 32 + lg * 8;
      h8 qv = (qi < p.Tq && d0 < p.HS) ? *reinterpret_cast<const h8*>(qb + qi * p.q_st + d0) : (h8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 8; ++j) qv[j] = f2e<BF>(e2f<BF>(qv[j]) * p.scale_log2e);
      qf[qt][ks] = qv;
    }
  }

  f4 ot[NDT][2];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) { ot[dt][0] = (f4){0, 0, 0, 0}; ot[dt][1] = (f4){0, 0, 0, 0}; }
  float m_run[2] = {0.f, 0.f};    // reference maximum (log2 units); set from the first tile, then only raised

  int ntiles = (p.Tk + 63) / 64;
  if (p.causal) {   // keys beyond the block's last query are never needed
    int last_q = min(p.Tq, blk.qb * 128 + 128) - 1;
    ntiles = min(ntiles, last_q / 64 + 1);
  }

  // K/V tiles: raw buffer loads with 32-bit offsets; rows >= Tk and padding columns fall outside the range check
  // and come back as zeros (no branches, no 64-bit address math)
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, (unsigned)(((long long)(p.Tk - 1) * p.k_st + p.HS) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, (unsigned)(((long long)(p.Tk - 1) * p.v_st + p.HS) * 2), 0x00020000);
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  union U4H8 { u4 u; h8 h; };
  h8 kreg[NCH], vreg[NCH];
  int st_row[NCH], st_col[NCH];
#pragma unroll
  for (int r = 0; r < NCH; ++r) {
    int idx = tid + 256 * r;
    st_row[r] = idx / CPR;
    st_col[r] = (idx - st_row[r] * CPR) * 8;
  }
  auto load_tile = [&](int t) {
#pragma unroll
    for (int r = 0; r < NCH; ++r) {
      int key = t * 64 + st_row[r];
      bool ok = st_row[r] < 64 && key < p.Tk && st_col[r] < p.HS;
      unsigned ko = ok ? (unsigned)(key * (int)p.k_st + st_col[r]) * 2u : 0x80000000u;
      unsigned vo = ok ? (unsigned)(key * (int)p.v_st + st_col[r]) * 2u : 0x80000000u;
      U4H8 a, b2;
      a.u = __builtin_amdgcn_raw_buffer_load_b128(rs_k, ko, 0, 0);
      b2.u = __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo, 0, 0);
      kreg[r] = a.h; vreg[r] = b2.h;
    }
  };
  auto store_tile = [&](int buf) {
    half_t* ks_ = smem + buf * STAGE_H;
    half_t* vs_ = ks_ + 64 * KS;
#pragma unroll
    for (int r = 0; r < NCH; ++r) {
      if (st_row[r] < 64 && st_col[r] < p.HS) {
        *reinterpret_cast<h8*>(ks_ + st_row[r] * KS + st_col[r]) = kreg[r];
        *reinterpret_cast<h8*>(vs_ + st_row[r] * VS + st_col[r]) = vreg[r];
      }
    }
  };

  load_tile(0);
  __syncthreads();           // zero-fill complete before the first tile lands on top of it
  store_tile(0);
  if (tid < 128) smem[(tid >> 6) * STAGE_H + 64 * KS + (tid & 63) * VS + DV - 1] = f2e<BF>(1.0f);   // ones column -> row sums
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile(t + 1);
    const half_t* ks_ = smem + buf * STAGE_H;
    const half_t* vs_ = ks_ + 64 * KS;

    // ---- S^T = K Q^T : st[kt][qt], key(kt, row) = 32 (kt>>1) + 8 (row>>2) + 4 (kt&1) + (row&3)
    f4 st[4][2];
    const f4 init4[2] = {{-m_run[0], -m_run[0], -m_run[0], -m_run[0]}, {-m_run[1], -m_run[1], -m_run[1], -m_run[1]}};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      int krow = 32 * (kt >> 1) + 8 * (lr >> 2) + 4 * (kt & 1) + (lr & 3);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        h8 kf = *reinterpret_cast<const h8*>(ks_ + krow * KS + ks * 32 + lg * 8);
        st[kt][0] = mfma16<BF>(kf, qf[0][ks], ks == 0 ? init4[0] : st[kt][0]);
        st[kt][1] = mfma16<BF>(kf, qf[1][ks], ks == 0 ? init4[1] : st[kt][1]);
      }
    }
    // ---- masks: this lane's keys are t*64 + 32 (kt>>1) + 8 lg + 4 (kt&1) + reg
    const int kbase = t * 64 + 8 * lg;
    if (t * 64 + 64 > p.Tk || p.causal) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int key = kbase + 32 * (kt >> 1) + 4 * (kt & 1) + e;
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            int qi = qblk + qt * 16 + lr;
            if (key >= p.Tk || (p.causal && key > qi)) st[kt][qt][e] = -INFINITY;
          }
        }
    }
    // ---- online softmax (per query column) in log2 units, P^T fragments.  st already holds s - m_run.  The reference
    // max is set by the first tile and afterwards only moves when some query's tile max exceeds it by more than
    // RESCALE_THR (then every accumulator of the wave is rescaled once), so P <= 2^RESCALE_THR: harmless in fp16 (fp32
    // accumulation), and the O-wide multiply leaves the steady state.
    constexpr float RESCALE_THR = 6.0f;
    float mx[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float m0_ = fmaxf(fmaxf(st[0][qt][0], st[0][qt][1]), fmaxf(st[0][qt][2], st[0][qt][3]));
#pragma unroll
      for (int kt = 1; kt < 4; ++kt) {
        m0_ = fmaxf(fmaxf(m0_, st[kt][qt][0]), st[kt][qt][1]);
        m0_ = fmaxf(fmaxf(m0_, st[kt][qt][2]), st[kt][qt][3]);
      }
      m0_ = max_over_lane_groups(m0_);
      mx[qt] = m0_;
    }
    if (t == 0 || __any((mx[0] > RESCALE_THR) || (mx[1] > RESCALE_THR))) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        // first tile: adopt its max whatever the sign (O is still zero, nothing to rescale); later: raise only
        float delta = mx[qt] == -INFINITY ? 0.f : (t == 0 ? mx[qt] : fmaxf(mx[qt], 0.f));
        m_run[qt] += delta;
        if (t != 0) {
          float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) ot[dt][qt] *= alpha;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int e = 0; e < 4; ++e) st[kt][qt][e] -= delta;
      }
    }
    h8 pf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[kt >> 1][qt][(kt & 1) * 4 + e] = f2e<BF>(__builtin_amdgcn_exp2f(st[kt][qt][e]));
    // ---- O^T += V^T P^T
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        const half_t* va = vs_ + (32 * kc + 8 * lg + (lr >> 2)) * VS + dt * 16 + 4 * (lr & 3);
        s4v v0 = lds_tr16(va), v1 = lds_tr16(va + 4 * VS);
        union { struct { s4v a, b; } s; h8 h; } u;
        u.s.a = v0; u.s.b = v1;
        ot[dt][0] = mfma16<BF>(u.h, pf[kc][0], ot[dt][0]);
        ot[dt][1] = mfma16<BF>(u.h, pf[kc][1], ot[dt][1]);
      }
    }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane holds O[query lr][d = 16 dt + 4 lg + {0..3}]
  half_t* ob = p.o + b * p.o_sb + h * p.o_sh;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    // row sum = O^T row DV-1 (the ones column of V): last accumulator tile, lane group 3, register 3
    float l = __shfl(ot[NDT - 1][qt][3], lr + 48, 64);
    float inv = l > 0.f ? 1.0f / l : 0.f;
    int qi = qblk + qt * 16 + lr;
    if (qi < p.Tq) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        int d = dt * 16 + lg * 4;
        if (d < p.HS) {
          h4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2e<BF>(ot[dt][qt][e] * inv);
          *reinterpret_cast<h4*>(ob + qi * p.o_st + d) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant for the head sizes of the SD UNets (40 / 80 / 160, plus 64 / 128).  In k_sdpa above a third of the
// time at d = 40 goes to moving K/V tiles global -> VGPR -> LDS (buffer loads, ds_writes, the vmcnt stall in front of
// them).  Here the tiles go global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, no
// registers) into a ring of S stages, S-1 tiles in flight, one s_barrier per tile behind a counted vmcnt.
//   K image: row r = 16 kt + lr holds key 32 (kt>>1) + 8 (lr>>2) + 4 (kt&1) + (lr&3) (the key each MFMA A-row needs, so
//            a 16-lane read walks 16 consecutive rows); pitch = an odd number of 16-B chunks -> conflict-free b128 reads.
//   V image: natural key order, pitch chosen so the 4 rows x 32 B of a transposed read fall in distinct banks.
//   Pad chunks are fetched "out of range" (zeros).  Columns >= HS read by the 32-deep QK^T steps hold finite data of
//   the neighbouring row and meet zero Q columns; the LDS is zeroed once so no stale NaN pattern can be there.
//   Row sums: one extra MFMA per (key half, query tile) against an all-ones A fragment.
typedef int i4v __attribute__((ext_vector_type(4)));

// raw buffer descriptor in SGPRs (base, no stride, num_records in bytes, 32-bit raw format word of gfx9)
__device__ __forceinline__ i4v sdpa_rsrc(const void* base, unsigned bytes) {
  unsigned long long a = (unsigned long long)base;
  i4v r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

// LDS-DMA issued from inline asm: the compiler does not see an LDS write, so it puts no vmcnt(0) in front of the
// ds_read_tr of the tile being consumed (it cannot tell the ring stages apart); ordering is the kernel's own
// counted s_waitcnt vmcnt + s_barrier.  16 B per lane, LDS destination = M0 + lane * 16, out-of-range lanes write 0.
__device__ __forceinline__ void sdpa_dma16(i4v rsrc, unsigned voffset_bytes, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane((int)lds_base)), "v"(voffset_bytes), "s"(rsrc) : "memory");   // M0 has no other user in k_sdpa_dma (gfx9 DS ops do not read it)
}

// LDS layouts are chosen against the real bank model of the chip (64 banks x 4 B; a ds_read_b128 is served in four groups of
// 16 NON-contiguous lanes, a ds_read_b64_tr_b16 in two groups of 32; tools/lds_conflicts.py enumerates the costs):
//   K rows (16 consecutive rows per 16-lane read, 4 chunk columns across the lane groups): conflict-free exactly when the
//     pitch in 16-B chunks is 2 (mod 4): 6 / 10 / 10 / 18 / 22 chunks for d = 40 / 64 / 80 / 128 / 160 (an odd pitch costs 2x);
//   V rows (keys k and k + 8 of a 32-key half land in the same 32-lane group): no plain pitch is conflict-free; a skew of a
//     few chunks after every 8 rows is: (pitch, skew) = (6, 8) / (10, 8) / (10, 8) / (18, 8) / (20, 2) chunks.
constexpr int sdpa_k_pitch(int ck) { return ck + (2 - ck % 4 + 4) % 4; }
constexpr int sdpa_v_pitch(int ck) { return ck <= 6 ? 6 : ck <= 10 ? 10 : ck <= 18 ? 18 : ck <= 20 ? 20 : 22; }
constexpr int sdpa_v_skew(int ck) { return ck <= 18 ? 8 : ck <= 20 ? 2 : 8; }

// ring depth: SdpaDma<HS>::S (<= 4) for both block sizes.  Six stages for the 8-wave form (two blocks per CU leave 80 KiB each) were slower:
// 64 x 64 d40 81.8 -> 86.9 us, 96 x 96 (B 8) 1116 -> 1162 us -- the tile fetch is not latency-starved at depth 4, and the prologue zero-fills the ring
constexpr int sdpa_ring(int stage_b, int nw, int s4) { return s4; }
template <int HS>
struct SdpaDma {
  static constexpr int DQK = (HS + 31) / 32 * 32, NKS = DQK / 32, NDT = (HS + 15) / 16, CK = HS / 8;
  // HAS_PAD: the P.V tiles have spare columns (HS % 16 != 0, d = 40): column HS of the V image is preset to 1 in LDS and kept out of the
  // DMA (EXEC-masked lanes write nothing), so the row sums come out of the P.V MFMAs themselves and the ones-MFMA (4 of 32 per tile) goes
  static constexpr bool HAS_PAD = (HS % 16) != 0;
  static constexpr int KPC = sdpa_k_pitch(CK), VPC = sdpa_v_pitch(CK), VSC = sdpa_v_skew(CK);   // pitches / skew in 16-B chunks
  static constexpr int KP = KPC * 8, VP = VPC * 8, VSK = VSC * 8; // pitches / skew in halves
  static constexpr int VGC = 8 * VPC + VSC;                        // chunks per group of 8 V rows (rows + skew pad)
  static constexpr int NKI = KPC, NVI = (8 * VGC + 63) / 64;       // 1-KiB pieces of the K / V image (64 rows each)
  static constexpr int K_BYTES = NKI * 1024, STAGE_B = (NKI + NVI) * 1024;
  static constexpr int S = (144 * 1024 / STAGE_B) >= 4 ? 4 : (144 * 1024 / STAGE_B);
  static constexpr int NI = NKI + NVI, LPW = (NI + 3) / 4;        // 1-KiB pieces per tile / per wave
  static_assert(S >= 2, "ring needs two stages");
};

// NW = waves per block (4, or 8 for long sequences: twice the queries per fetched K/V tile -- the tile fetch, not the barrier, is what the
// ablation prices at 19 % of the d = 40 loop at 9216 tokens -- and four waves per SIMD instead of three at two blocks per CU)
template <int HS, int QT, int DBG = 0, int NW = 4, bool BF = false>        // DBG: compile-time ablation mask (tools/sdpa_dbg.py; bits as SdpaP::dbg); BF: bfloat16 q / k / v / o (sdpa_bf16.hip)
__global__ void __launch_bounds__(NW * 64) k_sdpa_dma(const SdpaP p) {
  static_assert(!(BF && DBG), "the ablation instances are fp16");
  constexpr int QW = 16 * QT, QB = NW * QW, NT = NW * 64;   // queries per wave / per block, threads
  using C = SdpaDma<HS>;
  constexpr int NKS = C::NKS, NDT = C::NDT, CK = C::CK, KPC = C::KPC, VPC = C::VPC, KP = C::KP, VP = C::VP, VSK = C::VSK, VGC = C::VGC;
  constexpr int STAGE_B = C::STAGE_B, NI = C::NI, LPW = (C::NI + NW - 1) / NW;
  constexpr int S = sdpa_ring(STAGE_B, NW, C::S);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const SdpaBlk blk = sdpa_block(p, QB);
  const int b = blk.b, h = blk.h;
  const int qblk = blk.qb * QB + wid * QW;
  const half_t* qb = p.q + b * p.q_sb + h * p.q_sh;
  const half_t* kb = p.k + b * p.k_sb + h * p.k_sh;
  const half_t* vb = p.v + b * p.v_sb + h * p.v_sh;

  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  for (int i = tid; i < S * STAGE_B / 16; i += NT) reinterpret_cast<u4*>(smem_raw)[i] = (u4){0, 0, 0, 0};
  constexpr bool HAS_PAD = C::HAS_PAD;
  if constexpr (HAS_PAD) {
    __syncthreads();                     // ones column of V (column HS of every key row, every ring stage): written once
    for (int i = tid; i < S * 64; i += NT) {
      int st_ = i >> 6, R = i & 63;
      reinterpret_cast<half_t*>(smem_raw + st_ * STAGE_B + C::K_BYTES)[(R >> 3) * VGC * 8 + (R & 7) * VP + HS] = f2e<BF>(1.0f);
    }
  }

  h8 qf[QT][NKS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int qi = qblk + qt * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      int d0 = ks * 32 + lg * 8;
      h8 qv = (qi < p.Tq && d0 < HS) ? *reinterpret_cast<const h8*>(qb + qi * p.q_st + d0) : (h8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 8; ++j) qv[j] = f2e<BF>(e2f<BF>(qv[j]) * p.scale_log2e);
      qf[qt][ks] = qv;
    }
  }

  int ntiles = (p.Tk + 63) / 64;
  if (p.causal) {
    int last_q = min(p.Tq, blk.qb * QB + QB) - 1;
    ntiles = min(ntiles, last_q / 64 + 1);
  }

  // this wave's 1-KiB pieces of a tile: piece j = wid + 4 i (clamped: the spare slots of the last round repeat piece
  // NI-1, same bytes to the same place); j < KPC -> K image, else V image
  const i4v rs_k = sdpa_rsrc(kb, (unsigned)(((long long)(p.Tk - 1) * p.k_st + HS) * 2));
  const i4v rs_v = sdpa_rsrc(vb, (unsigned)(((long long)(p.Tk - 1) * p.v_st + HS) * 2));
  unsigned voff[LPW];
  const unsigned k_adv = 64u * (unsigned)p.k_st * 2u, v_adv = 64u * (unsigned)p.v_st * 2u;
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    int j = min(wid + NW * i, NI - 1);
    if (j < KPC) {
      int x = 64 * j + lane, r = x / KPC, cc = x - r * KPC;
      int kt = r >> 4, rr = r & 15;
      int key = 32 * (kt >> 1) + 8 * (rr >> 2) + 4 * (kt & 1) + (rr & 3);
      voff[i] = cc < CK ? (unsigned)(key * (int)p.k_st + cc * 8) * 2u : 0x80000000u;
    } else {
      // V image: groups of 8 rows (VPC chunks each) followed by VSC skew chunks; pad / skew / tail chunks are fetched out of range
      int x = 64 * (j - KPC) + lane, grp = x / VGC, rem = x - grp * VGC;
      int rr = rem / VPC, cc = rem - rr * VPC, r = 8 * grp + rr;
      voff[i] = (rr < 8 && grp < 8 && cc < CK) ? (unsigned)(r * (int)p.v_st + cc * 8) * 2u : 0x80000000u;
      if (HAS_PAD && rr < 8 && grp < 8 && cc == CK) voff[i] = 0xFFFFFFFFu;     // the preset ones column: this lane stays out of the DMA
    }
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
  auto issue = [&](int tt) {
    const unsigned base = lds0 + (unsigned)(tt % S) * STAGE_B;
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      int j = min(wid + NW * i, NI - 1);
      if (j < KPC) sdpa_dma16(rs_k, voff[i] + (unsigned)tt * k_adv, base + j * 1024);
      else if (!HAS_PAD) sdpa_dma16(rs_v, voff[i] + (unsigned)tt * v_adv, base + j * 1024);
      else if (voff[i] != 0xFFFFFFFFu) sdpa_dma16(rs_v, voff[i] + (unsigned)tt * v_adv, base + j * 1024);   // (EXEC-masked: the skipped lanes write nothing)
    }
  };

  f4 ot[NDT][QT], lt[QT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) { for (int q_ = 0; q_ < QT; ++q_) ot[dt][q_] = (f4){0, 0, 0, 0}; }
  for (int q_ = 0; q_ < QT; ++q_) lt[q_] = (f4){0, 0, 0, 0};
  float m_run[QT];
  for (int q_ = 0; q_ < QT; ++q_) m_run[q_] = 0.f;
  const half_t one1 = f2e<BF>(1.f);
  const h8 ones = {one1, one1, one1, one1, one1, one1, one1, one1};

  __syncthreads();                       // zero fill done (and drained) before the first DMA lands
#pragma unroll
  for (int tt = 0; tt < S - 1; ++tt)
    if (tt < ntiles) issue(tt);

  for (int t = 0; t < ntiles; ++t) {
    // tile t landed (this wave's pieces), leaving the younger tiles in flight; the barrier extends that to every wave
    // and tells that all of them are done reading tile t-1, whose slot the next issue refills
    {
      int younger = min(S - 2, ntiles - 1 - t);
      if (younger >= 4 && S >= 6 && 4 * LPW <= 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPW > 63 ? 63 : 4 * LPW) : "memory");
      else if (younger >= 3 && S >= 5 && 3 * LPW <= 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPW > 63 ? 63 : 3 * LPW) : "memory");
      else if (younger >= 2 && S >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW > 63 ? 63 : 2 * LPW) : "memory");
      else if (younger >= 1 && S >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW > 63 ? 63 : LPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!(DBG & 8)) __builtin_amdgcn_s_barrier();
    if (t + S - 1 < ntiles && !(DBG & 16)) issue(t + S - 1);

    const half_t* ks_ = reinterpret_cast<const half_t*>(smem_raw + (t % S) * STAGE_B);
    const half_t* vs_ = ks_ + C::K_BYTES / 2;

    f4 st[4][QT];
    f4 init4[QT];
#pragma unroll
    for (int q_ = 0; q_ < QT; ++q_) init4[q_] = (f4){-m_run[q_], -m_run[q_], -m_run[q_], -m_run[q_]};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        h8 kf = *reinterpret_cast<const h8*>(ks_ + (16 * kt + lr) * KP + ks * 32 + lg * 8);
#pragma unroll
        for (int q_ = 0; q_ < QT; ++q_) {
          if (DBG & 4) { if (ks == 0) st[kt][q_] = init4[q_] + (f4){(float)kf[0], (float)kf[1], (float)kf[2], (float)kf[3]}; }
          else st[kt][q_] = mfma16<BF>(kf, qf[q_][ks], ks == 0 ? init4[q_] : st[kt][q_]);
        }
      }
    }
    const int kbase = t * 64 + 8 * lg;
    if (t * 64 + 64 > p.Tk || p.causal) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int key = kbase + 32 * (kt >> 1) + 4 * (kt & 1) + e;
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            int qi = qblk + qt * 16 + lr;
            if (key >= p.Tk || (p.causal && key > qi)) st[kt][qt][e] = -INFINITY;
          }
        }
    }
    constexpr float RESCALE_THR = 6.0f;
    float mx[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (DBG & 64) { mx[qt] = st[0][qt][0]; continue; }
      float m0_ = fmaxf(fmaxf(st[0][qt][0], st[0][qt][1]), fmaxf(st[0][qt][2], st[0][qt][3]));
#pragma unroll
      for (int kt = 1; kt < 4; ++kt) {
        m0_ = fmaxf(fmaxf(m0_, st[kt][qt][0]), st[kt][qt][1]);
        m0_ = fmaxf(fmaxf(m0_, st[kt][qt][2]), st[kt][qt][3]);
      }
      m0_ = max_over_lane_groups(m0_);
      mx[qt] = m0_;
    }
    bool over = false;
#pragma unroll
    for (int q_ = 0; q_ < QT; ++q_) over = over || (mx[q_] > RESCALE_THR);
    if (t == 0 || __any(over)) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float delta = mx[qt] == -INFINITY ? 0.f : (t == 0 ? mx[qt] : fmaxf(mx[qt], 0.f));
        m_run[qt] += delta;
        if (t != 0) {
          float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) ot[dt][qt] *= alpha;
          lt[qt] *= alpha;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int e = 0; e < 4; ++e) st[kt][qt][e] -= delta;
      }
    }
    h8 pf[2][QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[kt >> 1][qt][(kt & 1) * 4 + e] = (DBG & 1) ? (half_t)st[kt][qt][e] : f2e<BF>(__builtin_amdgcn_exp2f(st[kt][qt][e]));
    if constexpr (!HAS_PAD) {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
        for (int q_ = 0; q_ < QT; ++q_) lt[q_] = mfma16<BF>(ones, pf[kc][q_], lt[q_]);
      }
    }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        const half_t* va = vs_ + (32 * kc + 8 * lg + (lr >> 2)) * VP + (4 * kc + lg) * VSK + dt * 16 + 4 * (lr & 3);
        union { struct { s4v a, b; } s; h8 h; } u;
        if (DBG & 32) u.h = pf[kc][0];
        else { u.s.a = lds_tr16(va); u.s.b = lds_tr16(va + 4 * VP); }
#pragma unroll
        for (int q_ = 0; q_ < QT; ++q_) {
          if (DBG & 2) ot[dt][q_] += (f4){(float)u.h[0] * (float)pf[kc][q_][0], (float)u.h[1], (float)u.h[2], (float)pf[kc][q_][7]};
          else ot[dt][q_] = mfma16<BF>(u.h, pf[kc][q_], ot[dt][q_]);
        }
      }
    }
  }

  half_t* ob = p.o + b * p.o_sb + h * p.o_sh;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    // row sum: the ones-MFMA's accumulator, or row HS of O^T (the preset ones column of V): accumulator tile HS / 16, lane group (HS % 16) / 4
    float l = lt[qt][0];
    if constexpr (HAS_PAD) l = __shfl(ot[(HS / 16) % NDT][qt][0], lr + 16 * ((HS % 16) / 4), 64);
    float inv = l > 0.f ? 1.0f / l : 0.f;
    int qi = qblk + qt * 16 + lr;
    if (qi < p.Tq) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        int d = dt * 16 + lg * 4;
        if (d < HS) {
          h4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = f2e<BF>(ot[dt][qt][e] * inv);
          *reinterpret_cast<h4*>(ob + qi * p.o_st + d) = o;
        }
      }
    }
  }
}

template <int HS, int QT, int NW = 4>
static int launch_sdpa_dma(const SdpaP& p, hipStream_t st) {
  constexpr int smem = sdpa_ring(SdpaDma<HS>::STAGE_B, NW, SdpaDma<HS>::S) * SdpaDma<HS>::STAGE_B;
  constexpr int QB = 16 * NW * QT;
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_sdpa_dma<HS, QT, 0, NW, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
#if defined(TF_ABLATION) && !TF_TU_BF   // the ablation library only (python -m tinyfusers_amd.build --ablation; tools/sdpa_dbg.py): wrong results by design
  if constexpr (HS == 40 && QT == 2 && NW == 4) {
    if (p.dbg) {
      const dim3 grid((unsigned)((p.Tq + QB - 1) / QB * p.NH * p.B));
      switch (p.dbg) {
#define TF_SDPA_DBG_CASE(M) case M: hipLaunchKernelGGL((k_sdpa_dma<HS, QT, M>), grid, dim3(256), smem, st, p); break;
        TF_SDPA_DBG_CASE(1) TF_SDPA_DBG_CASE(64) TF_SDPA_DBG_CASE(65) TF_SDPA_DBG_CASE(2) TF_SDPA_DBG_CASE(4) TF_SDPA_DBG_CASE(6) TF_SDPA_DBG_CASE(32)
        TF_SDPA_DBG_CASE(34) TF_SDPA_DBG_CASE(8) TF_SDPA_DBG_CASE(16) TF_SDPA_DBG_CASE(24) TF_SDPA_DBG_CASE(62) TF_SDPA_DBG_CASE(89) TF_SDPA_DBG_CASE(128)
#undef TF_SDPA_DBG_CASE
        default: tf_set_error("tf_sdpa_f16: no ablation build for TF_SDPA_DBG=%d", p.dbg); return TF_E_UNSUPPORTED;
      }
      TF_LAUNCH_CHECK();
      return TF_OK;
    }
  }
#endif
  hipLaunchKernelGGL((k_sdpa_dma<HS, QT, 0, NW, kBF>), dim3((unsigned)((p.Tq + QB - 1) / QB * p.NH * p.B)), dim3(NW * 64), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

template <int DQK, int DV>
static int launch_sdpa(const SdpaP& p, hipStream_t st) {
  constexpr int smem = 2 * (64 * (DQK + 8) + 64 * (DV + 8)) * 2;
  static bool attr_set = false;
  if (!attr_set) {
    TF_HIP(hipFuncSetAttribute((const void*)k_sdpa<DQK, DV, kBF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_sdpa<DQK, DV, kBF>), dim3((unsigned)((p.Tq + 127) / 128 * p.NH * p.B)), dim3(256), smem, st, p);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

static bool g_sdpa_generic = getenv("TF_SDPA_GENERIC") != nullptr;
#ifdef TF_ABLATION
static int g_sdpa_dbg = getenv("TF_SDPA_DBG") ? atoi(getenv("TF_SDPA_DBG")) : 0;   // ablation of k_sdpa_dma<40, 2> (see SdpaP::dbg)
#else
static const int g_sdpa_dbg = 0;   // the shipped library ignores TF_SDPA_DBG: it holds no kernel that returns wrong results
#endif
static int g_sdpa_nw = getenv("TF_SDPA_NW") ? atoi(getenv("TF_SDPA_NW")) : 0;      // A/B: 4 / 8 waves per block where both exist (0 = per-shape choice)
static int g_sdpa_qt = getenv("TF_SDPA_QT") ? atoi(getenv("TF_SDPA_QT")) : 0;   // debugging: force the register-staged kernel

// one body for both element types: this unit's kernels (kBF) behind tf_sdpa_f16 here, behind tfk_sdpa_bf16 in sdpa_bf16.hip (#define TF_TU_BF 1 + #include of this file)
int tfk_sdpa_bf16(void* o, const void* q, const void* k, const void* v, int B, int NH, int Tq, int Tk, int HS, long long q_sb, long long q_sh, long long q_st, long long k_sb,
                  long long k_sh, long long k_st, long long v_sb, long long v_sh, long long v_st, long long o_sb, long long o_sh, long long o_st, int causal, tfStream_t s);
#if TF_TU_BF
int tfk_sdpa_bf16(
#else
extern "C" int tf_sdpa_16(int dtype, void* o, const void* q, const void* k, const void* v, int B, int NH, int Tq, int Tk, int HS, long long q_sb, long long q_sh, long long q_st,
                          long long k_sb, long long k_sh, long long k_st, long long v_sb, long long v_sh, long long v_st, long long o_sb, long long o_sh, long long o_st, int causal,
                          tfStream_t s) {
  if (dtype == TF_DTYPE_F16) return tf_sdpa_f16(o, q, k, v, B, NH, Tq, Tk, HS, q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st, causal, s);
  TF_REQUIRE(dtype == TF_DTYPE_BF16, "tf_sdpa_16: dtype=%d (0 = float16, 1 = bfloat16)", dtype);
  return tfk_sdpa_bf16(o, q, k, v, B, NH, Tq, Tk, HS, q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st, causal, s);
}
extern "C" int tf_sdpa_f16(
#endif
                           void* o, const void* q, const void* k, const void* v, int B, int NH, int Tq, int Tk, int HS, long long q_sb,
                           long long q_sh, long long q_st, long long k_sb, long long k_sh, long long k_st, long long v_sb, long long v_sh,
                           long long v_st, long long o_sb, long long o_sh, long long o_st, int causal, tfStream_t s) {
  TF_REQUIRE(o && q && k && v, "tf_sdpa_f16: null tensor");
  TF_REQUIRE(B >= 0 && NH >= 1 && Tq >= 0 && Tk >= 1, "tf_sdpa_f16: bad sizes B=%d NH=%d Tq=%d Tk=%d", B, NH, Tq, Tk);
  TF_REQUIRE(HS >= 8 && HS % 8 == 0 && HS <= 160, "tf_sdpa_f16: head size %d must be a multiple of 8 in [8, 160]", HS);
  TF_REQUIRE((long long)((Tq + 63) / 64) * NH * B < (1ll << 31), "tf_sdpa_f16: too many (query block, head, batch) work items");
  const long long str[] = {q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st};
  for (int i = 0; i < 9; ++i) TF_REQUIRE(str[i] % 8 == 0, "tf_sdpa_f16: q/k/v strides must be multiples of 8 elements (16-B rows)");
  TF_REQUIRE(o_sb % 4 == 0 && o_sh % 4 == 0 && o_st % 4 == 0, "tf_sdpa_f16: output strides must be multiples of 4 elements");
  if (B == 0 || Tq == 0) return TF_OK;
  SdpaP p;
  p.q = (const half_t*)q; p.k = (const half_t*)k; p.v = (const half_t*)v; p.o = (half_t*)o;
  p.B = B; p.NH = NH; p.Tq = Tq; p.Tk = Tk; p.HS = HS;
  p.q_sb = q_sb; p.q_sh = q_sh; p.q_st = q_st; p.k_sb = k_sb; p.k_sh = k_sh; p.k_st = k_st;
  p.v_sb = v_sb; p.v_sh = v_sh; p.v_st = v_st; p.o_sb = o_sb; p.o_sh = o_sh; p.o_st = o_st;
  p.scale_log2e = (1.0f / sqrtf((float)HS)) * 1.4426950408889634f;
  p.causal = causal;
  p.dbg = g_sdpa_dbg;
  hipStream_t st = tf_hs(s);
  TfProfScope prof_(TF_PROF_FAM_SDPA, 4.0 * B * NH * (double)Tq * Tk * HS, st);      // (FLOPs = 4 B NH Tq Tk d)
  // K/V offsets inside a (batch, head) slice must fit the 32-bit buffer offsets of the DMA kernels
  const bool small = ((long long)Tk * k_st + HS) * 2 < (1ll << 31) && ((long long)Tk * v_st + HS) * 2 < (1ll << 31);
  if (small && !g_sdpa_generic) {
    // 16 queries per wave (QT = 1) doubles the waves in flight; g_sdpa_qt: 0 = per-shape choice, 1 / 2 forced (TF_SDPA_QT)
    // (measured: 32x32 d80 22.5 -> 19.7 us, 16x16 d160 13.4 -> 10.6 us with 16-query waves; 64x64 d40 93.8 -> 118.7 us: only when
    // the 32-query grid would leave CUs without a block)
    const long long blocks2 = (long long)((Tq + 127) / 128) * NH * B;
    const bool narrow = g_sdpa_qt ? g_sdpa_qt == 1 : blocks2 < 256;
    if (narrow) {
      if (HS == 40) return launch_sdpa_dma<40, 1>(p, st);
      if (HS == 64) return launch_sdpa_dma<64, 1>(p, st);
      if (HS == 80) return launch_sdpa_dma<80, 1>(p, st);
      if (HS == 128) return launch_sdpa_dma<128, 1>(p, st);
      if (HS == 160) return launch_sdpa_dma<160, 1>(p, st);
    }
    // eight waves per block (256 queries per K/V tile fetch) once that still gives every CU a block (measured: 64 x 64 d40, B 2: 86.9 -> 81.0 us
    // with ONE eight-wave block per CU; 96 x 96, B 8: 1214 -> 1103 us)
    const long long blocks8 = (long long)((Tq + 255) / 256) * NH * B;
    const bool wide8 = g_sdpa_nw ? g_sdpa_nw == 8 : blocks8 >= 256;
    if (HS == 40 && wide8) return launch_sdpa_dma<40, 2, 8>(p, st);
    if (HS == 80 && wide8) return launch_sdpa_dma<80, 2, 8>(p, st);
    if (HS == 40) return launch_sdpa_dma<40, 2>(p, st);
    if (HS == 64) return launch_sdpa_dma<64, 2>(p, st);
    if (HS == 80) return launch_sdpa_dma<80, 2>(p, st);
    if (HS == 128) return launch_sdpa_dma<128, 2>(p, st);
    if (HS == 160) return launch_sdpa_dma<160, 2>(p, st);
  }
  // (DQK, DV) = (HS rounded up to 32, HS + 1 rounded up to 16): the V tile always has room for the ones column
  if (HS <= 32) return launch_sdpa<32, 48>(p, st);
  if (HS <= 40) return launch_sdpa<64, 48>(p, st);
  if (HS <= 56) return launch_sdpa<64, 64>(p, st);
  if (HS <= 64) return launch_sdpa<64, 80>(p, st);
  if (HS <= 88) return launch_sdpa<96, 96>(p, st);
  if (HS <= 96) return launch_sdpa<96, 112>(p, st);
  if (HS <= 120) return launch_sdpa<128, 128>(p, st);
  if (HS <= 128) return launch_sdpa<128, 144>(p, st);
  return launch_sdpa<160, 176>(p, st);
}
