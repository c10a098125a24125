// Shared helpers for the gfx950 kernels behind libtinyfusers_hip.so (C-ABI in include/tinyfusers_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 half_t;
typedef __bf16 bf16_t;     // bfloat16 elements of the tf_*_bf16 entries
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32;

// ---- error plumbing: every exported function returns int (0 = ok, else hipError_t or TF_E_*),
// mirroring the reference's "C status -> RuntimeError" convention (storage/device.py:33-37).
enum { TF_OK = 0, TF_E_ARG = 10001, TF_E_UNSUPPORTED = 10002, TF_E_WORKSPACE = 10003, TF_E_STATE = 10004 };
void tf_set_error(const char* fmt, ...);
#define TF_HIP(expr)                                                                  \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      tf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return (int)_e;                                                                 \
    }                                                                                 \
  } while (0)
#define TF_REQUIRE(cond, ...)                  \
  do {                                         \
    if (!(cond)) {                             \
      tf_set_error(__VA_ARGS__);               \
      return TF_E_ARG;                         \
    }                                          \
  } while (0)
#define TF_LAUNCH_CHECK() TF_HIP(hipGetLastError())

struct tfStream_st { hipStream_t s; };
struct tfEvent_st { hipEvent_t e; };
struct tfGraph_st { hipGraph_t g; hipGraphExec_t x; };
static inline hipStream_t tf_hs(struct tfStream_st* s) { return s ? s->s : (hipStream_t)0; }

// ---- per-launch profiling of kernel FAMILIES with HIP events on the launch stream (runtime.hip; bench.py's roofline legs).  While
// tf_prof_enable(1) is on, a launcher brackets its kernel with `TfProfScope scope(family, work, stream);` -- work = the launch's algorithmic
// HBM bytes (norm / reduce families) or FLOPs (SDPA) -- and tf_prof_read_family hands back accumulated milliseconds, work and launches.
enum { TF_PROF_FAM_GROUP_NORM = 1, TF_PROF_FAM_SPLITK_REDUCE = 2, TF_PROF_FAM_LAYER_NORM = 3, TF_PROF_FAM_SDPA = 4, TF_PROF_NFAM = 5 };
extern bool g_tf_prof;
extern float g_tf_prof_overhead_ms;     // what an event bracket reads beyond the kernel's own duration (measured by tf_prof_enable, csrc/gemm.hip)
void tf_prof_fam_reset();
void tf_prof_fam_begin(int family, double work, hipStream_t st);
void tf_prof_fam_end(hipStream_t st);
void tf_prof_fam_add(int family, double work, double ms);          // a bracket the caller timed itself (the split-K reduce inside run_gemm)
struct TfProfScope {
  hipStream_t st; bool on;
  TfProfScope(int family, double work, hipStream_t s) : st(s), on(g_tf_prof) { if (on) tf_prof_fam_begin(family, work, s); }
  ~TfProfScope() { if (on) tf_prof_fam_end(st); }
};

// device-side helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// fast activations: v_exp_f32 / v_rcp_f32 (1 ulp) instead of the IEEE division sequence; inputs are fp16-range
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// tanh-GELU of storage/tensor.py:80-82: 0.5 x (1 + tanh(0.7978845608 x (1 + 0.044715 x^2)))
__device__ __forceinline__ float gelu_f(float x) {
  float u = 0.7978845608f * x * (1.0f + 0.044715f * x * x);
  // 0.5 (1 + tanh(u)) = sigmoid(2u)
  return x * sigmoid_f(2.0f * u);
}
// 8 floats -> 8 OCP e4m3 bytes (saturating at +-448: v_cvt_pk_fp8_f32 itself does not clamp)
__device__ __forceinline__ uint2 pack8_fp8(f4 v0, f4 v1) {
  auto cl = [](float f) { return fminf(fmaxf(f, -448.0f), 448.0f); };
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(cl(v0[0]), cl(v0[1]), lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(cl(v0[2]), cl(v0[3]), lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(cl(v1[0]), cl(v1[1]), hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(cl(v1[2]), cl(v1[3]), hi, true);
  return make_uint2((unsigned)lo, (unsigned)hi);
}

// ---- block-scaled e4m3 ("MX") activation tensors: what feeds v_mfma_scale_f32_16x16x128_f8f6f4 on the activation side.  One buffer per
// tensor: rows x C e4m3 codes (row = pixel / token, C % 32 == 0), followed by rows x C/32 E8M0 scale bytes -- every 32 consecutive channels
// of a row share one power-of-two scale 2^e with amax / 2^e <= 448, e = ceil(log2(amax / 448)); code = e4m3(x / 2^e), round to nearest even
// (no saturation can occur); a block of zeros has byte 0 (2^-127).  oracle/fp8.py::quant_act_mx restates it.
//   mx_quant8: this lane's 8 consecutive channels; the lanes l ^ 1, l ^ 2 hold the other 24 of the block (EVERY lane of the wave must call).
__device__ __forceinline__ uint2 mx_quant8(f4 v0, f4 v1, unsigned& scale_byte) {
  float a = fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))), fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
  a = fmaxf(a, __shfl_xor(a, 1, 64));
  a = fmaxf(a, __shfl_xor(a, 2, 64));
  const unsigned bits = __float_as_uint(a / 448.0f);      // (IEEE division: the same quotient as the oracle's)
  const unsigned E = bits >> 23, M = bits & 0x7fffffu;    // a >= 0: no sign bit
  unsigned byte = E == 0 ? 0u : (M ? E + 1u : E);         // e = ceil(log2(a / 448)), biased by 127; zero / subnormal quotient -> 2^-127
  byte = byte > 254u ? 254u : byte;
  scale_byte = byte;
  const int ne = 127 - (int)byte;                         // x * 2^(-e)
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(v0[0], ne), ldexpf(v0[1], ne), lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(v0[2], ne), ldexpf(v0[3], ne), lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(v1[0], ne), ldexpf(v1[1], ne), hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(v1[2], ne), ldexpf(v1[3], ne), hi, true);
  return make_uint2((unsigned)lo, (unsigned)hi);
}

typedef bf16_t b8v __attribute__((ext_vector_type(8)));   // MFMA operand of the bfloat16 instances (same register image as h8)
// ---- element type of a launch's 16-bit tensors: float16 (BF = false) or bfloat16 (BF = true: GemmP::bf16, round 5 -- every kernel of the family
// has both forms, the bfloat16 instances in translation units of their own, gemm_k_*_bf16.hip).  Registers and LDS hold the elements in h8 / h4 /
// half_t CONTAINERS either way (the loaders move bytes); only these helpers look inside.
template <bool BF> __device__ __forceinline__ f4 mfma16(h8 a, h8 b, f4 c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <bool BF> __device__ __forceinline__ float e2f(half_t raw) {
  if constexpr (BF) return (float)__builtin_bit_cast(bf16_t, raw);
  else return (float)raw;
}
template <bool BF> __device__ __forceinline__ half_t f2e(float f) {
  if constexpr (BF) return __builtin_bit_cast(half_t, (bf16_t)f);        // (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN)
  else return (half_t)f;
}
// (sum, sum of squares) of the 8 elements of x added to (s, q): 8 v_dot2 (LayerNorm fold: row statistics from staged / fragment data)
template <bool BF> __device__ __forceinline__ void dot2_stats(h8 x, float& s, float& q) {
  if constexpr (BF) {
    typedef bf16_t bb2 __attribute__((ext_vector_type(2)));
    const b8v xb = __builtin_bit_cast(b8v, x);
    const bb2 one2 = {(bf16_t)1.0f, (bf16_t)1.0f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bb2 v = {xb[2 * e], xb[2 * e + 1]};
      s = __builtin_amdgcn_fdot2_f32_bf16(v, one2, s, false);
      q = __builtin_amdgcn_fdot2_f32_bf16(v, v, q, false);
    }
  } else {
    typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
    const hh2 one2 = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hh2 v = {x[2 * e], x[2 * e + 1]};
      s = __builtin_amdgcn_fdot2(v, one2, s, false);
      q = __builtin_amdgcn_fdot2(v, v, q, false);
    }
  }
}

// The bfloat16 instances live in translation units of their own -- gemm_k_<family>_bf16.hip / sdpa_bf16.hip = `#define TF_TU_BF 1` + `#include` of the fp16
// unit -- so that the fp16 code objects are byte for byte what they were without them: in such a unit kBF is true and every launcher name
// carries the suffix _bf16 (TFK).
#ifndef TF_TU_BF
#define TF_TU_BF 0
#endif
#if TF_TU_BF
#define TFK(name) name##_bf16
#else
#define TFK(name) name
#endif
static constexpr bool kBF = TF_TU_BF != 0;

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }
