// HBM-bound elementwise / layout / sampler kernels (fp16 storage, fp32 math), 16 B per lane.
// Reference semantics: storage/tensor.py:64-86, ff/nn.py:10-12, vision/unet.py:72,81-83,92-97,
// variants/sd.py:14-46, native/cuda/{scale_tensor_func,add_bias_func,transpose,transpose4d,softmax_func}.cu.
#include "common.h"
#include "../../include/tinyfusers_hip.h"

#define EW_BLOCK 256
static inline int ew_grid(long long nvec) {
  long long g = ceil_div_ll(nvec, EW_BLOCK);
  if (g > 256 * 8) g = 256 * 8;  // 8 blocks/CU, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

enum { OP_SILU = 0, OP_SIGMOID = 1, OP_GELU = 2, OP_QGELU = 3 };

template <int OP>
__device__ __forceinline__ float act(float x) {
  if (OP == OP_SILU) return silu_f(x);
  if (OP == OP_SIGMOID) return sigmoid_f(x);
  if (OP == OP_GELU) return gelu_f(x);
  return x * sigmoid_f(1.702f * x);
}

template <int OP, typename T = half_t>      // T: half_t, or bf16_t for the bfloat16 step (config.set_dtype("bf16"))
__global__ void __launch_bounds__(EW_BLOCK) k_unary(T* __restrict__ y, const T* __restrict__ x, long long n) {
  typedef T T8 __attribute__((ext_vector_type(8)));
  long long nvec = n >> 3;
  long long stride = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < nvec; i += stride) {
    T8 v = *reinterpret_cast<const T8*>(x + i * 8), o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)act<OP>((float)v[j]);
    *reinterpret_cast<T8*>(y + i * 8) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    long long i = (nvec << 3) + threadIdx.x;
    y[i] = (T)act<OP>((float)x[i]);
  }
}
// fp16 <-> bfloat16 (the bfloat16 step runs its attention core on the fp16 kernel: one conversion in, one out)
template <typename TO, typename TI>
__global__ void __launch_bounds__(EW_BLOCK) k_convert16(TO* __restrict__ y, const TI* __restrict__ x, long long n) {
  typedef TO O8 __attribute__((ext_vector_type(8)));
  typedef TI I8 __attribute__((ext_vector_type(8)));
  long long nvec = n >> 3, stride = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < nvec; i += stride) {
    I8 v = *reinterpret_cast<const I8*>(x + i * 8);
    O8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (TO)(float)v[j];
    *reinterpret_cast<O8*>(y + i * 8) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) { long long i = (nvec << 3) + threadIdx.x; y[i] = (TO)(float)x[i]; }
}

template <typename T = half_t>
__global__ void __launch_bounds__(EW_BLOCK) k_add(T* __restrict__ y, const T* __restrict__ a, const T* __restrict__ b, long long n) {
  typedef T T8 __attribute__((ext_vector_type(8)));
  long long nvec = n >> 3;
  long long stride = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < nvec; i += stride) {
    T8 u = *reinterpret_cast<const T8*>(a + i * 8), v = *reinterpret_cast<const T8*>(b + i * 8), o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)((float)u[j] + (float)v[j]);
    *reinterpret_cast<T8*>(y + i * 8) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    long long i = (nvec << 3) + threadIdx.x;
    y[i] = (T)((float)a[i] + (float)b[i]);
  }
}

// x (rows, 2C) -> y (rows, C): a * gelu(gate)   (ff/nn.py:10-12)
__global__ void __launch_bounds__(EW_BLOCK) k_geglu(half_t* __restrict__ y, const half_t* __restrict__ x, long long rows, int C) {
  int cv = C >> 3;
  long long nvec = rows * cv;
  long long stride = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < nvec; i += stride) {
    long long r = i / cv;
    int c = (int)(i - r * cv) * 8;
    h8 a = *reinterpret_cast<const h8*>(x + r * 2 * C + c), g = *reinterpret_cast<const h8*>(x + r * 2 * C + C + c), o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)a[j] * gelu_f((float)g[j]));
    *reinterpret_cast<h8*>(y + r * C + c) = o;
  }
}

// y[n,hw,c] = x[n,hw,c] + b[n,c]   (vision/resnet.py:28)
__global__ void __launch_bounds__(EW_BLOCK) k_add_bias_nc(half_t* __restrict__ y, const half_t* __restrict__ x, const half_t* __restrict__ b, int N, long long HW, int C) {
  int cv = C >> 3;
  long long nvec = (long long)N * HW * cv;
  long long stride = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < nvec; i += stride) {
    long long p = i / cv;
    int c = (int)(i - p * cv) * 8;
    int n = (int)(p / HW);
    h8 u = *reinterpret_cast<const h8*>(x + p * C + c), v = *reinterpret_cast<const h8*>(b + (long long)n * C + c), o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)u[j] + (float)v[j]);
    *reinterpret_cast<h8*>(y + p * C + c) = o;
  }
}

// nearest 2x (vision/unet.py:81-83), NHWC
__global__ void __launch_bounds__(EW_BLOCK) k_upsample2x(half_t* __restrict__ y, const half_t* __restrict__ x, int N, int H, int W, int C) {
  int cv = C >> 3;
  long long nvec = (long long)N * (2 * H) * (2 * W) * cv;
  long long stride = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < nvec; i += stride) {
    long long p = i / cv;
    int c = (int)(i - p * cv) * 8;
    int wo = (int)(p % (2 * W));
    long long q = p / (2 * W);
    int ho = (int)(q % (2 * H));
    int n = (int)(q / (2 * H));
    const half_t* src = x + (((long long)n * H + (ho >> 1)) * W + (wo >> 1)) * C + c;
    *reinterpret_cast<h8*>(y + p * C + c) = *reinterpret_cast<const h8*>(src);
  }
}

// channel concat (vision/unet.py:72): y (rows, Ca+Cb) = [a (rows,Ca) | b (rows,Cb)]
__global__ void __launch_bounds__(EW_BLOCK) k_concat(half_t* __restrict__ y, const half_t* __restrict__ a, const half_t* __restrict__ b, long long rows, int Ca, int Cb) {
  int C = Ca + Cb, cv = C >> 3;
  long long nvec = rows * cv;
  long long stride = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < nvec; i += stride) {
    long long r = i / cv;
    int c = (int)(i - r * cv) * 8;
    const half_t* src = c < Ca ? a + r * Ca + c : b + r * Cb + (c - Ca);
    *reinterpret_cast<h8*>(y + r * C + c) = *reinterpret_cast<const h8*>(src);
  }
}

// im2col for tiny channel counts (conv_in, C = 4): y (N*Hb*Wo, Kpad), k = (r*S + s)*C + c, zero padded; output rows
// [ho0, ho0 + Hb) of every image (a row band: the 10000 x 10000 input of the reference's tests/conv2d.py runs in bands)
__global__ void __launch_bounds__(EW_BLOCK) k_im2col(half_t* __restrict__ y, const half_t* __restrict__ x, int N, int H, int W, int C, int R, int S,
                                                     int stride, int pad, int ho0, int Hb, int Wo, int Kpad) {
  long long total = (long long)N * Hb * Wo * Kpad;
  long long gs = (long long)gridDim.x * EW_BLOCK;
  int K = R * S * C;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += gs) {
    int k = (int)(i % Kpad);
    long long m = i / Kpad;
    half_t v = (half_t)0.0f;
    if (k < K) {
      int c = k % C, t = k / C, s = t % S, r = t / S;
      int wo = (int)(m % Wo);
      long long q = m / Wo;
      int ho = ho0 + (int)(q % Hb), n = (int)(q / Hb);
      int hi = ho * stride - pad + r, wi = wo * stride - pad + s;
      if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = x[(((long long)n * H + hi) * W + wi) * C + c];
    }
    y[i] = v;
  }
}

__global__ void __launch_bounds__(EW_BLOCK) k_scale_cast_f32_f16(half_t* __restrict__ y, const float* __restrict__ x, float sc, long long n) {
  long long gs = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += gs) y[i] = (half_t)(x[i] * sc);
}
__global__ void __launch_bounds__(EW_BLOCK) k_cast_f32_f16(half_t* __restrict__ y, const float* __restrict__ x, long long n) {
  long long gs = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += gs) y[i] = (half_t)x[i];
}
__global__ void __launch_bounds__(EW_BLOCK) k_cast_f16_f32(float* __restrict__ y, const half_t* __restrict__ x, long long n) {
  long long gs = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += gs) y[i] = (float)x[i];
}

// (N,C,HW) f32 -> (N,HW,C) f16 through a 32x32 LDS tile (both sides coalesced)
__global__ void __launch_bounds__(256) k_nchw_to_nhwc(half_t* __restrict__ dst, const float* __restrict__ src, int C, int HW) {
  __shared__ float tile[32][33];
  int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, p = p0 + tx;
    tile[j][tx] = (c < C && p < HW) ? src[((long long)n * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int p = p0 + j, c = c0 + tx;
    if (c < C && p < HW) dst[((long long)n * HW + p) * C + c] = (half_t)tile[tx][j];
  }
}
__global__ void __launch_bounds__(256) k_nhwc_to_nchw(float* __restrict__ dst, const half_t* __restrict__ src, int C, int HW) {
  __shared__ float tile[32][33];
  int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    int p = p0 + j, c = c0 + tx;
    tile[j][tx] = (c < C && p < HW) ? (float)src[((long long)n * HW + p) * C + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, p = p0 + tx;
    if (c < C && p < HW) dst[((long long)n * C + c) * HW + p] = tile[tx][j];
  }
}

// f16 <-> f16 re-layout through a 32x32 LDS tile: (N, A, B) -> (N, B, A)
__global__ void __launch_bounds__(256) k_transpose_f16(half_t* __restrict__ dst, const half_t* __restrict__ src, int A, int B) {
  __shared__ half_t tile[32][34];
  int n = blockIdx.z, a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    int a = a0 + j, b = b0 + tx;
    tile[j][tx] = (a < A && b < B) ? src[((long long)n * A + a) * B + b] : (half_t)0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int b = b0 + j, a = a0 + tx;
    if (a < A && b < B) dst[((long long)n * B + b) * A + a] = tile[tx][j];
  }
}

// ---- reference own-runtime kernels (fp32) ------------------------------------------------------
__global__ void __launch_bounds__(EW_BLOCK) k_scale_f32(float* x, float s, long long n) {
  long long gs = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += gs) x[i] *= s;
}
// out is a column-major (BT x OC) cuBLAS result: out[(idx%OC)*BT + idx/OC] += bias[idx%OC]  (add_bias_func.cu:1-9)
__global__ void __launch_bounds__(EW_BLOCK) k_add_bias_cm(float* out, const float* bias, int BT, int OC) {
  long long n = (long long)BT * OC, gs = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += gs) {
    int oc = (int)(i % OC);
    long long bt = i / OC;
    out[(long long)oc * BT + bt] += bias[oc];
  }
}
struct PermArgs { int ndim; int oshape[4]; long long istride_of_o[4]; };
__global__ void __launch_bounds__(EW_BLOCK) k_permute_f32(float* __restrict__ out, const float* __restrict__ inp, PermArgs a, long long n) {
  long long gs = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += gs) {
    long long rem = i, off = 0;
#pragma unroll
    for (int d = 3; d >= 0; --d) {
      if (d < a.ndim) { int idx = (int)(rem % a.oshape[d]); rem /= a.oshape[d]; off += idx * a.istride_of_o[d]; }
    }
    out[i] = inp[off];
  }
}

// numerically stable row softmax (N, C) fp32: one 256-thread block per row, three sweeps over the row (max, sum, normalise; the
// second and third hit L2) like the reference's softmax.cu:24-112, with wave64 shuffles instead of its 32-lane ones.  Own-runtime
// API only (Device.softmax): the UNet's attention never materialises scores (k_sdpa*).
__global__ void __launch_bounds__(256) k_softmax_rows(float* __restrict__ out, const float* __restrict__ inp, int C) {
  __shared__ float red[8];
  const float* x = inp + (long long)blockIdx.x * C;
  float* y = out + (long long)blockIdx.x * C;
  int t = threadIdx.x, w = t >> 6, l = t & 63;
  float m = -INFINITY;
  for (int i = t; i < C; i += 256) m = fmaxf(m, x[i]);
  m = wave_max(m);
  if (l == 0) red[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int i = t; i < C; i += 256) s += __expf(x[i] - m);
  s = wave_sum(s);
  if (l == 0) red[4 + w] = s;
  __syncthreads();
  s = red[4] + red[5] + red[6] + red[7];
  float inv = 1.0f / s;
  for (int i = t; i < C; i += 256) y[i] = __expf(x[i] - m) * inv;
}

// row softmax of fp16 scores for the UNFUSED attention path (attention/sdpa.py:53-77 as written: matmul, + mask, softmax kernel,
// matmul) that serves what the flash kernels do not: arbitrary additive / boolean masks (:67-68) and head sizes beyond 160
// (the single-head d = 512 attention of the VAE's AttnBlock).  y[r, c] = softmax_c(scale * x[r, c] + mask[r % mask_rows, c]);
// columns [C, ldc) of y are zero-filled (the P.V GEMM runs over the padded width).  fp32 math, one block per row.
template <typename TIN>                                  // TIN = half_t (in place on fp16 scores) or float (fp32 scores, rows ldi apart)
__global__ void __launch_bounds__(256) k_softmax_mask_rows_f16(half_t* __restrict__ y, const TIN* __restrict__ x, const float* __restrict__ mask,
                                                               int C, int ldc, int ldi, float scale, long long mask_rows) {
  __shared__ float red[8];
  const long long r = blockIdx.x;
  const TIN* xr = x + r * ldi;
  half_t* yr = y + r * ldc;
  const float* mr = mask ? mask + (r % mask_rows) * C : nullptr;
  int t = threadIdx.x, w = t >> 6, l = t & 63;
  float m = -INFINITY;
  for (int i = t; i < C; i += 256) m = fmaxf(m, scale * (float)xr[i] + (mr ? mr[i] : 0.f));
  m = wave_max(m);
  if (l == 0) red[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int i = t; i < C; i += 256) s += __expf(scale * (float)xr[i] + (mr ? mr[i] : 0.f) - m);
  s = wave_sum(s);
  if (l == 0) red[4 + w] = s;
  __syncthreads();
  s = red[4] + red[5] + red[6] + red[7];
  const float inv = 1.0f / s;
  for (int i = t; i < ldc; i += 256) yr[i] = i < C ? (half_t)(__expf(scale * (float)xr[i] + (mr ? mr[i] : 0.f) - m) * inv) : (half_t)0.f;
}

// ---- sampler pieces ----------------------------------------------------------------------------
__global__ void k_set_params(float* p, float t, float a_t, float a_prev, float g) {
  if (threadIdx.x == 0) { p[0] = t; p[1] = a_t; p[2] = a_prev; p[3] = g; }
}
// the same, and in the same launch a copy of `n16` 16-byte units src -> dst: the cached time-embedding row of this timestep goes into the
// buffer the captured step reads (vision/unet.py:54-56, resnet.py:28 of the reference compute it afresh every step; it depends on t only)
__global__ void __launch_bounds__(256) k_set_params_copy(float* p, float t, float a_t, float a_prev, float g, uint4* __restrict__ dst, const uint4* __restrict__ src, long long n16) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { p[0] = t; p[1] = a_t; p[2] = a_prev; p[3] = g; }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
// out (1, dim) f16 = [cos(t f_i), sin(t f_i)], f_i = exp(-ln(max_period) i / half)   (vision/unet.py:92-97)
template <typename T = half_t>
__global__ void k_timestep_embedding(T* out, const float* params, int dim, float max_period) {
  int half_dim = dim / 2;
  float t = params[0];
  for (int i = threadIdx.x; i < half_dim; i += blockDim.x) {
    float f = expf(-logf(max_period) * (float)i / (float)half_dim);
    float a = t * f;
    out[i] = (T)cosf(a);
    out[half_dim + i] = (T)sinf(a);
  }
}
// latent (B,C,H,W) f32 -> x (2B,H,W,C) f16, both CFG halves the same latent (variants/sd.py:31)
template <typename T = half_t>
__global__ void __launch_bounds__(EW_BLOCK) k_cfg_duplicate(T* __restrict__ x, const float* __restrict__ lat, int B, int C, int HW) {
  long long n = (long long)B * C * HW, gs = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += gs) {
    int c = (int)(i % C);
    long long p = i / C;
    int hw = (int)(p % HW), b = (int)(p / HW);
    T v = (T)lat[((long long)b * C + c) * HW + hw];
    x[i] = v;
    x[n + i] = v;
  }
}
// e = e_u + g (e_c - e_u); pred_x0 = (x - sqrt(1-a_t) e)/sqrt(a_t); x' = sqrt(a_prev) pred_x0 + sqrt(1-a_prev) e
template <typename T = half_t>
__global__ void __launch_bounds__(EW_BLOCK) k_cfg_ddim(float* __restrict__ lat, const T* __restrict__ eps2, const T* __restrict__ eps_c,
                                                       const float* __restrict__ params, int B, int C, int HW) {
  float a_t = params[1], a_prev = params[2], g = params[3];
  float s1 = sqrtf(1.0f - a_t), r = sqrtf(a_t), sp = sqrtf(a_prev), dp = sqrtf(1.0f - a_prev);
  long long n = (long long)B * C * HW, gs = (long long)gridDim.x * EW_BLOCK;
  if (!eps_c) eps_c = eps2 + n;                          // one (2B, ...) tensor [uncond x B ; cond x B], or the two halves apart
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += gs) {
    int hw = (int)(i % HW);
    long long q = i / HW;
    int c = (int)(q % C), b = (int)(q / C);
    long long j = ((long long)b * HW + hw) * C + c;
    float eu = (float)eps2[j], ec = (float)eps_c[j];
    float e = eu + g * (ec - eu);
    float x = lat[i];
    float px0 = (x - s1 * e) / r;
    lat[i] = sp * px0 + dp * e;
  }
}

// ---- embedding gather (+ position rows): out[i, :] = table[ids[i], :] + pos[i % T, :]   (ff/embedding.py:10-24 as intended:
// the reference builds a one-hot matrix on the host and multiplies it through cuBLAS).  One 16-B chunk per thread.
__global__ void __launch_bounds__(256) k_embedding(half_t* __restrict__ out, const half_t* __restrict__ table, const int* __restrict__ ids,
                                                   const half_t* __restrict__ pos, long long n_tok, int D, int vocab, int T) {
  const int cv = D >> 3;
  long long total = n_tok * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long tkn = i / cv;
    int c = (int)(i - tkn * cv) * 8;
    int id = ids[tkn];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);     // memory safety only; the host entry validates host-side ids
    h8 v = *reinterpret_cast<const h8*>(table + (long long)id * D + c);
    if (pos) {
      h8 pv = *reinterpret_cast<const h8*>(pos + (long long)(tkn % T) * D + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)v[j] + (float)pv[j]);
    }
    *reinterpret_cast<h8*>(out + tkn * D + c) = v;
  }
}

extern "C" {

#define EW_UNARY(NAME, OP)                                                                 \
  int NAME(void* y, const void* x, long long n, tfStream_t s) {                            \
    TF_REQUIRE(y && x && n >= 0, #NAME ": bad arguments");                                 \
    if (n == 0) return TF_OK;                                                              \
    hipLaunchKernelGGL(k_unary<OP>, dim3(ew_grid(n >> 3)), dim3(EW_BLOCK), 0, tf_hs(s), (half_t*)y, (const half_t*)x, n); \
    TF_LAUNCH_CHECK();                                                                     \
    return TF_OK;                                                                          \
  }
EW_UNARY(tf_silu_f16, OP_SILU)
EW_UNARY(tf_sigmoid_f16, OP_SIGMOID)
EW_UNARY(tf_gelu_f16, OP_GELU)
EW_UNARY(tf_quick_gelu_f16, OP_QGELU)

int tf_add_f16(void* y, const void* a, const void* b, long long n, tfStream_t s) {
  TF_REQUIRE(y && a && b && n >= 0, "tf_add_f16: bad arguments");
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL(k_add<half_t>, dim3(ew_grid(n >> 3)), dim3(EW_BLOCK), 0, tf_hs(s), (half_t*)y, (const half_t*)a, (const half_t*)b, n);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_add_16(int dtype, void* y, const void* a, const void* b, long long n, tfStream_t s) {
  if (dtype == TF_DTYPE_F16) return tf_add_f16(y, a, b, n, s);
  TF_REQUIRE(dtype == TF_DTYPE_BF16, "tf_add_16: dtype=%d (0 = float16, 1 = bfloat16)", dtype);
  TF_REQUIRE(y && a && b && n >= 0, "tf_add_16: bad arguments");
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL(k_add<bf16_t>, dim3(ew_grid(n >> 3)), dim3(EW_BLOCK), 0, tf_hs(s), (bf16_t*)y, (const bf16_t*)a, (const bf16_t*)b, n);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
// ---- the bfloat16 step (config.set_dtype("bf16")): the small kernels of the sampler and the time-embedding chain on bfloat16 tensors, and the
// fp16 <-> bfloat16 conversion around the attention core (tf_sdpa_f16 stays the fp16 kernel)
int tf_silu_bf16(void* y, const void* x, long long n, tfStream_t s) {
  TF_REQUIRE(y && x && n >= 0, "tf_silu_bf16: bad arguments");
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL((k_unary<OP_SILU, bf16_t>), dim3(ew_grid(n >> 3)), dim3(EW_BLOCK), 0, tf_hs(s), (bf16_t*)y, (const bf16_t*)x, n);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_convert_f16_to_bf16(void* y_bf16, const void* x_f16, long long n, tfStream_t s) {
  TF_REQUIRE(y_bf16 && x_f16 && n >= 0, "tf_convert_f16_to_bf16: bad arguments");
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL((k_convert16<bf16_t, half_t>), dim3(ew_grid(n >> 3)), dim3(EW_BLOCK), 0, tf_hs(s), (bf16_t*)y_bf16, (const half_t*)x_f16, n);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_convert_bf16_to_f16(void* y_f16, const void* x_bf16, long long n, tfStream_t s) {
  TF_REQUIRE(y_f16 && x_bf16 && n >= 0, "tf_convert_bf16_to_f16: bad arguments");
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL((k_convert16<half_t, bf16_t>), dim3(ew_grid(n >> 3)), dim3(EW_BLOCK), 0, tf_hs(s), (half_t*)y_f16, (const bf16_t*)x_bf16, n);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_timestep_embedding_bf16(void* out, const void* step_params, int dim, float max_period, tfStream_t s) {
  TF_REQUIRE(out && step_params && dim > 0 && dim % 2 == 0, "tf_timestep_embedding_bf16: dim=%d must be even", dim);
  hipLaunchKernelGGL(k_timestep_embedding<bf16_t>, dim3(1), dim3(256), 0, tf_hs(s), (bf16_t*)out, (const float*)step_params, dim, max_period);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_cfg_duplicate_bf16(void* x2b, const void* latent, int B, int C, int H, int W, tfStream_t s) {
  TF_REQUIRE(x2b && latent && B > 0 && C > 0, "tf_cfg_duplicate_bf16: bad arguments");
  long long n = (long long)B * C * H * W;
  hipLaunchKernelGGL(k_cfg_duplicate<bf16_t>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, tf_hs(s), (bf16_t*)x2b, (const float*)latent, B, C, H * W);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_cfg_ddim_step_bf16(void* latent, const void* eps2, const void* params, int B, int C, int H, int W, tfStream_t s) {
  TF_REQUIRE(latent && eps2 && params && B > 0 && C > 0, "tf_cfg_ddim_step_bf16: bad arguments");
  long long n = (long long)B * C * H * W;
  hipLaunchKernelGGL(k_cfg_ddim<bf16_t>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, tf_hs(s), (float*)latent, (const bf16_t*)eps2, (const bf16_t*)nullptr, (const float*)params, B, C, H * W);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_embedding_f16(void* out, const void* table, const void* ids, const void* pos, long long n_tokens, int dim, int vocab, int T, tfStream_t s) {
  TF_REQUIRE(out && table && ids && n_tokens >= 0, "tf_embedding_f16: bad arguments");
  TF_REQUIRE(dim > 0 && dim % 8 == 0 && vocab >= 1 && (pos == nullptr || T >= 1), "tf_embedding_f16: dim=%d must be a positive multiple of 8 (vocab=%d T=%d)", dim, vocab, T);
  if (n_tokens == 0) return TF_OK;
  hipLaunchKernelGGL(k_embedding, dim3(ew_grid(n_tokens * (dim >> 3))), dim3(256), 0, tf_hs(s), (half_t*)out, (const half_t*)table, (const int*)ids,
                     (const half_t*)pos, n_tokens, dim, vocab, T > 0 ? T : 1);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
// Debugging aid (no reference counterpart): *flag = 1 when x holds a non-finite value.  Stream-ordered and capturable, so a
// whole step (eager or graph replay) can be instrumented without a host sync (tools/diag_graph.py).
__global__ void k_debug_nonfinite(const unsigned* __restrict__ x, long long nwords, int is_f32, int* flag) {
  bool bad = false;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nwords; i += (long long)gridDim.x * blockDim.x) {
    unsigned v = x[i];
    bad |= is_f32 ? ((v & 0x7f800000u) == 0x7f800000u) : (((v & 0x7c00u) == 0x7c00u) || ((v & 0x7c000000u) == 0x7c000000u));
  }
  if (bad) *flag = 1;
}
// Debugging aid: *sum (device u64) += position-weighted integer checksum of the words at x (order independent, so two runs of
// the same program over the same inputs must give identical sums for every array).
__global__ void k_debug_checksum(const unsigned* __restrict__ x, long long nwords, unsigned long long* sum) {
  unsigned long long acc = 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nwords; i += (long long)gridDim.x * blockDim.x)
    acc += (unsigned long long)x[i] * (unsigned long long)(((unsigned)i * 2654435761u) | 1u);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(sum, acc);
}
int tf_debug_checksum(const void* x, long long nbytes, void* sum, tfStream_t s) {
  TF_REQUIRE(x && sum && nbytes >= 0, "tf_debug_checksum: bad arguments");
  if (nbytes < 4) return TF_OK;
  hipLaunchKernelGGL(k_debug_checksum, dim3(ew_grid(nbytes / 4)), dim3(256), 0, tf_hs(s), (const unsigned*)x, nbytes / 4, (unsigned long long*)sum);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_debug_nonfinite(const void* x, long long nbytes, int is_f32, void* flag, tfStream_t s) {
  TF_REQUIRE(x && flag && nbytes >= 0, "tf_debug_nonfinite: bad arguments");
  if (nbytes < 4) return TF_OK;
  hipLaunchKernelGGL(k_debug_nonfinite, dim3(ew_grid(nbytes / 4)), dim3(256), 0, tf_hs(s), (const unsigned*)x, nbytes / 4, is_f32, (int*)flag);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_geglu_f16(void* y, const void* x, int rows, int C, tfStream_t s) {
  TF_REQUIRE(y && x && rows >= 0 && C > 0 && C % 8 == 0, "tf_geglu_f16: C=%d must be a positive multiple of 8", C);
  if (rows == 0) return TF_OK;
  hipLaunchKernelGGL(k_geglu, dim3(ew_grid((long long)rows * (C >> 3))), dim3(EW_BLOCK), 0, tf_hs(s), (half_t*)y, (const half_t*)x, (long long)rows, C);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_add_bias_nc_f16(void* y, const void* x, const void* b, int N, int HW, int C, tfStream_t s) {
  TF_REQUIRE(y && x && b && C % 8 == 0, "tf_add_bias_nc_f16: C=%d must be a multiple of 8", C);
  if ((long long)N * HW == 0) return TF_OK;
  hipLaunchKernelGGL(k_add_bias_nc, dim3(ew_grid((long long)N * HW * (C >> 3))), dim3(EW_BLOCK), 0, tf_hs(s), (half_t*)y, (const half_t*)x, (const half_t*)b, N, (long long)HW, C);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_upsample2x_nhwc_f16(void* y, const void* x, int N, int H, int W, int C, tfStream_t s) {
  TF_REQUIRE(y && x && C % 8 == 0, "tf_upsample2x_nhwc_f16: C=%d must be a multiple of 8", C);
  if ((long long)N * H * W == 0) return TF_OK;
  hipLaunchKernelGGL(k_upsample2x, dim3(ew_grid((long long)N * 4 * H * W * (C >> 3))), dim3(EW_BLOCK), 0, tf_hs(s), (half_t*)y, (const half_t*)x, N, H, W, C);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_concat_channels_f16(void* y, const void* a, const void* b, long long rows, int Ca, int Cb, tfStream_t s) {
  TF_REQUIRE(y && a && b && Ca % 8 == 0 && Cb % 8 == 0, "tf_concat_channels_f16: Ca=%d Cb=%d must be multiples of 8", Ca, Cb);
  if (rows == 0) return TF_OK;
  hipLaunchKernelGGL(k_concat, dim3(ew_grid(rows * ((Ca + Cb) >> 3))), dim3(EW_BLOCK), 0, tf_hs(s), (half_t*)y, (const half_t*)a, (const half_t*)b, rows, Ca, Cb);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_im2col_rows_nhwc_f16(void* y, const void* x, int N, int H, int W, int C, int R, int S, int stride, int pad, int Kpad, int ho_begin, int ho_end,
                            tfStream_t s) {
  TF_REQUIRE(y && x && Kpad >= R * S * C && stride >= 1, "tf_im2col_rows_nhwc_f16: Kpad=%d < R*S*C=%d", Kpad, R * S * C);
  int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
  TF_REQUIRE(ho_begin >= 0 && ho_begin <= ho_end && ho_end <= Ho, "tf_im2col_rows_nhwc_f16: rows [%d, %d) outside [0, %d)", ho_begin, ho_end, Ho);
  long long total = (long long)N * (ho_end - ho_begin) * Wo * Kpad;
  if (total <= 0) return TF_OK;
  hipLaunchKernelGGL(k_im2col, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, tf_hs(s), (half_t*)y, (const half_t*)x, N, H, W, C, R, S, stride, pad, ho_begin,
                     ho_end - ho_begin, Wo, Kpad);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_im2col_nhwc_f16(void* y, const void* x, int N, int H, int W, int C, int R, int S, int stride, int pad, int Kpad, tfStream_t s) {
  TF_REQUIRE(stride >= 1, "tf_im2col_nhwc_f16: stride=%d", stride);
  return tf_im2col_rows_nhwc_f16(y, x, N, H, W, C, R, S, stride, pad, Kpad, 0, (H + 2 * pad - R) / stride + 1, s);
}
int tf_cast_f32_to_f16(void* dst, const void* src, long long n, tfStream_t s) {
  TF_REQUIRE(dst && src && n >= 0, "tf_cast_f32_to_f16: bad arguments");
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL(k_cast_f32_f16, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, tf_hs(s), (half_t*)dst, (const float*)src, n);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_scale_cast_f32_to_f16(void* dst, const void* src, float scale, long long n, tfStream_t s) {
  TF_REQUIRE(dst && src && n >= 0, "tf_scale_cast_f32_to_f16: bad arguments");
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL(k_scale_cast_f32_f16, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, tf_hs(s), (half_t*)dst, (const float*)src, scale, n);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_cast_f16_to_f32(void* dst, const void* src, long long n, tfStream_t s) {
  TF_REQUIRE(dst && src && n >= 0, "tf_cast_f16_to_f32: bad arguments");
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL(k_cast_f16_f32, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, tf_hs(s), (float*)dst, (const half_t*)src, n);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_nchw_f32_to_nhwc_f16(void* dst, const void* src, int N, int C, int H, int W, tfStream_t s) {
  TF_REQUIRE(dst && src && N >= 0 && C > 0 && N <= 65535, "tf_nchw_f32_to_nhwc_f16: bad arguments");
  int HW = H * W;
  if ((long long)N * HW == 0) return TF_OK;
  hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(ceil_div(HW, 32), ceil_div(C, 32), N), dim3(256), 0, tf_hs(s), (half_t*)dst, (const float*)src, C, HW);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_nhwc_f16_to_nchw_f32(void* dst, const void* src, int N, int C, int H, int W, tfStream_t s) {
  TF_REQUIRE(dst && src && N >= 0 && C > 0 && N <= 65535, "tf_nhwc_f16_to_nchw_f32: bad arguments");
  int HW = H * W;
  if ((long long)N * HW == 0) return TF_OK;
  hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(ceil_div(HW, 32), ceil_div(C, 32), N), dim3(256), 0, tf_hs(s), (float*)dst, (const half_t*)src, C, HW);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_nhwc_to_nchw_f16(void* dst, const void* src, int N, int C, int H, int W, tfStream_t s) {
  TF_REQUIRE(dst && src && N >= 0 && N <= 65535 && C > 0, "tf_nhwc_to_nchw_f16: bad arguments");
  int HW = H * W;
  if ((long long)N * HW == 0) return TF_OK;
  hipLaunchKernelGGL(k_transpose_f16, dim3(ceil_div(C, 32), ceil_div(HW, 32), N), dim3(256), 0, tf_hs(s), (half_t*)dst, (const half_t*)src, HW, C);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_nchw_to_nhwc_f16(void* dst, const void* src, int N, int C, int H, int W, tfStream_t s) {
  TF_REQUIRE(dst && src && N >= 0 && N <= 65535 && C > 0, "tf_nchw_to_nhwc_f16: bad arguments");
  int HW = H * W;
  if ((long long)N * HW == 0) return TF_OK;
  hipLaunchKernelGGL(k_transpose_f16, dim3(ceil_div(HW, 32), ceil_div(C, 32), N), dim3(256), 0, tf_hs(s), (half_t*)dst, (const half_t*)src, C, HW);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
// decode tail (variants/sd.py:51-53): (x + 1) / 2 -> clip [0,1] -> * 255 -> uint8, NHWC f16 -> HWC u8
__global__ void __launch_bounds__(EW_BLOCK) k_to_u8(unsigned char* __restrict__ out, const half_t* __restrict__ x, long long n) {
  long long gs = (long long)gridDim.x * EW_BLOCK;
  for (long long i = (long long)blockIdx.x * EW_BLOCK + threadIdx.x; i < n; i += gs) {
    float v = ((float)x[i] + 1.0f) * 0.5f;
    v = fminf(fmaxf(v, 0.f), 1.f) * 255.f;
    out[i] = (unsigned char)v;
  }
}
int tf_image_to_u8(void* out, const void* x, long long n, tfStream_t s) {
  TF_REQUIRE(out && x && n >= 0, "tf_image_to_u8: bad arguments");
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL(k_to_u8, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, tf_hs(s), (unsigned char*)out, (const half_t*)x, n);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_scale_f32(void* x, float scale, long long n, tfStream_t s) {
  TF_REQUIRE(x && n >= 0, "tf_scale_f32: bad arguments");
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL(k_scale_f32, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, tf_hs(s), (float*)x, scale, n);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_add_bias_colmajor_f32(void* out, const void* bias, int BT, int OC, tfStream_t s) {
  TF_REQUIRE(out && bias && BT >= 0 && OC >= 0, "tf_add_bias_colmajor_f32: bad arguments");
  if ((long long)BT * OC == 0) return TF_OK;
  hipLaunchKernelGGL(k_add_bias_cm, dim3(ew_grid((long long)BT * OC)), dim3(EW_BLOCK), 0, tf_hs(s), (float*)out, (const float*)bias, BT, OC);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_transpose_f32(void* out, const void* inp, int ndim, const int* shape, const int* axes, tfStream_t s) {
  TF_REQUIRE(out && inp && shape && axes && ndim >= 1 && ndim <= 4, "tf_transpose_f32: ndim=%d must be 1..4", ndim);
  long long istride[4], n = 1;
  bool seen[4] = {false, false, false, false};
  for (int d = ndim - 1, st = 1; d >= 0; --d) { istride[d] = n; n *= shape[d]; (void)st; }
  PermArgs a;
  a.ndim = ndim;
  for (int d = 0; d < 4; ++d) { a.oshape[d] = 1; a.istride_of_o[d] = 0; }
  for (int d = 0; d < ndim; ++d) {
    TF_REQUIRE(axes[d] >= 0 && axes[d] < ndim && !seen[axes[d]], "tf_transpose_f32: axes is not a permutation");
    seen[axes[d]] = true;
    a.oshape[d] = shape[axes[d]];
    a.istride_of_o[d] = istride[axes[d]];
  }
  if (n == 0) return TF_OK;
  hipLaunchKernelGGL(k_permute_f32, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, tf_hs(s), (float*)out, (const float*)inp, a, n);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_softmax_rows_f32(void* out, const void* inp, int N, int C, tfStream_t s) {
  TF_REQUIRE(out && inp && N >= 0 && C >= 1, "tf_softmax_rows_f32: bad arguments");
  if (N == 0) return TF_OK;
  hipLaunchKernelGGL(k_softmax_rows, dim3(N), dim3(256), 0, tf_hs(s), (float*)out, (const float*)inp, C);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_softmax_mask_rows_f16(void* out, const void* inp, const void* mask_f32, long long rows, int C, int ldc, float scale, long long mask_rows,
                             tfStream_t s) {
  TF_REQUIRE(out && inp && rows >= 0 && C >= 1 && ldc >= C && rows < (1LL << 31), "tf_softmax_mask_rows_f16: bad arguments (rows=%lld C=%d ldc=%d)", rows, C, ldc);
  TF_REQUIRE(!mask_f32 || mask_rows >= 1, "tf_softmax_mask_rows_f16: mask_rows=%lld", mask_rows);
  if (rows == 0) return TF_OK;
  hipLaunchKernelGGL(k_softmax_mask_rows_f16<half_t>, dim3((unsigned)rows), dim3(256), 0, tf_hs(s), (half_t*)out, (const half_t*)inp, (const float*)mask_f32, C, ldc,
                     ldc, scale, mask_rows > 0 ? mask_rows : 1);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_softmax_mask_rows_f32in_f16(void* out, int ldo, const void* inp_f32, int ldi, const void* mask_f32, long long rows, int C, float scale, long long mask_rows,
                                   tfStream_t s) {
  TF_REQUIRE(out && inp_f32 && rows >= 0 && C >= 1 && ldo >= C && ldi >= C && rows < (1LL << 31), "tf_softmax_mask_rows_f32in_f16: bad arguments (rows=%lld C=%d ldo=%d ldi=%d)", rows, C, ldo, ldi);
  TF_REQUIRE(!mask_f32 || mask_rows >= 1, "tf_softmax_mask_rows_f32in_f16: mask_rows=%lld", mask_rows);
  if (rows == 0) return TF_OK;
  hipLaunchKernelGGL(k_softmax_mask_rows_f16<float>, dim3((unsigned)rows), dim3(256), 0, tf_hs(s), (half_t*)out, (const float*)inp_f32, (const float*)mask_f32, C, ldo,
                     ldi, scale, mask_rows > 0 ? mask_rows : 1);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_set_step_params(void* step_params, float timestep, float a_t, float a_prev, float guidance, tfStream_t s) {
  TF_REQUIRE(step_params, "tf_set_step_params: null pointer");
  hipLaunchKernelGGL(k_set_params, dim3(1), dim3(64), 0, tf_hs(s), (float*)step_params, timestep, a_t, a_prev, guidance);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_set_step_params_copy(void* step_params, float timestep, float a_t, float a_prev, float guidance, void* dst, const void* src, long long nbytes, tfStream_t s) {
  TF_REQUIRE(step_params && dst && src && nbytes >= 0 && nbytes % 16 == 0, "tf_set_step_params_copy: null pointer or nbytes=%lld not a multiple of 16", nbytes);
  TF_REQUIRE((((uintptr_t)dst | (uintptr_t)src) & 15) == 0, "tf_set_step_params_copy: dst and src must be 16-byte aligned");
  const long long n16 = nbytes / 16;
  int grid = (int)((n16 + 255) / 256);
  if (grid < 1) grid = 1;
  if (grid > 64) grid = 64;
  hipLaunchKernelGGL(k_set_params_copy, dim3(grid), dim3(256), 0, tf_hs(s), (float*)step_params, timestep, a_t, a_prev, guidance, (uint4*)dst, (const uint4*)src, n16);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_timestep_embedding_f16(void* out, const void* step_params, int dim, float max_period, tfStream_t s) {
  TF_REQUIRE(out && step_params && dim > 0 && dim % 2 == 0, "tf_timestep_embedding_f16: dim=%d must be even", dim);
  hipLaunchKernelGGL(k_timestep_embedding<half_t>, dim3(1), dim3(256), 0, tf_hs(s), (half_t*)out, (const float*)step_params, dim, max_period);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_cfg_duplicate_f16(void* x2b, const void* latent, int B, int C, int H, int W, tfStream_t s) {
  TF_REQUIRE(x2b && latent && B > 0 && C > 0, "tf_cfg_duplicate_f16: bad arguments");
  long long n = (long long)B * C * H * W;
  hipLaunchKernelGGL(k_cfg_duplicate<half_t>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, tf_hs(s), (half_t*)x2b, (const float*)latent, B, C, H * W);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_cfg_ddim_step_f32(void* latent, const void* eps2, const void* params, int B, int C, int H, int W, tfStream_t s) {
  TF_REQUIRE(latent && eps2 && params && B > 0 && C > 0, "tf_cfg_ddim_step_f32: bad arguments");
  long long n = (long long)B * C * H * W;
  hipLaunchKernelGGL(k_cfg_ddim<half_t>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, tf_hs(s), (float*)latent, (const half_t*)eps2, (const half_t*)nullptr, (const float*)params, B, C, H * W);
  TF_LAUNCH_CHECK();
  return TF_OK;
}
int tf_cfg_ddim_step2_f32(void* latent, const void* eps_uncond, const void* eps_cond, const void* params, int B, int C, int H, int W, tfStream_t s) {
  TF_REQUIRE(latent && eps_uncond && eps_cond && params && B > 0 && C > 0, "tf_cfg_ddim_step2_f32: bad arguments");
  long long n = (long long)B * C * H * W;
  hipLaunchKernelGGL(k_cfg_ddim<half_t>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, tf_hs(s), (float*)latent, (const half_t*)eps_uncond, (const half_t*)eps_cond, (const float*)params, B, C, H * W);
  TF_LAUNCH_CHECK();
  return TF_OK;
}

}  // extern "C"
