/*
 * tinyfusers_hip.h -- C ABI of libtinyfusers_hip.so, the MI355X (gfx950) native backend that takes the
 * place of the reference's `tinyfusers/native` ctypes layer (libcuda / libcudart / libnvrtc / libcublas
 * bindings, native/cuda/ops.py, native/cublas/ops.py, native/nvrtc/ops.py) and of the cuDNN / CuPy
 * calls the reference's per-op Python surface makes.
 *
 * Conventions (same as the reference's FFI):
 *   - every function returns int, 0 = success, otherwise a hipError_t value or a TF_E_* code; callers
 *     raise RuntimeError("<fn> failed with status N") exactly as storage/device.py:33-37 does.
 *     tf_last_error() additionally returns a thread-local human-readable string.
 *   - plain pointers and sizes only; device memory is caller-owned raw `void*`; opaque handles are
 *     created through an out-pointer and destroyed by the caller; no callbacks; nothing aborts.
 *   - all kernels are asynchronous on the given stream (NULL = the default stream, which is what the
 *     reference always uses: `stream := cuda.CUstream()`), safe to capture into a HIP graph.
 *   - activations are fp16 ("f16"), images in NHWC, tokens in (rows, C) row-major; accumulation fp32.
 *     NCHW fp32 (the reference's layout/dtype) only appears in the layout converters.
 * Each entry cites the reference interface (file:line under /root/reference) it replaces.
 */
#ifndef TINYFUSERS_HIP_H
#define TINYFUSERS_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tfStream_st* tfStream_t;
typedef struct tfEvent_st* tfEvent_t;
typedef struct tfGraph_st* tfGraph_t;
typedef struct tfComm_st* tfComm_t;
typedef struct tfFunction_st* tfFunction_t;   /* a kernel compiled at run time (tf_rtc_load) */

#define TF_OK 0
#define TF_E_ARG 10001
#define TF_E_UNSUPPORTED 10002
#define TF_E_WORKSPACE 10003
#define TF_E_STATE 10004

/* cudaMemcpy kinds kept numerically identical to native/cuda/ops.py:95-96 */
#define TF_MEMCPY_H2D 1
#define TF_MEMCPY_D2H 2
#define TF_MEMCPY_D2D 3

/* ---- context / device ------------------------------------------------------------------------ */
/* cuInit + cuCtxCreate_v2 (native/cuda/ops.py:7-18; call site storage/device.py:32-37) */
int tf_init(int device);
int tf_device_count(int* count);
/* cudaDeviceGetAttribute (native/cuda/ops.py:62-66). attr: 0 = CU count, 1 = max clock kHz,
 * 2 = wavefront size, 3 = LDS bytes per workgroup, 4 = L2 bytes, 5 = total memory MiB. */
int tf_device_attr(int* value, int attr, int device);
/* gcn arch name ("gfx950:...") copied into buf */
int tf_device_arch(char* buf, int buflen, int device);

/* ---- run-time compilation: replaces nvrtcCreateProgram / nvrtcCompileProgram / nvrtcGetCUBIN (native/nvrtc/ops.py:3-45) + cuModuleLoadData /
 * cuModuleGetFunction / cuLaunchKernel (native/cuda/ops.py:3-39) as storage/device.py:31-77 and its wrappers :79-233 use them.  hiprtc compiles
 * HIP source for the current device's architecture; a failed compilation returns TF_E_ARG with the compiler's log in tf_last_error().
 * params[i] points at the i-th kernel argument (cuLaunchKernel's kernelParams). */
int tf_rtc_load(tfFunction_t* out_fn, const char* source, const char* func_name);
int tf_rtc_launch(tfFunction_t fn, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned shared_bytes, tfStream_t s, void** params);
const char* tf_last_error(void);
int tf_version(void);

/* ---- memory: cudaMalloc / cudaMemcpy / cudaFree (native/cuda/ops.py:45-60; storage/tensor.py:20-46) */
int tf_malloc(void** out, size_t nbytes);
int tf_free(void* ptr);
int tf_memcpy(void* dst, const void* src, size_t nbytes, int kind);
int tf_memcpy_async(void* dst, const void* src, size_t nbytes, int kind, tfStream_t stream);
int tf_memset_async(void* dst, int value, size_t nbytes, tfStream_t stream);
/* strided device-to-device row copy (weight re-packing at install time: GEGLU block interleave, K padding) */
int tf_memcpy_2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t height,
                       tfStream_t stream);
int tf_host_alloc(void** out, size_t nbytes);   /* pinned host memory for async H2D of step parameters */
int tf_host_free(void* ptr);

/* ---- streams / events / graphs.  The reference has only the NULL stream and device-wide syncs
 * (attention/sdpa.py:70-71, variants/sd.py:40-41, ff/group_norm.py:6); timing is host-side
 * (example/sd1.py:71).  Here: explicit streams, HIP-event timing and whole-step HIP graphs. */
int tf_stream_create(tfStream_t* out);
int tf_stream_destroy(tfStream_t s);
int tf_stream_sync(tfStream_t s);
int tf_device_sync(void);
int tf_event_create(tfEvent_t* out);
int tf_event_destroy(tfEvent_t e);
int tf_event_record(tfEvent_t e, tfStream_t s);
int tf_event_sync(tfEvent_t e);
/* work launched on s after this call waits for everything recorded into e (fork / join of side branches, also while a
 * graph is being captured: the side stream joins the capture and the dependency becomes a graph edge) */
int tf_stream_wait_event(tfStream_t s, tfEvent_t e);
int tf_event_elapsed_ms(float* ms, tfEvent_t start, tfEvent_t stop);
int tf_graph_begin_capture(tfStream_t s);
int tf_graph_end_capture(tfStream_t s, tfGraph_t* out);
/* after a failed capture (an op raised between begin and end): leave capture mode and discard what was recorded, so that the
 * stream is usable again; no-op on a stream that is not capturing */
int tf_graph_abort_capture(tfStream_t s);
int tf_graph_launch(tfGraph_t g, tfStream_t s);
int tf_graph_destroy(tfGraph_t g);

/* ---- fp8 (OCP e4m3) conv / linear path, BASELINE config 5 (SD1.5 768x768 batch 32 on 8 GPUs, "fp8 MFMA conv/linear path"); the ops
 * extended are vision/conv2d.py:9-28 and ff/linear.py:112-121.  Weights: e4m3 with ONE fp32 scale per output channel, computed at pack
 * time (tf_pack_weight_fp8: scale[n] = max|w[n,:]| / 448); activations: e4m3 with scale 1, written directly by the kernels that
 * produce normalised tensors (tf_group_norm_apply_fp8, tf_layer_norm_fp8, the GEGLU epilogue of tf_linear_fp8) or by tf_quantize_fp8_f16;
 * fp32 accumulate on v_mfma_f32_16x16x32_fp8_fp8, fp16 residual stream (bias / time embedding / residual adds and the output stay
 * fp16).  Every channel count (conv) / K (linear) a multiple of 64.  x8 / x28 / w8 hold one byte per element in the layouts of the
 * fp16 entries (NHWC, KRSC, row-major). */
int tf_quantize_fp8_f16(void* y8, const void* x, long long n, float scale, tfStream_t s);
int tf_pack_weight_fp8(void* w8, void* scale_f32, const void* w, int N, int K, tfStream_t s);
size_t tf_conv2d_fp8_workspace(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample);
int tf_conv2d_fp8(void* y, const void* x8, const void* x28, const void* w8, const void* wscale, const void* bias, const void* bias_nc,
                  long long bias_nc_stride, const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad,
                  int upsample, void* workspace, size_t workspace_bytes, void* gn_partial, size_t gn_partial_bytes, int gn_groups, int* gn_chunks,
                  tfStream_t s);
int tf_linear_fp8(void* y, const void* x8, const void* w8, const void* wscale, const void* bias, const void* residual, int M, int N, int K, int act,
                  int out_fp8, void* workspace, size_t workspace_bytes, tfStream_t s);
int tf_group_norm_apply_fp8(void* y8, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                            int groups1, const void* partial2, int chunks2, int groups2, int N, int HW, int C1, int C2, int G, float eps, int silu,
                            tfStream_t s);
int tf_layer_norm_fp8(void* y8, const void* x, const void* gamma, const void* beta, int rows, int C, float eps, tfStream_t s);

/* ---- block-scaled e4m3 activations (round 4; the same reference ops: vision/conv2d.py:9-28, ff/linear.py:112-121, ff/group_norm.py:13-21,
 * ff/layer_norm.py:34-49, ff/nn.py:5-12).  An "mx8" tensor is ONE buffer: rows x C e4m3 codes (row = pixel / token, NHWC), then rows x C/32
 * E8M0 bytes: every 32 consecutive channels of a row share the power-of-two scale 2^(byte - 127) with amax / 2^e <= 448, code = e4m3(x / 2^e)
 * (round to nearest even, no saturation).  The GEMM feeds the bytes to the scale operand of v_mfma_scale_f32_16x16x128_f8f6f4; weights keep
 * tf_pack_weight_fp8's one fp32 scale per output channel.  The kernel is the 192- / 256-row ping-pong kernel: tf_mx8_gemm_supported /
 * tf_mx8_conv_supported say whether a shape fills the chip with its tiles (stride-1 convolutions without up-sampling, linears) -- a caller
 * keeps fp16 operands otherwise; an unsupported launch returns 10002.  out_mx: the GEGLU output (act = 1) as an mx8 tensor of M x N. */
size_t tf_mx8_bytes(long long rows, int C);
int tf_quantize_mx8_f16(void* y_mx, const void* x, long long rows, int C, tfStream_t s);
int tf_group_norm_apply_mx8(void* y_mx, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                            int groups1, const void* partial2, int chunks2, int groups2, int N, int HW, int C1, int C2, int G, float eps, int silu,
                            tfStream_t s);
int tf_layer_norm_mx8(void* y_mx, const void* x, const void* gamma, const void* beta, int rows, int C, float eps, tfStream_t s);
int tf_mx8_gemm_supported(int M, int N, int K, int act, int out_mx);
int tf_mx8_conv_supported(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample);
int tf_conv2d_mx8(void* y, const void* x_mx, const void* x2_mx, const void* w8, const void* wscale, const void* bias, const void* bias_nc,
                  long long bias_nc_stride, const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad,
                  void* workspace, size_t workspace_bytes, void* gn_partial, size_t gn_partial_bytes, int gn_groups, int* gn_chunks, tfStream_t s);
int tf_linear_mx8(void* y, const void* x_mx, const void* w8, const void* wscale, const void* bias, const void* residual, int M, int N, int K, int act,
                  int out_mx, void* workspace, size_t workspace_bytes, tfStream_t s);

/* ---- multi-GPU: one process per GPU, the path shards by image, and the only exchange is the one-off broadcast of
 * the packed weight arena.  The reference has no communication (device_id = 0, storage/device.py:23).  RCCL over xGMI, opened on
 * first use.  tf_comm_unique_id: rank 0 fills 128 bytes and hands them to the other ranks over any host channel; every rank then
 * calls tf_comm_init_rank on its own device (tf_init first); tf_bcast is stream-ordered and in place. */
#define TF_COMM_UNIQUE_ID_BYTES 128
int tf_comm_unique_id(void* id_out);
int tf_comm_init_rank(tfComm_t* out, const void* unique_id, int nranks, int rank);
int tf_bcast(tfComm_t comm, void* ptr, size_t nbytes, int root, tfStream_t s);
int tf_comm_destroy(tfComm_t comm);
/* per-launch profiling of the GEMM/conv kernel family with HIP events on the launch stream
 * (bench.py roofline leg): enable, run eagerly, then read back accumulated ms / flops / launches. */
int tf_prof_enable(int on);
int tf_prof_read(double* gemm_ms, double* gemm_flops, long long* gemm_launches);
/* the same accumulators with both brackets: `ms_with_reduce` spans every GEMM launch together with the split-K reduce launch that
 * finishes it (the time the conv / linear family really takes in a step), `ms_gemm_kernel_only` ends behind the k_igemm* kernel itself
 * (what tf_prof_read reports, the figure rocprofv3 lists under that kernel name) */
int tf_prof_read_full(double* ms_with_reduce, double* ms_gemm_kernel_only, double* gemm_flops, long long* gemm_launches);
float tf_prof_overhead_us(void);   /* the per-bracket event overhead tf_prof_enable(1) measured (spin-kernel pair, see csrc/gemm.hip) and subtracts */
int tf_prof_dump(const char* csv_path);   /* per-shape table: M,N,K,taps,tile,split-K,launches,ms,TFLOP/s */
/* the same event brackets around the launches of the other kernel families of the step while tf_prof_enable(1) is on (eager launches only):
 * family 1 = GroupNorm (k_gn_stats / k_gn_apply; ff/group_norm.py:3-21), 2 = the split-K reduce launches (incl. the forms that emit / apply the next
 * GroupNorm), 3 = LayerNorm (ff/layer_norm.py:8-49), 4 = SDPA (attention/sdpa.py:53-77).  `work` = the launches' ALGORITHMIC work: HBM bytes
 * (every operand read once, every result written once) for families 1-3, FLOPs (4 B NH Tq Tk d) for family 4 -- what bench.py prices against
 * the 8 TB/s HBM roof / the MFMA peak */
int tf_prof_read_family(int family, double* ms, double* work, long long* launches);
/* element type of the split-K partial slabs a split GEMM hands to its reduce launch: 16 (default) = fp16 -- half the bytes of that seam,
 * accumulated in fp32 in split order by the reducer -- or 32 = fp32 (rounds 1-3) */
int tf_gemm_splitk_partials(int bits);
/* test / tuning hook: force the GEMM tile (bm x bn in {256,128,64} x {256,160,128,64}) and split-K; 0,0,0 = heuristic */
int tf_gemm_force_config(int bm, int bn, int splitk);
/* per-shape choice of (tile, split-K, ring variant).  mode 1 (default): a shape that is not in the table is timed on its first eager
 * call and cached; mode 0: cost model only; mode 2: table only -- a shape without a row is an error naming the shape (what every rank
 * of a multi-GPU run uses, so that all ranks run the same kernels: tinyfusers_amd.native sets it when WORLD_SIZE > 1) */
int tf_gemm_autotune(int mode);
int tf_gemm_tune_save(const char* path);
int tf_gemm_tune_load(const char* path);
/* host-side view of the table (no device work).  key = {M, N, K, C1, C2, S, stride, upsample, act, flags} as tf_gemm_tune_save writes
 * a row, cfg = {bm, bn, splitk, variant, order}; tf_gemm_tune_query returns TF_E_STATE (10004) for a shape without a row */
int tf_gemm_tune_query(const int* key, int* cfg);
int tf_gemm_tune_count(int* n);
int tf_gemm_tune_entry(int index, int* key, int* cfg);
/* tf_gemm_tune_trace(1) starts remembering every shape key a launch looks up, tf_gemm_tune_trace_dump writes them (ten key fields + 1 / 0:
 * had a row), one per line -- which rows a workload needs (tools/gemm_keys.py) */
int tf_gemm_tune_trace(int on);
int tf_gemm_tune_trace_dump(const char* path);
/* test / tuning hook, kernel SELECTION only (every setting computes the same result): bits 3/4 force the deep/wide ring, 5/6 the
 * n-fastest/m-fastest block order, 7 (128) the patch variant of the 3x3 convolutions (k_igemm_patch), 8 (256) the variant whose consumer
 * waves issue part of the weight loads, each where the shape is eligible; 9 (512) the 256-row ping-pong kernel k_igemm_pp (with
 * tf_gemm_force_config(256, BN, split), BN in {128, 160, 256}: fails where it cannot take the launch), 10 (1024) the persistent short-K
 * kernel k_gemm_c4, 11 (2048) the patch form of the ping-pong kernel k_igemm_pp3 (3x3 / stride 1 convolutions on 96 / 48 / 24-pixel output rows, with
 * tf_gemm_force_config(192, BN, 1): the tile width is the instance's own; fails where it cannot take the launch), 13 (8192) the ping-pong kernel's
 * one-phase-per-k-step form on the three-slot ring, 14 (16384) the 256-row persistent short-K kernel k_gemm_c8 (with tf_gemm_force_config(256, 128, 1)), 15 (32768) the activation-resident short-K kernel k_gemm_ar (K = 256 / 320; with tf_gemm_force_config(128, 128, 1)).  The ABLATION bits -- 0 no stores,
 * 1 no MFMA, 2 no staging, 12 (4096) no fragment reads: wrong results by design -- exist only in the second library built with
 * -DTF_ABLATION (python -m tinyfusers_amd.build --ablation, loaded by the tools/ ablation scripts); the shipped library refuses them (10001) */
int tf_gemm_debug(int flags);

/* ---- layout / dtype converters (the API edge: the reference's arrays are fp32 NCHW) ----------- */
int tf_nchw_f32_to_nhwc_f16(void* dst, const void* src, int N, int C, int H, int W, tfStream_t s);
int tf_nhwc_f16_to_nchw_f32(void* dst, const void* src, int N, int C, int H, int W, tfStream_t s);
int tf_cast_f32_to_f16(void* dst, const void* src, long long n, tfStream_t s);
int tf_cast_f16_to_f32(void* dst, const void* src, long long n, tfStream_t s);
int tf_scale_cast_f32_to_f16(void* dst, const void* src, float scale, long long n, tfStream_t s);  /* 1/0.18215 * x (variants/sd.py:49) */
/* fp16 re-layouts for the VAE AttnBlock, which the reference runs on NCHW q/k/v (attention/attention.py:19-24) */
int tf_nhwc_to_nchw_f16(void* dst, const void* src, int N, int C, int H, int W, tfStream_t s);
int tf_nchw_to_nhwc_f16(void* dst, const void* src, int N, int C, int H, int W, tfStream_t s);
/* decode tail (variants/sd.py:51-53): (x+1)/2 -> clip -> *255 -> uint8; x (H,W,C) f16 -> out (H,W,C) u8 */
int tf_image_to_u8(void* out, const void* x, long long n, tfStream_t s);

/* ---- conv2d / linear: one implicit-GEMM MFMA kernel family ------------------------------------
 * tf_conv2d_f16 replaces conv_2d + Conv2d.__call__ (vision/conv2d.py:9-28, :48-58: cuDNN conv_fprop graph
 * rebuilt per call + NHWC->NCHW re-view + separate bias kernel), and folds in what surrounds it in the
 * UNet: the channel concat of vision/unet.py:72 (x2/C2), the nearest-2x upsample of unet.py:81-83
 * (upsample=1), the time-embedding broadcast add of vision/resnet.py:28 (bias_nc, one row per image) and
 * the residual add of resnet.py:30 / attention.py:75 (residual).
 *   x  : (N, H, W, C1) f16;  x2 : (N, H, W, C2) f16 or NULL (C2 = 0)
 *   w  : (Cout, R, S, C1+C2) f16 ("KRSC": the reference's (K,C,R,S) weight, vision/conv2d.py:52, with C innermost)
 *   bias : (Cout) f16 or NULL; residual : (N, Ho, Wo, Cout) f16 or NULL
 *   bias_nc : per-image bias rows f16 or NULL, image i reads bias_nc + i*bias_nc_stride (stride 0 = one row
 *             broadcast to every image, which is what resnet.py:28 does with its (1, C) embedding)
 *   y  : (N, Ho, Wo, Cout) f16, Ho = (H*(1+upsample) + 2*pad - R)/stride + 1
 *   workspace: fp32 scratch for split-K (may be NULL if tf_conv2d_workspace() returned 0).
 * Requirements: (C1+C2) % 8 == 0; R,S in {1,3} (any odd), dilation 1, groups 1 (all the UNet uses). */
int tf_conv2d_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc,
                  long long bias_nc_stride, const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R,
                  int S, int stride, int pad, int upsample, void* workspace, size_t workspace_bytes, tfStream_t s);
size_t tf_conv2d_workspace(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample);
/* tf_conv2d_fused_f16 = tf_conv2d_f16 plus two optional fusions of what surrounds a conv in ResBlock (vision/resnet.py:6-31):
 *
 * (1) extra 1x1 sources x3 | x4 (C3 | C4 channels, same N,H,W as x; NULL / 0 = off): K continues after the R*S taps with
 *     the channels of x3 then x4 read at the output pixel itself, i.e. y += conv1x1([x3 | x4]).  w is then
 *     (Cout, R*S*(C1+C2) + C3 + C4): each row = the KRSC conv row followed by the 1x1 row; bias = the sum of both biases.
 *     This is ResBlock's ``skip_connection(x) + h`` (:24, :31) computed inside its last conv instead of a separate
 *     Conv2d(1x1) launch plus a residual read.  Not combinable with upsample.
 *
 * (2) GroupNorm statistics of the OUTPUT (gn_partial != NULL), for the GroupNorm(32) that consumes it next
 *     (vision/resnet.py:17-18 after :11; attention/attention.py:60 after resnet.py:31): the group_norm of
 *     ff/group_norm.py:3-11 then needs no statistics pass of its own.
 *   gn_partial : f32 scratch of tf_conv2d_gn_partial_bytes(N, gn_groups) bytes, laid out (N, chunks, groups, 2)
 *                = per-chunk partial (sum, sum of squares) of the fp16 outputs, summed in a fixed order
 *   *gn_chunks : out; number of chunks written per image, to be handed to tf_group_norm_apply_f16.  0 means the
 *                statistics could not ride along for this shape (group wider than 64 channels or narrower than
 *                4, tiles that straddle images): y is complete, gn_partial is untouched, and the caller runs
 *                tf_group_norm_f16 instead. */
int tf_conv2d_fused_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc,
                        long long bias_nc_stride, const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R,
                        int S, int stride, int pad, int upsample, void* workspace, size_t workspace_bytes, const void* x3,
                        const void* x4, int C3, int C4, void* gn_partial, size_t gn_partial_bytes, int gn_groups,
                        int* gn_chunks, tfStream_t s);
/* tf_conv2d_fused_f16 that also APPLIES the GroupNorm (+ SiLU) reading its output (conv -> GroupNorm -> SiLU of vision/resnet.py:17-22,
 * :13-15 of the next block, attention/attention.py:66): when the shape runs split-K, the reduce kernel owns whole (image, group) slabs
 * (k_splitk_reduce_gn_apply), finishes the statistics and writes z = silu?(GroupNorm(y) * gamma + beta) next to y -- the
 * tf_group_norm_apply_f16 launch disappears.  *z_written = 1 when z was produced; 0 (shape ran unsplit, or groups the reduce cannot
 * hold): the caller runs tf_group_norm_apply_f16 with the statistics as before.  gn_groups must be the GroupNorm's group count. */
int tf_conv2d_fused_norm_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                             const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                             void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                             size_t gn_partial_bytes, int gn_groups, int* gn_chunks, void* z, const void* z_gamma, const void* z_beta, float z_eps,
                             int z_silu, int* z_written, tfStream_t s);
/* tf_conv2d_fused_f16 with the GroupNorm (+ SiLU) of its INPUT applied inside the launch: GroupNorm -> SiLU -> Conv2d of
 * vision/resnet.py:13-17, :22-27 and GroupNorm -> 1x1 conv of attention/attention.py:66-68 as ONE kernel.  The statistics of x (and
 * x2) arrive as the partials their producing convs emitted (gn_partial of tf_conv2d_fused_f16: in_partial / in_chunks / in_groups1
 * for x, in_partial2 / in_chunks2 / in_groups2 for x2 or NULL; sub-group contract of tf_group_norm_apply_cat_f16); in_groups is the
 * GroupNorm's group count over C1 + C2, in_gamma / in_beta its affine (or both NULL).  The extra 1x1 sources x3 / x4 stay raw.
 * Ask tf_conv2d_gn_supported first (3x3 / stride 1 / pad 1 and 1x1 convolutions with every channel count a multiple of 64, image
 * sizes the tile grid divides); TF_E_UNSUPPORTED otherwise, and the caller runs tf_group_norm_apply_f16 + the plain conv. */
int tf_conv2d_gn_supported(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample, int C3, int C4, int in_groups);
int tf_conv2d_gn_f16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                     const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                     void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                     size_t gn_partial_bytes, int gn_groups, int* gn_chunks, const void* in_gamma, const void* in_beta, const void* in_partial,
                     int in_chunks, int in_groups1, const void* in_partial2, int in_chunks2, int in_groups2, int in_groups, float in_eps, int in_silu,
                     tfStream_t s);
size_t tf_conv2d_fused_workspace(int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                                 int C3, int C4);
size_t tf_conv2d_gn_partial_bytes(int N, int groups);
/* cublasSgemm_v2 / cublasSgemmBatched (native/cublas/ops.py:22-53) as linear_cublas / gemm_batch call them
 * (ff/linear.py:58-61, :98-101): fp32, column-major, C = alpha op(A) op(B) + beta C; transa/transb 0 = N, 1 = T, 2 = C
 * (the cublasOperation_t numbers, native/cublas/ops.py:55-58).  alpha / beta are passed by value (the reference passes
 * host pointers to them).  The batched form takes DEVICE arrays of device pointers, like cuBLAS.  Test-only paths of the
 * reference (tests/linear.py:64-110): a plain fp32 FMA kernel, ascending-k accumulation. */
int tf_sgemm_f32(int transa, int transb, int m, int n, int k, float alpha, const void* A, int lda, const void* B, int ldb,
                 float beta, void* C, int ldc, tfStream_t s);
int tf_sgemm_batched_f32(int transa, int transb, int m, int n, int k, float alpha, const void* const* Aarray, int lda,
                         const void* const* Barray, int ldb, float beta, void* const* Carray, int ldc, int batch, tfStream_t s);
/* tf_linear_f16 replaces Linear.__call__ (ff/linear.py:112-121; live branch cp.dot(x, W^T)+b) and the
 * test-only cuBLAS/cuDNN paths linear_cublas / linear / gemm_batch (ff/linear.py:8-110):
 *   y(M,N) = act(x(M,K) . w(N,K)^T + bias(N)) + residual(M,N)
 * act: 0 none; 1 GEGLU (ff/nn.py:10-12): w holds 2*N rows packed by tf_pack_geglu order
 *      (16-row blocks alternating value / gate), bias likewise, y(M,N) = a * gelu_tanh(gate). */
int tf_linear_f16(void* y, const void* x, const void* w, const void* bias, const void* residual, int M, int N, int K,
                  int act, void* workspace, size_t workspace_bytes, tfStream_t s);
size_t tf_linear_workspace(int M, int N, int K, int act);
/* LayerNorm folded into the Linear that consumes it (ff/layer_norm.py:34-49 followed by ff/linear.py:112-121, the
 * norm1->to_q/k/v, norm2->to_q, norm3->GEGLU pairs of attention/attention.py:52-56):
 *   Linear(LN(x)) = rstd[m] * ( x . w'^T - mean[m] * colsum[n] ) + bias'[n],   w' = w * gamma, colsum = rowsum(w'),
 *   bias' = w . beta + bias.  tf_ln_fold_weights_f16 prepares (w', bias', colsum) once per weight set; the GEMM then runs
 *   on the RAW x and gets mean / rstd of every row from the activation fragments it streams anyway (no LayerNorm
 *   launch, no normalised tensor in HBM).  act as in tf_linear_f16 (for GEGLU fold first, then pack). */
int tf_ln_fold_weights_f16(void* w_out, void* bias_out, void* colsum_out_f32, const void* w, const void* bias, const void* gamma,
                           const void* beta, int N, int K, tfStream_t s);
int tf_linear_ln_f16(void* y, const void* x, const void* w_folded, const void* bias_folded, const void* colsum_f32, const void* residual,
                     int M, int N, int K, int act, float eps, tfStream_t s);
/* M == 1..8 weight-streaming GEMV with optional SiLU on the input (the time-embedding MLP and the 22
 * ResBlock emb_layers, vision/unet.py:53-54, vision/resnet.py:27): y(M,N) = silu?(x)(M,K) . w(N,K)^T + b */
int tf_gemv_f16(void* y, const void* x, const void* w, const void* bias, int M, int N, int K, int silu_input, tfStream_t s);

/* ---- attention: fused flash-style SDPA, replaces scaled_dot_product_attention (attention/sdpa.py:53-77:
 * cp.matmul + softmax_kernel (native/cuda/softmax.cu:24-112) + cp.matmul with the score matrix in HBM).
 * o = softmax(q k^T / sqrt(HS)) v; element strides (batch, head, token) per tensor, innermost dim HS is
 * contiguous.  Output strides select the head-merge layout: (NH*T*HS, T*HS, HS) reproduces the reference's
 * CrossAttention (attention/attention.py:38-39, no transpose back: SURVEY D11); (T*NH*HS, HS, NH*HS) is the
 * LDM-intended merge.  causal != 0 applies the CLIP causal mask (attention.py:94).  HS % 8 == 0, HS <= 160. */
int tf_sdpa_f16(void* o, const void* q, const void* k, const void* v, int B, int NH, int Tq, int Tk, int HS,
                long long q_sb, long long q_sh, long long q_st, long long k_sb, long long k_sh, long long k_st,
                long long v_sb, long long v_sh, long long v_st, long long o_sb, long long o_sh, long long o_st,
                int causal, tfStream_t s);
/* row softmax over (N, C) fp32 -- Device.softmax (storage/device.py:129-157; softmax_func.cu:22-113) */
int tf_softmax_rows_f32(void* out, const void* inp, int N, int C, tfStream_t s);
/* softmax step of the UNFUSED attention (attention/sdpa.py:63-75 as the reference runs it: scale * matmul, + mask (:67-68: bool ->
 * -inf where false, anything else additive), softmax kernel, matmul).  Serves what tf_sdpa_f16 does not: arbitrary masks and head sizes
 * beyond 160 (AttnBlock's single head of 512, attention/attention.py:10-24).  out[r, c] = softmax_c(scale * inp[r, c] + mask[r % mask_rows, c]);
 * rows are ldc apart (ldc >= C; out's pad columns are zero-filled); mask_f32: (mask_rows, C) fp32 additive or NULL. */
int tf_softmax_mask_rows_f16(void* out, const void* inp, const void* mask_f32, long long rows, int C, int ldc, float scale, long long mask_rows,
                             tfStream_t s);
/* The same two steps with the scores kept in fp32 between them, as the reference keeps them (attention/sdpa.py:63-66: fp32 cp.matmul
 * into `preatt`, softmax kernel native/cuda/softmax.cu:24-112 on fp32): tf_linear_f32out_f16 stores the raw fp32 accumulators
 * y32(M,N) = x(M,K) . w(N,K)^T (fp16 operands, no bias / activation), tf_softmax_mask_rows_f32in_f16 reads them (rows ldi floats apart)
 * and writes fp16 probabilities (rows ldo halves apart, pad columns C..ldo-1 zero-filled).  A scaled logit of several hundred -- the
 * d = 512 single-head AttnBlock on real weights -- would lose 0.1 ... 1 in fp16 in front of the exp. */
int tf_linear_f32out_f16(void* y_f32, const void* x, const void* w, int M, int N, int K, tfStream_t s);
int tf_softmax_mask_rows_f32in_f16(void* out, int ldo, const void* inp_f32, int ldi, const void* mask_f32, long long rows, int C, float scale,
                                   long long mask_rows, tfStream_t s);

/* ---- normalisation ---------------------------------------------------------------------------
 * group_norm + GroupNorm affine (+ the SiLU that always follows it in ResBlock / UNet.out)
 * (ff/group_norm.py:3-11, :13-21; storage/tensor.py:68-70): x,y (N, HW, C) f16 NHWC, G groups, biased var.
 * x2/C2: optional second source for the channel concat (vision/unet.py:72): channels [C1, C1+C2) come from x2.
 * gamma/beta f16 (C) or NULL (plain group_norm).  workspace: tf_group_norm_workspace bytes. */
int tf_group_norm_f16(void* y, const void* x, const void* x2, const void* gamma, const void* beta, int N, int HW,
                      int C1, int C2, int G, float eps, int silu, void* workspace, size_t workspace_bytes, tfStream_t s);
size_t tf_group_norm_workspace(int N, int HW, int C, int G);
/* second half of tf_group_norm_f16 (normalise + affine [+ SiLU]) on statistics already produced by
 * tf_conv2d_fused_f16: partial (N, chunks, G, 2) f32.  Same arithmetic as tf_group_norm_f16 from the fold onwards. */
int tf_group_norm_apply_f16(void* y, const void* x, const void* gamma, const void* beta, const void* partial, int chunks,
                            int N, int HW, int C, int G, float eps, int silu, tfStream_t s);
/* GroupNorm(G) of the channel concat [x | x2] of two EQUALLY wide tensors (vision/unet.py:72 into resnet.py:8) whose
 * statistics came from tf_conv2d_fused_f16 as G-group partials of each half: a group of the concat is two adjacent
 * groups of one half, so no statistics pass over the concat is needed.  partial (N, chunks, G, 2), partial2 (N, chunks2, G, 2). */
int tf_group_norm_apply2_f16(void* y, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial,
                             int chunks, const void* partial2, int chunks2, int N, int HW, int C1, int G, float eps, int silu,
                             tfStream_t s);
/* General form: the producers' partials have groups1 / groups2 sub-groups of equal width (C1/groups1 == C2/groups2) that tile the groups of
 * the concat ((C1+C2)/G = m * width, m <= 8); a group may straddle the two sources (1280 + 640 channels, 32 groups: partials with 64 and
 * 32 sub-groups of 20 channels).  tf_group_norm_apply2_f16 is the equal-split case groups1 = groups2 = G. */
int tf_group_norm_apply_cat_f16(void* y, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                                int groups1, const void* partial2, int chunks2, int groups2, int N, int HW, int C1, int C2, int G, float eps,
                                int silu, tfStream_t s);
/* LayerNorm over the last dim (ff/layer_norm.py:8-32, :34-49; semantics = F.layer_norm, tests/layer_norm.py:38) */
int tf_layer_norm_f16(void* y, const void* x, const void* gamma, const void* beta, int rows, int C, float eps, tfStream_t s);

/* ---- bfloat16 forms ---------------------------------------------------------------------------
 * The reference's op tests run every case in bfloat16 as well as float16 (tests/group_norm.py:12-19 and
 * tests/layer_norm.py:13-27 with atol = rtol = 0.125; tests/linear.py:13 lists it): the same four operators with every
 * 16-bit tensor (x, x2, w, gamma, beta, bias, residual, y) holding bfloat16.  Statistics and accumulation are fp32 as
 * in the float16 forms; the GEMMs use the gfx950 bf16 MFMA (v_mfma_f32_16x16x32_bf16) on the tuned kernels of the float16 forms (round 5; these
 * entries take no workspace, so they never split K: tf_linear_16 / tf_conv2d_fused_16 below are the full forms). */
int tf_group_norm_bf16(void* y, const void* x, const void* x2, const void* gamma, const void* beta, int N, int HW,
                       int C1, int C2, int G, float eps, int silu, void* workspace, size_t workspace_bytes, tfStream_t s);
int tf_layer_norm_bf16(void* y, const void* x, const void* gamma, const void* beta, int rows, int C, float eps, tfStream_t s);
/* y(M,N) = x(M,K) . w(N,K)^T + bias(N) + residual(M,N)   (ff/linear.py:112-121) */
int tf_linear_bf16(void* y, const void* x, const void* w, const void* bias, const void* residual, int M, int N, int K, tfStream_t s);
/* tf_conv2d_f16's arguments without the workspace (vision/conv2d.py:9-28): NHWC x (+ concat x2), w (Cout, R, S, C1 + C2) */
/* the bfloat16 STEP (round 4: config.set_dtype("bf16"), bench.py --dtype bf16): the sampler on bfloat16 tensors throughout -- conv / linear /
 * GEGLU on the bf16 MFMA (tf_conv2d_bf16, tf_linear_act_bf16), GroupNorm / LayerNorm / SiLU in bfloat16, the time-embedding chain, CFG duplicate
 * and the CFG + DDIM update reading bfloat16 (vision/unet.py:51-97, variants/sd.py:14-46); since round 5 the step runs on the dtype-tagged fused
 * entries below (tf_*_16) and attention is native bfloat16 (tf_sdpa_16); tf_convert_* remain for callers that mix the types */
int tf_linear_act_bf16(void* y, const void* x, const void* w, const void* bias, const void* residual, int M, int N, int K, int act, tfStream_t s);
int tf_silu_bf16(void* y, const void* x, long long n, tfStream_t s);
int tf_convert_f16_to_bf16(void* y_bf16, const void* x_f16, long long n, tfStream_t s);
int tf_convert_bf16_to_f16(void* y_f16, const void* x_bf16, long long n, tfStream_t s);
int tf_timestep_embedding_bf16(void* out, const void* step_params, int dim, float max_period, tfStream_t s);
int tf_cfg_duplicate_bf16(void* x2b, const void* latent, int B, int C, int H, int W, tfStream_t s);
int tf_cfg_ddim_step_bf16(void* latent, const void* eps2, const void* params, int B, int C, int H, int W, tfStream_t s);
int tf_conv2d_bf16(void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                   const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                   tfStream_t s);

/* ---- dtype-tagged forms of the step's fused entries (round 5): the bfloat16 step runs the SAME structure as the float16 step -- every fusion,
 * split-K, the tuned tiles -- so each of these is its _f16 namesake (same arguments, same semantics, same reference lines) with one leading tag
 * saying what every 16-bit tensor of the call holds: TF_DTYPE_F16 (= the _f16 entry) or TF_DTYPE_BF16 (v_mfma_f32_16x16x32_bf16; fp32 accumulation,
 * statistics and split-K partial slabs).  The reference's own op tests run bfloat16 next to float16 (tests/group_norm.py:12-19, tests/layer_norm.py:12-19,
 * tests/linear.py:13).  tf_sdpa_16: attention natively in bfloat16 (Q K^T and P V on the bf16 MFMA, P rounded to bfloat16): no fp16 hop, so
 * activations beyond 65504 stay finite. */
#define TF_DTYPE_F16 0
#define TF_DTYPE_BF16 1
int tf_conv2d_fused_16(int dtype, void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                       const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                       void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                       size_t gn_partial_bytes, int gn_groups, int* gn_chunks, tfStream_t s);
int tf_conv2d_fused_norm_16(int dtype, void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                            const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                            void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                            size_t gn_partial_bytes, int gn_groups, int* gn_chunks, void* z, const void* z_gamma, const void* z_beta, float z_eps,
                            int z_silu, int* z_written, tfStream_t s);
int tf_conv2d_gn_16(int dtype, void* y, const void* x, const void* x2, const void* w, const void* bias, const void* bias_nc, long long bias_nc_stride,
                    const void* residual, int N, int H, int W, int C1, int C2, int Cout, int R, int S, int stride, int pad, int upsample,
                    void* workspace, size_t workspace_bytes, const void* x3, const void* x4, int C3, int C4, void* gn_partial,
                    size_t gn_partial_bytes, int gn_groups, int* gn_chunks, const void* in_gamma, const void* in_beta, const void* in_partial,
                    int in_chunks, int in_groups1, const void* in_partial2, int in_chunks2, int in_groups2, int in_groups, float in_eps, int in_silu,
                    tfStream_t s);
int tf_linear_16(int dtype, void* y, const void* x, const void* w, const void* bias, const void* residual, int M, int N, int K, int act,
                 void* workspace, size_t workspace_bytes, tfStream_t s);
int tf_ln_fold_weights_16(int dtype, void* w_out, void* bias_out, void* colsum_out_f32, const void* w, const void* bias, const void* gamma, const void* beta,
                          int N, int K, tfStream_t s);
int tf_linear_ln_16(int dtype, void* y, const void* x, const void* w_folded, const void* bias_folded, const void* colsum_f32, const void* residual,
                    int M, int N, int K, int act, float eps, tfStream_t s);
int tf_gemv_16(int dtype, void* y, const void* x, const void* w, const void* bias, int M, int N, int K, int silu_input, tfStream_t s);
int tf_sdpa_16(int dtype, void* o, const void* q, const void* k, const void* v, int B, int NH, int Tq, int Tk, int HS,
               long long q_sb, long long q_sh, long long q_st, long long k_sb, long long k_sh, long long k_st,
               long long v_sb, long long v_sh, long long v_st, long long o_sb, long long o_sh, long long o_st, int causal, tfStream_t s);
int tf_group_norm_apply_16(int dtype, void* y, const void* x, const void* gamma, const void* beta, const void* partial, int chunks,
                           int N, int HW, int C, int G, float eps, int silu, tfStream_t s);
int tf_group_norm_apply_cat_16(int dtype, void* y, const void* x, const void* x2, const void* gamma, const void* beta, const void* partial, int chunks,
                               int groups1, const void* partial2, int chunks2, int groups2, int N, int HW, int C1, int C2, int G, float eps,
                               int silu, tfStream_t s);
int tf_add_16(int dtype, void* y, const void* a, const void* b, long long n, tfStream_t s);

/* ---- elementwise (storage/tensor.py:64-86; ff/nn.py:10-12; vision/unet.py:72, :81-83) ---------- */
int tf_silu_f16(void* y, const void* x, long long n, tfStream_t s);
int tf_sigmoid_f16(void* y, const void* x, long long n, tfStream_t s);
int tf_gelu_f16(void* y, const void* x, long long n, tfStream_t s);
int tf_quick_gelu_f16(void* y, const void* x, long long n, tfStream_t s);
/* Embedding lookup (ff/embedding.py:10-24; the CLIP text encoder's token + position embeddings, vae/encoder.py:68-73):
 * out(n_tokens, dim) f16 = table[ids[i], :] (+ pos[i % T, :] when pos != NULL).  ids: n_tokens int32 ON THE DEVICE. */
int tf_embedding_f16(void* out, const void* table, const void* ids, const void* pos, long long n_tokens, int dim, int vocab,
                     int T, tfStream_t s);
/* Debugging aid, no reference counterpart: *flag (device int) = 1 when the nbytes at x hold a non-finite f16 (is_f32 = 0) or f32
 * value.  Stream-ordered and capturable (tools/diag_graph.py instruments a whole step with it). */
int tf_debug_nonfinite(const void* x, long long nbytes, int is_f32, void* flag, tfStream_t s);
int tf_debug_checksum(const void* x, long long nbytes, void* sum, tfStream_t s);   /* *sum (device u64) += weighted word checksum of x */
int tf_geglu_f16(void* y, const void* x, int rows, int C, tfStream_t s);          /* x (rows,2C) -> y (rows,C) */
int tf_add_f16(void* y, const void* a, const void* b, long long n, tfStream_t s);
int tf_add_bias_nc_f16(void* y, const void* x, const void* bias_nc, int N, int HW, int C, tfStream_t s);
int tf_upsample2x_nhwc_f16(void* y, const void* x, int N, int H, int W, int C, tfStream_t s);
int tf_concat_channels_f16(void* y, const void* a, const void* b, long long rows, int Ca, int Cb, tfStream_t s);
/* im2col for the 4-channel conv_in (K = R*S*C padded to Kpad): y (N*Ho*Wo, Kpad) */
int tf_im2col_nhwc_f16(void* y, const void* x, int N, int H, int W, int C, int R, int S, int stride, int pad, int Kpad, tfStream_t s);
/* the same for output rows [ho_begin, ho_end) of every image: y (N*(ho_end-ho_begin)*Wo, Kpad).  The 10000 x 10000 input of the
 * reference's own conv test (tests/conv2d.py:13-33) runs as row bands, each band's patch matrix below the 2 GiB a buffer descriptor spans */
int tf_im2col_rows_nhwc_f16(void* y, const void* x, int N, int H, int W, int C, int R, int S, int stride, int pad, int Kpad, int ho_begin, int ho_end,
                            tfStream_t s);
/* own-runtime kernels of the reference (native/cuda kernels via storage/device.py:79-233), fp32 */
int tf_scale_f32(void* x, float scale, long long n, tfStream_t s);                        /* scale_tensor_func.cu:5-10 */
int tf_add_bias_colmajor_f32(void* out, const void* bias, int BT, int OC, tfStream_t s);  /* add_bias_func.cu:1-9 */
int tf_transpose_f32(void* out, const void* inp, int ndim, const int* shape, const int* axes, tfStream_t s); /* transpose.cu, transpose4d.cu */

/* ---- sampler pieces (vision/unet.py:92-97; variants/sd.py:14-25, :27-46) ------------------------
 * step_params (device, fp32): [0] timestep, [1] a_t, [2] a_prev, [3] guidance -- written once per step by
 * tf_set_step_params (values travel as kernel arguments, so there is no host buffer to race with) ahead of the
 * replay of the step's HIP graph, whose kernels read them from device memory. */
int tf_set_step_params(void* step_params, float timestep, float a_t, float a_prev, float guidance, tfStream_t s);
/* the same launch also copies nbytes (a multiple of 16, both pointers 16-byte aligned) src -> dst: the time-embedding chain of
 * vision/unet.py:54-56 + the ResBlocks' Linear(SiLU(emb)) (vision/resnet.py:28) depends on the timestep alone, so a sampler computes the
 * row of a timestep once, keeps it, and hands it to the captured step through this copy instead of four launches per step */
int tf_set_step_params_copy(void* step_params, float timestep, float a_t, float a_prev, float guidance, void* dst, const void* src, long long nbytes, tfStream_t s);
int tf_timestep_embedding_f16(void* out, const void* step_params, int dim, float max_period, tfStream_t s);
/* latent (B,C,H,W) f32 NCHW  ->  unet input (2B,H,W,C) f16 NHWC = [latent ; latent]  (variants/sd.py:31) */
int tf_cfg_duplicate_f16(void* x2b_nhwc, const void* latent_nchw_f32, int B, int C, int H, int W, tfStream_t s);
/* e = e_u + g (e_c - e_u); DDIM sigma=0 update, latent updated in place (variants/sd.py:44-45, :14-25) */
int tf_cfg_ddim_step_f32(void* latent_nchw_f32, const void* unet_out_2b_nhwc_f16, const void* step_params, int B, int C,
                         int H, int W, tfStream_t s);
/* the same with the two UNet outputs in separate tensors (B,4,H,W each): the unconditional and the conditional half of the CFG pair
 * (variants/sd.py:31-45) run as two independent UNet chains on two streams / graph branches (config.cfg_parallel) */
int tf_cfg_ddim_step2_f32(void* latent, const void* eps_uncond, const void* eps_cond, const void* step_params, int B, int C, int H, int W, tfStream_t s);

#ifdef __cplusplus
}
#endif
#endif
