// Context / memory / stream / event / graph entry points of libtinyfusers_hip.so.
// Replaces the reference's ctypes bindings to libcuda / libcudart (native/cuda/ops.py:3-67).
#include "common.h"
#include "../../include/tinyfusers_hip.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void tf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- family profiling (common.h: TfProfScope) -----------------------------------------------------------------------------------
#include <vector>
bool g_tf_prof = false;
float g_tf_prof_overhead_ms = 0.f;
namespace {
struct FamRec { hipEvent_t a, b; int fam; double work; };
std::vector<FamRec> g_fam_pending;
FamRec g_fam_open = {nullptr, nullptr, 0, 0.0};
double g_fam_ms[TF_PROF_NFAM], g_fam_work[TF_PROF_NFAM];
long long g_fam_n[TF_PROF_NFAM];
}
void tf_prof_fam_reset() {
  for (auto& r : g_fam_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_fam_pending.clear();
  for (int i = 0; i < TF_PROF_NFAM; ++i) { g_fam_ms[i] = 0.0; g_fam_work[i] = 0.0; g_fam_n[i] = 0; }
}
void tf_prof_fam_begin(int family, double work, hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cs);
  g_fam_open = {nullptr, nullptr, 0, 0.0};
  if (cs != hipStreamCaptureStatusNone || family < 1 || family >= TF_PROF_NFAM) return;     // (never inside a capture: an event pair there is a graph node)
  FamRec r = {nullptr, nullptr, family, work};
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, st);
  g_fam_open = r;
}
void tf_prof_fam_end(hipStream_t st) {
  if (!g_fam_open.fam) return;
  (void)hipEventRecord(g_fam_open.b, st);
  g_fam_pending.push_back(g_fam_open);
  g_fam_open = {nullptr, nullptr, 0, 0.0};
}
void tf_prof_fam_add(int family, double work, double ms) {
  if (family < 1 || family >= TF_PROF_NFAM) return;
  g_fam_ms[family] += ms; g_fam_work[family] += work; g_fam_n[family] += 1;
}

extern "C" {

int tf_prof_read_family(int family, double* ms, double* work, long long* launches) {
  TF_REQUIRE(family >= 1 && family < TF_PROF_NFAM, "tf_prof_read_family: family=%d (1 GroupNorm, 2 split-K reduce, 3 LayerNorm, 4 SDPA)", family);
  for (auto& r : g_fam_pending) {
    float t = 0.f;
    TF_HIP(hipEventSynchronize(r.b));
    TF_HIP(hipEventElapsedTime(&t, r.a, r.b));
    t -= g_tf_prof_overhead_ms;
    g_fam_ms[r.fam] += t > 0.f ? t : 0.f; g_fam_work[r.fam] += r.work; g_fam_n[r.fam] += 1;
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  g_fam_pending.clear();
  if (ms) *ms = g_fam_ms[family];
  if (work) *work = g_fam_work[family];
  if (launches) *launches = g_fam_n[family];
  return TF_OK;
}

const char* tf_last_error(void) { return g_err; }
int tf_version(void) { return 100; }

int tf_init(int device) {
  int n = 0;
  TF_HIP(hipGetDeviceCount(&n));
  TF_REQUIRE(device >= 0 && device < n, "tf_init: device %d out of range (%d visible)", device, n);
  TF_HIP(hipSetDevice(device));
  TF_HIP(hipFree(0));  // force context creation (cuCtxCreate_v2 in the reference)
  return TF_OK;
}

int tf_device_count(int* count) {
  TF_REQUIRE(count, "tf_device_count: null out pointer");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) { *count = 0; tf_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); (void)hipGetLastError(); return (int)e; }
  return TF_OK;
}

int tf_device_attr(int* value, int attr, int device) {
  TF_REQUIRE(value, "tf_device_attr: null out pointer");
  hipDeviceProp_t p;
  TF_HIP(hipGetDeviceProperties(&p, device));
  switch (attr) {
    case 0: *value = p.multiProcessorCount; break;
    case 1: *value = p.clockRate; break;
    case 2: *value = p.warpSize; break;
    case 3: *value = (int)p.sharedMemPerBlock; break;
    case 4: *value = p.l2CacheSize; break;
    case 5: *value = (int)(p.totalGlobalMem >> 20); break;
    default: tf_set_error("tf_device_attr: unknown attribute %d", attr); return TF_E_ARG;
  }
  return TF_OK;
}

int tf_device_arch(char* buf, int buflen, int device) {
  TF_REQUIRE(buf && buflen > 0, "tf_device_arch: bad buffer");
  hipDeviceProp_t p;
  TF_HIP(hipGetDeviceProperties(&p, device));
  strncpy(buf, p.gcnArchName, buflen - 1);
  buf[buflen - 1] = 0;
  return TF_OK;
}

int tf_malloc(void** out, size_t nbytes) {
  TF_REQUIRE(out, "tf_malloc: null out pointer");
  TF_HIP(hipMalloc(out, nbytes ? nbytes : 1));
  return TF_OK;
}
int tf_free(void* ptr) { TF_HIP(hipFree(ptr)); return TF_OK; }

static hipMemcpyKind kind_of(int kind) {
  return kind == TF_MEMCPY_H2D ? hipMemcpyHostToDevice : kind == TF_MEMCPY_D2H ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
}
int tf_memcpy(void* dst, const void* src, size_t nbytes, int kind) {
  TF_REQUIRE(kind >= 1 && kind <= 3, "tf_memcpy: kind %d (1=H2D 2=D2H 3=D2D)", kind);
  TF_HIP(hipMemcpy(dst, src, nbytes, kind_of(kind)));
  // A copy from pageable host memory may return once the data sits in the staging buffer, with the DMA still queued on the
  // NULL stream -- and the non-blocking streams this library creates are not ordered behind that stream.  The reference uses
  // the call as "copy, then launch" (storage/tensor.py:25-29), so make it hold: drain the NULL stream after every copy that
  // writes device memory (one host wait per upload; uploads are not on the step path).
  if (kind != 2) TF_HIP(hipStreamSynchronize(nullptr));
  return TF_OK;
}
int tf_memcpy_async(void* dst, const void* src, size_t nbytes, int kind, tfStream_t s) {
  TF_REQUIRE(kind >= 1 && kind <= 3, "tf_memcpy_async: kind %d", kind);
  TF_HIP(hipMemcpyAsync(dst, src, nbytes, kind_of(kind), tf_hs(s)));
  return TF_OK;
}
int tf_memset_async(void* dst, int value, size_t nbytes, tfStream_t s) {
  TF_HIP(hipMemsetAsync(dst, value, nbytes, tf_hs(s)));
  return TF_OK;
}
int tf_memcpy_2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, tfStream_t s) {
  TF_REQUIRE(dst && src && width <= dpitch && width <= spitch, "tf_memcpy_2d_async: bad pitches");
  if (width == 0 || height == 0) return TF_OK;
  TF_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, tf_hs(s)));
  return TF_OK;
}
int tf_host_alloc(void** out, size_t nbytes) { TF_REQUIRE(out, "null out"); TF_HIP(hipHostMalloc(out, nbytes, hipHostMallocDefault)); return TF_OK; }
int tf_host_free(void* ptr) { TF_HIP(hipHostFree(ptr)); return TF_OK; }

int tf_stream_create(tfStream_t* out) {
  TF_REQUIRE(out, "tf_stream_create: null out pointer");
  tfStream_st* s = new tfStream_st;
  hipError_t e = hipStreamCreateWithFlags(&s->s, hipStreamNonBlocking);
  if (e != hipSuccess) { delete s; tf_set_error("hipStreamCreate: %s", hipGetErrorString(e)); return (int)e; }
  *out = s;
  return TF_OK;
}
int tf_stream_destroy(tfStream_t s) { if (!s) return TF_OK; TF_HIP(hipStreamDestroy(s->s)); delete s; return TF_OK; }
int tf_stream_sync(tfStream_t s) { TF_HIP(hipStreamSynchronize(tf_hs(s))); return TF_OK; }
int tf_device_sync(void) { TF_HIP(hipDeviceSynchronize()); return TF_OK; }

int tf_event_create(tfEvent_t* out) {
  TF_REQUIRE(out, "tf_event_create: null out pointer");
  tfEvent_st* e = new tfEvent_st;
  hipError_t r = hipEventCreate(&e->e);
  if (r != hipSuccess) { delete e; tf_set_error("hipEventCreate: %s", hipGetErrorString(r)); return (int)r; }
  *out = e;
  return TF_OK;
}
int tf_event_destroy(tfEvent_t e) { if (!e) return TF_OK; TF_HIP(hipEventDestroy(e->e)); delete e; return TF_OK; }
int tf_event_record(tfEvent_t e, tfStream_t s) { TF_REQUIRE(e, "null event"); TF_HIP(hipEventRecord(e->e, tf_hs(s))); return TF_OK; }
int tf_stream_wait_event(tfStream_t s, tfEvent_t e) {
  TF_REQUIRE(e, "tf_stream_wait_event: null event");
  TF_HIP(hipStreamWaitEvent(tf_hs(s), e->e, 0));
  return TF_OK;
}
int tf_event_sync(tfEvent_t e) { TF_REQUIRE(e, "null event"); TF_HIP(hipEventSynchronize(e->e)); return TF_OK; }
int tf_event_elapsed_ms(float* ms, tfEvent_t a, tfEvent_t b) {
  TF_REQUIRE(ms && a && b, "tf_event_elapsed_ms: null argument");
  TF_HIP(hipEventElapsedTime(ms, a->e, b->e));
  return TF_OK;
}

int tf_graph_begin_capture(tfStream_t s) {
  TF_REQUIRE(s, "tf_graph_begin_capture: needs an explicit stream (the NULL stream cannot be captured)");
  TF_HIP(hipStreamBeginCapture(s->s, hipStreamCaptureModeThreadLocal));
  return TF_OK;
}
int tf_graph_end_capture(tfStream_t s, tfGraph_t* out) {
  TF_REQUIRE(s && out, "tf_graph_end_capture: null argument");
  tfGraph_st* g = new tfGraph_st;
  g->g = nullptr; g->x = nullptr;
  hipError_t e = hipStreamEndCapture(s->s, &g->g);
  if (e != hipSuccess) { delete g; tf_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return (int)e; }
  e = hipGraphInstantiate(&g->x, g->g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g->g); delete g; tf_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return (int)e; }
  *out = g;
  return TF_OK;
}
int tf_graph_abort_capture(tfStream_t s) {
  // leave capture mode after a failed capture: whatever was recorded is discarded; a stream that is not capturing is left alone
  TF_REQUIRE(s, "tf_graph_abort_capture: needs the captured stream");
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s->s, &cs);
  if (cs == hipStreamCaptureStatusNone) return TF_OK;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(s->s, &g);      // an invalidated capture returns an error AND leaves capture mode
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  (void)e;
  return TF_OK;
}
int tf_graph_launch(tfGraph_t g, tfStream_t s) { TF_REQUIRE(g, "null graph"); TF_HIP(hipGraphLaunch(g->x, tf_hs(s))); return TF_OK; }
int tf_graph_destroy(tfGraph_t g) {
  if (!g) return TF_OK;
  if (g->x) TF_HIP(hipGraphExecDestroy(g->x));
  if (g->g) TF_HIP(hipGraphDestroy(g->g));
  delete g;
  return TF_OK;
}

}  // extern "C"
