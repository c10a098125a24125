// Run-time kernel compilation: the MI355X counterpart of the reference's own-runtime loader (storage/device.py:31-77: nvrtcCreateProgram ->
// nvrtcCompileProgram(--gpu-architecture=sm_XY) -> nvrtcGetCUBIN -> cuModuleLoadData -> cuModuleGetFunction, then cuLaunchKernel in the
// wrappers :79-233, through native/nvrtc/ops.py:3-45 and native/cuda/ops.py:3-39).  Here: hiprtc compiles HIP source for the device's own
// architecture (gfx950), the code object is loaded with hipModuleLoadData and launched with hipModuleLaunchKernel.  libhiprtc is opened on
// first use (dlopen), so a host that never compiles at run time does not depend on it.  Nothing on the denoising path goes through here:
// its kernels ship pre-built in this library.
#include "common.h"
#include "../../include/tinyfusers_hip.h"
#include <dlfcn.h>
#include <string.h>
#include <string>
#include <vector>

namespace {
typedef void* rtcProgram;
struct Rtc {
  void* h = nullptr;
  int (*create)(rtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*compile)(rtcProgram, int, const char**) = nullptr;
  int (*log_size)(rtcProgram, size_t*) = nullptr;
  int (*get_log)(rtcProgram, char*) = nullptr;
  int (*code_size)(rtcProgram, size_t*) = nullptr;
  int (*get_code)(rtcProgram, char*) = nullptr;
  int (*destroy)(rtcProgram*) = nullptr;
  const char* (*err)(int) = nullptr;
} g_rtc;

int rtc_open() {
  if (g_rtc.h) return TF_OK;
  const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
  if (!h) { tf_set_error("tf_rtc_load: cannot open libhiprtc.so (%s)", dlerror()); return TF_E_STATE; }
  g_rtc.create = (decltype(g_rtc.create))dlsym(h, "hiprtcCreateProgram");
  g_rtc.compile = (decltype(g_rtc.compile))dlsym(h, "hiprtcCompileProgram");
  g_rtc.log_size = (decltype(g_rtc.log_size))dlsym(h, "hiprtcGetProgramLogSize");
  g_rtc.get_log = (decltype(g_rtc.get_log))dlsym(h, "hiprtcGetProgramLog");
  g_rtc.code_size = (decltype(g_rtc.code_size))dlsym(h, "hiprtcGetCodeSize");
  g_rtc.get_code = (decltype(g_rtc.get_code))dlsym(h, "hiprtcGetCode");
  g_rtc.destroy = (decltype(g_rtc.destroy))dlsym(h, "hiprtcDestroyProgram");
  g_rtc.err = (decltype(g_rtc.err))dlsym(h, "hiprtcGetErrorString");
  if (!g_rtc.create || !g_rtc.compile || !g_rtc.log_size || !g_rtc.get_log || !g_rtc.code_size || !g_rtc.get_code || !g_rtc.destroy) {
    tf_set_error("tf_rtc_load: libhiprtc.so lacks a required symbol");
    dlclose(h);
    return TF_E_STATE;
  }
  g_rtc.h = h;
  return TF_OK;
}
}  // namespace

extern "C" {

// Compile `source` (HIP C++; kernels declared extern "C" __global__ keep their names) for the current device, load the code object and
// return the kernel `func_name` (storage/device.py:31-77 in one call).  A failed compilation returns TF_E_ARG with the head of the
// compiler's log in tf_last_error().  The module stays loaded for the life of the process (the reference caches the function handle too).
int tf_rtc_load(tfFunction_t* out_fn, const char* source, const char* func_name) {
  TF_REQUIRE(out_fn && source && func_name, "tf_rtc_load: null argument");
  int rc = rtc_open();
  if (rc) return rc;
  int dev = 0;
  TF_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  TF_HIP(hipGetDeviceProperties(&prop, dev));
  std::string arch = std::string("--offload-arch=") + prop.gcnArchName;      // e.g. gfx950:sramecc+:xnack-
  const char* opts[] = {arch.c_str(), "-O3", "-std=c++17"};
  rtcProgram prog = nullptr;
  int r = g_rtc.create(&prog, source, "tf_rtc.hip", 0, nullptr, nullptr);
  if (r) { tf_set_error("hiprtcCreateProgram failed with status %d (%s)", r, g_rtc.err ? g_rtc.err(r) : "?"); return TF_E_STATE; }
  r = g_rtc.compile(prog, 3, opts);
  if (r) {
    size_t n = 0;
    std::string log;
    if (g_rtc.log_size(prog, &n) == 0 && n > 1) { log.resize(n); g_rtc.get_log(prog, &log[0]); }
    tf_set_error("hiprtcCompileProgram failed with status %d: %.380s", r, log.c_str());
    g_rtc.destroy(&prog);
    return TF_E_ARG;
  }
  size_t n = 0;
  std::vector<char> code;
  r = g_rtc.code_size(prog, &n);
  if (!r) { code.resize(n); r = g_rtc.get_code(prog, code.data()); }
  g_rtc.destroy(&prog);
  if (r || code.empty()) { tf_set_error("hiprtcGetCode failed with status %d", r); return TF_E_STATE; }
  // the code object stays alive with its module (one entry per loaded kernel: the loader may keep pointers into the image), and the
  // modules are unloaded when the library goes away -- a Device that recompiles many sources no longer leaks them silently
  struct Loaded { hipModule_t mod; std::vector<char> image; };
  static std::vector<Loaded*> g_loaded;
  Loaded* L = new Loaded{nullptr, std::move(code)};
  hipError_t le = hipModuleLoadData(&L->mod, L->image.data());
  if (le != hipSuccess) { tf_set_error("hipModuleLoadData failed: %s", hipGetErrorString(le)); delete L; return (int)le; }
  g_loaded.push_back(L);
  hipModule_t mod = L->mod;
  hipFunction_t fn;
  hipError_t e = hipModuleGetFunction(&fn, mod, func_name);
  if (e != hipSuccess) {
    tf_set_error("hipModuleGetFunction(%s) failed: %s", func_name, hipGetErrorString(e));
    (void)hipModuleUnload(mod); (void)hipGetLastError();
    g_loaded.pop_back(); delete L;
    return TF_E_ARG;
  }
  *out_fn = (tfFunction_t)fn;
  return TF_OK;
}

// cuLaunchKernel of the reference's wrappers (storage/device.py:89-99 ...): params[i] points at the i-th kernel argument
int tf_rtc_launch(tfFunction_t fn, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned shared_bytes,
                  tfStream_t s, void** params) {
  TF_REQUIRE(fn, "tf_rtc_launch: null function");
  TF_REQUIRE(gx && gy && gz && bx && by && bz && (unsigned long long)bx * by * bz <= 1024, "tf_rtc_launch: bad launch geometry (%u,%u,%u)x(%u,%u,%u)", gx, gy, gz, bx, by, bz);
  if (shared_bytes > 64 * 1024) {                            // more dynamic LDS than the default limit: raise the kernel's attribute (160 KiB per CU on gfx950)
    TF_REQUIRE(shared_bytes <= 160 * 1024, "tf_rtc_launch: %u bytes of dynamic LDS exceed the 160 KiB of a CU", shared_bytes);
    TF_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shared_bytes));
  }
  TF_HIP(hipModuleLaunchKernel((hipFunction_t)fn, gx, gy, gz, bx, by, bz, shared_bytes, tf_hs(s), params, nullptr));
  return TF_OK;
}

}  // extern "C"
