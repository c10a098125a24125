"""In-tree build of libtinyfusers_hip.so (hipcc, gfx950 only).  `python -m tinyfusers_amd.build [-f] [--ablation] [--asan-host]`.

`--asan-host` builds lib/libtinyfusers_hip_asan.so: the HOST side of every translation unit with -fsanitize=address,undefined (device code
unsanitised: -fno-gpu-sanitize; GPU sanitizers are not available on this pool) -- the table loader, run_gemm's host logic, the status paths,
tf_rtc_* / tf_comm_* argument checks under ASan + UBSan.  CPU container only (`sanitizer_env()` gives the environment a Python process needs to
load it: the ASan runtime preloaded, leak checking off -- CPython is not instrumented); never loaded by default, never taken to the GPU box.

`--ablation` builds a SECOND library, lib/libtinyfusers_hip_ablation.so, from the same sources with -DTF_ABLATION: it additionally holds the
ablation instances (kernels that skip work and return wrong results by design) that tools/*_dbg.py time; those tools load it explicitly
through TF_LIB_PATH.  The shipped library has none of them and ignores TF_SDPA_DBG / the ablation bits of tf_gemm_debug."""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtinyfusers_hip.so")
LIB_ABLATION = os.path.join(LIBDIR, "libtinyfusers_hip_ablation.so")
LIB_ASAN = os.path.join(LIBDIR, "libtinyfusers_hip_asan.so")
ASAN_FLAGS = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-fno-sanitize-recover=undefined"]
# the GEMM family is one translation unit per kernel family (gemm_k_*.hip) so that its instances compile in parallel; the largest first
SOURCES = ["gemm_k_pp16.hip", "gemm_k_pp16_bf16.hip", "gemm_k_pp8.hip", "gemm_k_igemm_128.hip", "gemm_k_igemm_128_bf16.hip", "gemm_k_igemm_64.hip", "gemm_k_igemm_64_bf16.hip",
           "gemm_k_igemm_160.hip", "gemm_k_igemm_160_bf16.hip", "gemm_k_patch.hip", "gemm_k_patch_bf16.hip", "sdpa.hip", "sdpa_bf16.hip",
           "gemm_k_igemm8.hip", "gemm_k_c4.hip", "gemm_k_c4_bf16.hip", "gemm_k_c8.hip", "gemm_k_c8_bf16.hip", "gemm_k_ar.hip", "gemm_k_ar_bf16.hip", "gemm_k_pp3.hip", "gemm_k_pp3_bf16.hip", "gemm.hip", "norm.hip", "elementwise.hip", "runtime.hip", "sgemm.hip",
           "comm.hip", "rtc.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-ffp-contract=fast"]
# sdpa.hip: keep the MFMA accumulators in VGPRs (the softmax reads every score: no v_accvgpr_read traffic) and drop
# the NaN-canonicalising v_max in front of every fmaxf on MFMA outputs (scores are never NaN; -inf masks still work)
EXTRA = {"sdpa.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"], "sdpa_bf16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def sanitizer_env(base=None):
    """Environment for a (non-instrumented) Python process that loads the --asan-host library: TF_LIB_PATH, the ASan runtime preloaded."""
    import glob
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        raise RuntimeError("libclang_rt.asan-x86_64.so not found under /opt/rocm/lib/llvm")
    env = dict(os.environ if base is None else base)
    env.update(TF_LIB_PATH=LIB_ASAN, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:detect_odr_violation=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    return env


def build(force=False, verbose=False, ablation=False, tag=None, defs=(), asan_host=False):
    """tag / defs: an EXPERIMENTAL build for same-box A/B runs (tools/ab_lib.sh, TF_LIB_PATH): the sources compiled with extra -D definitions into
    lib/<tag>/ and lib/libtinyfusers_hip_<tag>.so (`python -m tinyfusers_amd.build --tag prio1 -DTF_PP_PRIO=1`); never loaded by default."""
    objdir = os.path.join(LIBDIR, "asan") if asan_host else os.path.join(LIBDIR, "ablation") if ablation else os.path.join(LIBDIR, tag) if tag else LIBDIR
    lib = LIB_ASAN if asan_host else LIB_ABLATION if ablation else os.path.join(LIBDIR, f"libtinyfusers_hip_{tag}.so") if tag else LIB
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))] + [os.path.join(os.path.dirname(HERE), "include", "tinyfusers_hip.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        twin = [os.path.join(CSRC, s.replace("_bf16.hip", ".hip"))] if s.endswith("_bf16.hip") else []
        if force or _stale(obj, [src] + twin + headers):
            jobs.append([hipcc] + FLAGS + (["-DTF_ABLATION"] if ablation else []) + (ASAN_FLAGS if asan_host else []) + list(defs) + EXTRA.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
    with cf.ThreadPoolExecutor(max_workers=min(int(os.environ.get("TF_BUILD_JOBS", "8")), max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + (["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan"] if asan_host else []) + ["-o", lib] + objs)
    return lib


if __name__ == "__main__":
    _tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else None
    print(build(force="-f" in sys.argv, verbose=True, ablation="--ablation" in sys.argv, tag=_tag, defs=[a for a in sys.argv[1:] if a.startswith("-D")], asan_host="--asan-host" in sys.argv))
