"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol that
include/tinyfusers_hip.h declares; argument validation and the status -> RuntimeError convention work."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__
    __graft_entry__.build()          # hipcc cross-compiles gfx950 without a GPU
    import tinyfusers_amd.native as n
    return n


def test_header_symbols_exported(native):
    hdr = open(os.path.join(ROOT, "include", "tinyfusers_hip.h")).read()
    names = sorted(set(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S))))
    assert len(names) >= 55
    assert sorted(native.declared_symbols()) == names
    dll = ctypes.CDLL(native.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), f"{n} declared in the header but not exported by the library"


def test_every_entry_cites_the_reference():
    hdr = open(os.path.join(ROOT, "include", "tinyfusers_hip.h")).read()
    # each block of prototypes is preceded by a comment naming the reference file:line it replaces
    assert len(re.findall(r"[a-z_/]+\.(?:py|cu):\d+", hdr)) >= 30


def test_status_convention_without_gpu(native):
    n = ctypes.c_int(-1)
    rc = native.lib.tf_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible: error-path check not applicable")
    assert n.value == 0
    with pytest.raises(RuntimeError, match=r"tf_init failed with status \d+"):
        native.hip.tf_init(0)
    assert native.lib.tf_last_error()          # human-readable reason travels with the status
    # argument validation happens before any device work
    assert native.lib.tf_linear_f16(None, None, None, None, None, 4, 4, 8, 0, None, 0, None) == 10001
    assert b"null tensor" in native.lib.tf_last_error()
    assert native.lib.tf_sdpa_f16(*([ctypes.c_void_p(8)] * 4), 1, 1, 4, 4, 12, *([8] * 12), 0, None) == 10001   # HS % 8
    assert native.lib.tf_version() >= 100


def test_no_cpu_fallback(native, monkeypatch):
    """The product path must fail loudly when the HIP extension is missing."""
    import sys
    h = sys.modules["tinyfusers_amd.native.hip"]     # the module (the package attribute `hip` is the shim instance)
    monkeypatch.setattr(h, "LIB_PATH", "/nonexistent/libtinyfusers_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        h._load()


def test_product_never_imports_the_oracle():
    import subprocess, sys
    out = subprocess.run(["grep", "-rlE", r"^\s*(from|import)\s+oracle", os.path.join(ROOT, "tinyfusers_amd")], capture_output=True, text=True)
    assert out.stdout.strip() == "", out.stdout


def test_shipped_library_holds_no_ablation_kernel(native):
    """Kernels that skip work and return wrong results by design (the DBG template instances tools/*_dbg.py time) are compiled only under
    -DTF_ABLATION into a second library; the shipped code objects hold none, no environment variable can select one, and the ablation
    bits of tf_gemm_debug are refused (one code path per op, as attention/sdpa.py:53-77 of the reference has)."""
    import sys
    sys.path.insert(0, ROOT)
    from tools.kernel_regs import LLVM, census
    if not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")):
        pytest.skip("llvm tools not found")
    ks = census(os.path.dirname(native.LIB_PATH))
    assert 200 <= len(ks) <= 400, len(ks)                       # (round 5: every MFMA kernel in float16 and in bfloat16)
    for k in ks:
        m = re.match(r"void k_igemm_pp<(\d+), (\d+), (true|false), (true|false),", k["name"])
        assert not (m and m.group(4) == "true"), k["name"]                    # k_igemm_pp<BN, NP, FASTA, DBG, ...>
        m = re.match(r"void k_sdpa_dma<(\d+), (\d+), (\d+), (\d+), (true|false)>", k["name"])
        assert not k["name"].startswith("void k_sdpa_dma<") or (m and m.group(3) == "0"), k["name"]   # k_sdpa_dma<HS, QT, DBG, NW, BF>
        assert k["vspill"] == 0 and k["scratch"] == 0, k                      # no kernel spills vector registers
    assert native.lib.tf_gemm_debug(1) == 10001 and b"TF_ABLATION" in native.lib.tf_last_error()
    assert native.lib.tf_gemm_debug(512) == 0 and native.lib.tf_gemm_debug(0) == 0   # (variant selection is a test hook, not an ablation)
    src = open(os.path.join(ROOT, "tinyfusers_amd", "csrc", "sdpa.hip")).read()
    assert re.search(r"#ifdef TF_ABLATION\s*\nstatic int g_sdpa_dbg = getenv", src)   # TF_SDPA_DBG is read by the ablation build only


def test_host_code_is_clean_under_asan_ubsan():
    """SURVEY 5 "race detection / sanitizers": the host side of the library (table loader, run_gemm's tile logic, shape predicates, status
    paths, tf_rtc_* / tf_comm_* argument handling) under AddressSanitizer + UndefinedBehaviorSanitizer -- `python -m tinyfusers_amd.build
    --asan-host` builds lib/libtinyfusers_hip_asan.so (device code unsanitised; CPU container only), tests/aux/host_sanitizer_drive.py drives
    it in a child process with the ASan runtime preloaded; any finding aborts the child.  Skips where that library has not been built."""
    import subprocess
    import sys
    from tinyfusers_amd import build as b
    if not os.path.exists(b.LIB_ASAN):
        pytest.skip("lib/libtinyfusers_hip_asan.so not built (python -m tinyfusers_amd.build --asan-host)")
    srcs = [os.path.join(b.CSRC, f) for f in os.listdir(b.CSRC)]
    if any(os.path.getmtime(f) > os.path.getmtime(b.LIB_ASAN) for f in srcs):
        pytest.skip("lib/libtinyfusers_hip_asan.so is older than the sources: rebuild it with --asan-host")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "aux", "host_sanitizer_drive.py")], env=b.sanitizer_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "HOST_SANITIZER_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]
