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
