"""Thin ctypes shim over libtinyfusers_hip.so -- the MI355X replacement for the reference's
``tinyfusers/native`` package (ctypes classes over libcuda/libcudart/libnvrtc/libcublas,
native/cuda/ops.py:3-67, native/cublas/ops.py:3-53, native/nvrtc/ops.py:3-45).

Same conventions as the reference: every C entry returns an int status and callers raise
``RuntimeError("<fn> failed with status N")`` on non-zero (storage/device.py:33-37).  The library
is located relative to ``__file__`` (fixes SURVEY D9) and loading fails loudly -- there is no CPU
fallback anywhere in this package.
"""
from .hip import hip, lib, check, LIB_PATH, declared_symbols  # noqa: F401
