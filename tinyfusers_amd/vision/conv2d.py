"""conv_2d / Conv2d -- mirrors tinyfusers/vision/conv2d.py:9-58 (cuDNN conv_fprop graph rebuilt per call,
NHWC->NCHW re-view, separate bias kernel).  Here: one implicit-GEMM MFMA launch on NHWC fp16 with bias, the
time-embedding add, the residual add, the nearest-2x upsample and the channel concat all folded in."""
import ctypes
import functools
import math
import operator

import numpy as np

from ..native import hip
from ..storage.tensor import DeviceArray, _sh, asarray
from ..ff.linear import workspace, linear_f16


def _conv(x, w, bias, padding, stride, dilation, bias_nc=None, residual=None, upsample=False, gn=0):
    x2 = None
    if isinstance(x, (tuple, list)):
        x, x2 = x
    assert tuple(dilation) == (1, 1), "dilation 1 only (all the UNet uses)"
    assert stride[0] == stride[1] and padding[0] == padding[1]
    n, c1, h, wd = x.shape
    c2 = x2.shape[1] if x2 is not None else 0
    k, c, r, s = w.shape
    assert c == c1 + c2, (w.shape, x.shape)
    up = 1 if upsample else 0
    ho = ((h << up) + 2 * padding[0] - r) // stride[0] + 1
    wo = ((wd << up) + 2 * padding[1] - s) // stride[1] + 1
    y = DeviceArray.empty((n, k, ho, wo), np.float16, "nhwc")
    nb = hip.tf_conv2d_workspace(n, h, wd, c1, c2, k, r, s, stride[0], padding[0], up)
    ws = workspace(nb)
    bnc_stride = 0
    if bias_nc is not None:
        bnc_stride = k if bias_nc.size // k > 1 else 0
    args = (y.ptr, x.ptr, x2.ptr if x2 is not None else None, w.ptr, bias.ptr if bias is not None else None,
            bias_nc.ptr if bias_nc is not None else None, bnc_stride, residual.ptr if residual is not None else None,
            n, h, wd, c1, c2, k, r, s, stride[0], padding[0], up, ws.ptr if ws else None, nb)
    if gn:
        # the GroupNorm(gn) that consumes y next gets its statistics from this conv's epilogue (when the shape allows)
        pb = hip.tf_conv2d_gn_partial_bytes(n, gn)
        part, chunks = workspace(pb), ctypes.c_int(0)
        hip.tf_conv2d_gn_f16(*args, part.ptr, pb, gn, ctypes.byref(chunks), _sh())
        if chunks.value > 0:
            y.gn = (part, chunks.value, gn)
    else:
        hip.tf_conv2d_f16(*args, _sh())
    return y


def _conv_small_c(x, w, bias, padding, stride, cache):
    """Cin % 8 != 0 (the 4-channel conv_in): im2col to K padded to 64, then the same GEMM kernel."""
    n, c, h, wd = x.shape
    k, _, r, s = w.shape
    kk = r * s * c
    kpad = (kk + 63) // 64 * 64
    ho = (h + 2 * padding[0] - r) // stride[0] + 1
    wo = (wd + 2 * padding[1] - s) // stride[1] + 1
    key = (w.ptr, kpad)
    if cache.get("key") != key:
        wp = DeviceArray.zeros((k, kpad), np.float16, "row")
        hip.tf_memcpy_2d_async(wp.ptr, kpad * 2, w.ptr, kk * 2, kk * 2, k, _sh())
        cache["key"], cache["w"] = key, wp
    col = DeviceArray.empty((n * ho * wo, kpad), np.float16, "row")
    hip.tf_im2col_nhwc_f16(col.ptr, x.ptr, n, h, wd, c, r, s, stride[0], padding[0], kpad, _sh())
    y = linear_f16(col, cache["w"], bias)
    return y.view((n, k, ho, wo), "nhwc")


def conv_2d(X_gpu, W_gpu, padding, stride, dilation):
    """vision/conv2d.py:9-28: NCHW cross-correlation, no bias.  X (N,C,H,W) and W (K,C,R,S) logical shapes."""
    if X_gpu.shape[1] % 8 != 0:
        return _conv_small_c(X_gpu, W_gpu, None, padding, stride, {})
    return _conv(X_gpu, W_gpu, None, padding, stride, dilation)


class Conv2d:
    def __init__(self, in_channels, out_channels, kernel_size, stride=[1, 1], padding=[0, 0], dilation=[1, 1], groups=1, bias=True, init=True):
        assert groups == 1, "groups=1 only (all the UNet uses)"
        self.kernel_size = kernel_size
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        shape = (out_channels, in_channels // groups, *kernel_size)
        self._shape = shape
        self.weight, self.bias = None, None
        if init:
            # reference init (vision/conv2d.py:52-54): U(-sqrt3, sqrt3) weight, U(-1/sqrt(fan_in), +) bias
            self.weight = asarray(np.random.uniform(-math.sqrt(3.0), math.sqrt(3.0), shape).astype(np.float16))
            bound = 1 / math.sqrt(functools.reduce(operator.mul, shape[1:], 1))
            self.bias = asarray(np.random.uniform(-bound, bound, (out_channels,)).astype(np.float16)) if bias else None
        self._cache = {}

    def __call__(self, x, bias_nc=None, residual=None, upsample=False, gn=0):
        """gn = G: also emit the statistics of the output for the GroupNorm(G) that reads it next (y.gn)."""
        cin = (x[0].shape[1] + x[1].shape[1]) if isinstance(x, (tuple, list)) else x.shape[1]
        if cin % 8 != 0:
            assert bias_nc is None and residual is None and not upsample
            return _conv_small_c(x, self.weight, self.bias, self.padding, self.stride, self._cache)
        return _conv(x, self.weight, self.bias, self.padding, self.stride, self.dilation, bias_nc, residual, upsample, gn)
