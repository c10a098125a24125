"""conv_2d / Conv2d -- mirrors tinyfusers/vision/conv2d.py:9-58 (cuDNN conv_fprop graph rebuilt per call,
NHWC->NCHW re-view, separate bias kernel).  Here: one implicit-GEMM MFMA launch on NHWC fp16 with bias, the
time-embedding add, the residual add, the nearest-2x upsample and the channel concat all folded in."""
import ctypes
import functools
import math
import operator

import numpy as np

from .. import config
from ..native import hip, lib
from ..storage.tensor import DeviceArray, _sh, asarray, dtag, is_bfloat16
from ..ff.linear import workspace, linear_f16


_gi_support = {}


def _gn_in_args(x, x2, norm, k, r, s, stride, pad, up, c3, c4):
    """Arguments of tf_conv2d_gn_f16 for applying GroupNorm ``norm`` to the conv's input inside the launch, or None when the
    statistics did not come with the input(s) or the geometry cannot carry it (the caller then runs the norm on its own)."""
    if not config.fuse_group_norm or x.gn is None or (x2 is not None and x2.gn is None) or k < 64:
        return None
    if r != 1 and not config.fuse_group_norm_3x3:
        return None
    n, c1, h, wd = x.shape
    c2 = x2.shape[1] if x2 is not None else 0
    if n * h * wd * (c1 + c2) > config.fuse_group_norm_max_elems:      # (large inputs: the apply launch is cheaper than normalising per n-tile)
        return None
    G = norm.num_groups
    if x2 is None:
        if x.gn[2] != G:
            return None
        stats = (x.gn[0].ptr, x.gn[1], G, None, 0, 0)
    else:
        g1, g2 = x.gn[2], x2.gn[2]
        cpg = (c1 + c2) // G
        if not (config.concat_stats and c1 % g1 == 0 and c2 % g2 == 0 and c1 // g1 == c2 // g2 and cpg % (c1 // g1) == 0 and cpg // (c1 // g1) <= 8):
            return None
        stats = (x.gn[0].ptr, x.gn[1], g1, x2.gn[0].ptr, x2.gn[1], g2)
    key = (n, h, wd, c1, c2, k, r, s, stride, pad, up, c3, c4, G)
    ok = _gi_support.get(key)
    if ok is None:
        ok = _gi_support[key] = bool(lib.tf_conv2d_gn_supported(*key))
    if not ok:
        return None
    return (norm.weight.ptr if norm.weight is not None else None, norm.bias.ptr if norm.bias is not None else None, *stats, G, float(norm.eps))


def _conv(x, w, bias, padding, stride, dilation, bias_nc=None, residual=None, upsample=False, gn=0, extra=None, out=None, gn_in=None, out_norm=None):
    """gn_in = (GroupNorm module, silu): the conv reads GroupNorm(x) [-> SiLU]; applied inside the conv launch when the statistics came
    with x (x.gn) and the geometry allows (tf_conv2d_gn_f16), as a GroupNorm launch in front otherwise.
    out_norm = (GroupNorm module, silu): the module that reads y next; when the shape runs split-K its reduce kernel also writes the
    normalised tensor (y.normed), and that module's call returns it without a launch (tf_conv2d_fused_norm_f16)."""
    if gn_in is not None:
        norm, silu = gn_in
        xa, xb = x if isinstance(x, (tuple, list)) else (x, None)
        nd = xa.normed if xb is None else None
        if nd is not None and nd[0] is norm and nd[1] == bool(silu):
            return _conv(nd[2], w, bias, padding, stride, dilation, bias_nc, residual, upsample, gn, extra, out, None, out_norm)
        if extra is not None:
            kk, rr, ss = extra["cout"], extra["r"], extra["s"]
            e3, e4 = extra["x"] if isinstance(extra["x"], (tuple, list)) else (extra["x"], None)
            cc3, cc4 = e3.shape[1], (e4.shape[1] if e4 is not None else 0)
        else:
            kk, _, rr, ss = w.shape
            cc3 = cc4 = 0
        gi = _gn_in_args(xa, xb, norm, kk, rr, ss, stride[0], padding[0], 1 if upsample else 0, cc3, cc4) if not upsample else None
        if gi is None:
            return _conv(norm(x, silu=silu), w, bias, padding, stride, dilation, bias_nc, residual, upsample, gn, extra, out, None, out_norm)
    else:
        gi = None
    x2 = None
    if isinstance(x, (tuple, list)):
        x, x2 = x
    assert tuple(dilation) == (1, 1), "dilation 1 only (all the UNet uses)"
    assert stride[0] == stride[1] and padding[0] == padding[1]
    n, c1, h, wd = x.shape
    c2 = x2.shape[1] if x2 is not None else 0
    x3 = x4 = None
    c3 = c4 = 0
    if extra is not None:
        # w is then a packed (K, r*s*(c1+c2) + c3 + c4) row matrix: conv rows followed by the 1x1 rows of the extra sources
        k, r, s = extra["cout"], extra["r"], extra["s"]
        x3, x4 = extra["x"] if isinstance(extra["x"], (tuple, list)) else (extra["x"], None)
        c3, c4 = x3.shape[1], (x4.shape[1] if x4 is not None else 0)
        assert w.shape == (k, r * s * (c1 + c2) + c3 + c4), (w.shape, k, r, s, c1, c2, c3, c4)
    else:
        k, c, r, s = w.shape
        assert c == c1 + c2, (w.shape, x.shape)
    up = 1 if upsample else 0
    ho = ((h << up) + 2 * padding[0] - r) // stride[0] + 1
    wo = ((wd << up) + 2 * padding[1] - s) // stride[1] + 1
    dt = dtag(x.dtype)
    assert dtag(w.dtype) == dt, "conv: activations and weights must hold the same 16-bit type"
    y = out if out is not None else DeviceArray.empty((n, k, ho, wo), x.dtype, "nhwc")
    assert y.shape == (n, k, ho, wo) and y.layout == "nhwc"
    nb = hip.tf_conv2d_fused_workspace(n, h, wd, c1, c2, k, r, s, stride[0], padding[0], up, c3, c4)
    ws = workspace(nb)
    bnc_stride = 0
    if bias_nc is not None:
        bnc_stride = k if bias_nc.size // k > 1 else 0
    args = (y.ptr, x.ptr, x2.ptr if x2 is not None else None, w.ptr, bias.ptr if bias is not None else None,
            bias_nc.ptr if bias_nc is not None else None, bnc_stride, residual.ptr if residual is not None else None,
            n, h, wd, c1, c2, k, r, s, stride[0], padding[0], up, ws.ptr if ws else None, nb)
    ex = (x3.ptr if x3 is not None else None, x4.ptr if x4 is not None else None, c3, c4)
    if gi is not None:
        pb, part, chunks = 0, None, ctypes.c_int(0)
        if gn:
            pb = hip.tf_conv2d_gn_partial_bytes(n, gn)
            part = workspace(pb)
        hip.tf_conv2d_gn_16(dt, *args, *ex, part.ptr if part is not None else None, pb, gn, ctypes.byref(chunks), *gi, 1 if gn_in[1] else 0, _sh())
        if chunks.value > 0:
            y.gn = (part, chunks.value, gn)
        y._base = (y._base, x.gn[0], x2.gn[0] if x2 is not None else None)   # the statistics stay referenced while the launch is queued
    elif gn and out_norm is not None and config.fuse_reduce_norm and out_norm[0].num_groups == gn and out is None:
        # ... and, behind a split-K shape, is applied by the reduce kernel as well: z rides along with y
        pb = hip.tf_conv2d_gn_partial_bytes(n, gn)
        part, chunks, zw = workspace(pb), ctypes.c_int(0), ctypes.c_int(0)
        z = DeviceArray.empty((n, k, ho, wo), x.dtype, "nhwc")
        nm = out_norm[0]
        hip.tf_conv2d_fused_norm_16(dt, *args, *ex, part.ptr, pb, gn, ctypes.byref(chunks), z.ptr, nm.weight.ptr if nm.weight is not None else None,
                                     nm.bias.ptr if nm.bias is not None else None, float(nm.eps), 1 if out_norm[1] else 0, ctypes.byref(zw), _sh())
        if chunks.value > 0:
            y.gn = (part, chunks.value, gn)
        if zw.value:
            y.normed = (nm, bool(out_norm[1]), z)
    elif gn:
        # the GroupNorm(gn) that consumes y next gets its statistics from this conv's epilogue (when the shape allows)
        pb = hip.tf_conv2d_gn_partial_bytes(n, gn)
        part, chunks = workspace(pb), ctypes.c_int(0)
        hip.tf_conv2d_fused_16(dt, *args, *ex, part.ptr, pb, gn, ctypes.byref(chunks), _sh())
        if chunks.value > 0:
            y.gn = (part, chunks.value, gn)
    else:
        hip.tf_conv2d_fused_16(dt, *args, *ex, None, 0, 0, None, _sh())
    return y


_BAND_BYTES = 1 << 30      # patch-matrix bytes per launch of the small-C path


def _conv_small_c(x, w, bias, padding, stride, cache, gn=0):
    """Cin % 8 != 0 (the 4-channel conv_in): im2col to K padded to 64, then the same GEMM kernel -- as a 1x1 convolution over the
    (n, ho, wo, kpad) patch image, so that the GroupNorm statistics of the output can ride along (gn) like for any other conv."""
    n, c, h, wd = x.shape
    k, _, r, s = w.shape
    kk = r * s * c
    kpad = (kk + 63) // 64 * 64
    ho = (h + 2 * padding[0] - r) // stride[0] + 1
    wo = (wd + 2 * padding[1] - s) // stride[1] + 1
    key = (w.wkey, kpad)
    if cache.get("key") != key:
        wp = DeviceArray.zeros((k, kpad), w.dtype, "row")
        hip.tf_memcpy_2d_async(wp.ptr, kpad * 2, w.ptr, kk * 2, kk * 2, k, _sh())
        cache["key"], cache["w"] = key, wp
    if n * ho * wo * kpad * 2 <= _BAND_BYTES:
        col = DeviceArray.empty((n * ho * wo, kpad), x.dtype, "row")
        hip.tf_im2col_nhwc_f16(col.ptr, x.ptr, n, h, wd, c, r, s, stride[0], padding[0], kpad, _sh())     # (a 2-byte gather: either element type)
        return _conv(col.view((n, kpad, ho, wo), "nhwc"), cache["w"].view((k, kpad, 1, 1), "nhwc"), bias, [0, 0], [1, 1], [1, 1], gn=gn)
    # large images (the reference's own conv test is 10000 x 10000, tests/conv2d.py:13-33): bands of output rows, image by image,
    # each band's patch matrix and GEMM operands far below the 2 GiB one buffer descriptor spans
    assert gn == 0, "banded small-C conv: no GroupNorm statistics"
    y = DeviceArray.empty((n, k, ho, wo), x.dtype, "nhwc")
    rows = max(1, _BAND_BYTES // (wo * kpad * 2))
    wv = cache["w"].view((k, kpad, 1, 1), "nhwc")
    for img in range(n):
        xi = x.view((1, c, h, wd), "nhwc", img * h * wd * c)
        for o0 in range(0, ho, rows):
            o1 = min(ho, o0 + rows)
            col = DeviceArray.empty(((o1 - o0) * wo, kpad), x.dtype, "row")
            hip.tf_im2col_rows_nhwc_f16(col.ptr, xi.ptr, 1, h, wd, c, r, s, stride[0], padding[0], kpad, o0, o1, _sh())
            _conv(col.view((1, kpad, o1 - o0, wo), "nhwc"), wv, bias, [0, 0], [1, 1], [1, 1],
                  out=y.view((1, k, o1 - o0, wo), "nhwc", (img * ho + o0) * wo * k))
    return y


def conv_2d(X_gpu, W_gpu, padding, stride, dilation):
    """vision/conv2d.py:9-28: NCHW cross-correlation, no bias.  X (N,C,H,W) and W (K,C,R,S) logical shapes."""
    if X_gpu.shape[1] % 8 != 0:
        return _conv_small_c(X_gpu, W_gpu, None, padding, stride, {})
    return _conv(X_gpu, W_gpu, None, padding, stride, dilation)


def conv2d_bf16(x, w, bias, padding, stride, dilation, residual=None, bias_nc=None, upsample=False):
    """The same operator on bfloat16 tensors (x NHWC -- or the concat pair (x, x2) --, w (K, R, S, C) stored): tf_conv2d_bf16, channel counts
    multiples of 8; bias, the time-embedding bias (bias_nc), the residual and the nearest-2x up-sampling folded in as for the fp16 conv."""
    x2 = None
    if isinstance(x, (tuple, list)):
        x, x2 = x
    n, c, h, wd = x.shape
    c2 = x2.shape[1] if x2 is not None else 0
    k, cw, r, s_ = w.shape
    assert c + c2 == cw and c % 8 == 0 and c2 % 8 == 0 and is_bfloat16(w.dtype), (x.shape, w.shape)
    assert list(dilation) == [1, 1] and stride[0] == stride[1] and padding[0] == padding[1]
    up = 1 if upsample else 0
    ho = ((h << up) + 2 * padding[0] - r) // stride[0] + 1
    wo = ((wd << up) + 2 * padding[1] - s_) // stride[1] + 1
    y = DeviceArray.empty((n, k, ho, wo), x.dtype, "nhwc")
    bnc_stride = 0
    if bias_nc is not None:
        bnc_stride = k if bias_nc.size // k > 1 else 0
    hip.tf_conv2d_bf16(y.ptr, x.ptr, x2.ptr if x2 is not None else None, w.ptr, bias.ptr if bias is not None else None,
                       bias_nc.ptr if bias_nc is not None else None, bnc_stride,
                       residual.ptr if residual is not None else None, n, h, wd, c, c2, k, r, s_, stride[0], padding[0], up, _sh())
    return y


def _conv_small_c_bf16(x, w, bias, padding, stride, cache):
    """Cin % 8 != 0 on bfloat16 (the 4-channel conv_in of the bfloat16 step): im2col (a 2-byte copy: the same kernel) to K padded to 64, then a
    1x1 bfloat16 conv over the patch image."""
    n, c, h, wd = x.shape
    k, _, r, s = w.shape
    kk = r * s * c
    kpad = (kk + 63) // 64 * 64
    ho = (h + 2 * padding[0] - r) // stride[0] + 1
    wo = (wd + 2 * padding[1] - s) // stride[1] + 1
    key = (w.wkey, kpad, "bf16")
    if cache.get("key") != key:
        wp = DeviceArray.zeros((k, kpad), w.dtype, "row")
        hip.tf_memcpy_2d_async(wp.ptr, kpad * 2, w.ptr, kk * 2, kk * 2, k, _sh())
        cache["key"], cache["w"] = key, wp
    col = DeviceArray.empty((n * ho * wo, kpad), x.dtype, "row")
    hip.tf_im2col_nhwc_f16(col.ptr, x.ptr, n, h, wd, c, r, s, stride[0], padding[0], kpad, _sh())
    return conv2d_bf16(col.view((n, kpad, ho, wo), "nhwc"), cache["w"].view((k, kpad, 1, 1), "nhwc"), bias, [0, 0], [1, 1], [1, 1])


def pad_image(x, left, right, top, bottom):
    """Zero padding of an NHWC image tensor on the device: the asymmetric [0, 1, 0, 1] (left, right, top, bottom) of the VAE encoder's
    stride-2 convolutions (vae/encoder.py:19 hands that list to the conv as its ``padding``; a symmetric conv cannot express it)."""
    n, c, h, w = x.shape
    assert x.layout == "nhwc" and x.dtype.itemsize == 2
    hp, wp = h + top + bottom, w + left + right
    y = DeviceArray.empty((n, c, hp, wp), x.dtype, "nhwc")
    hip.tf_memset_async(y.ptr, 0, y.nbytes, _sh())
    for i in range(n):
        hip.tf_memcpy_2d_async(y.ptr + ((i * hp + top) * wp + left) * c * 2, wp * c * 2, x.ptr + i * h * w * c * 2, w * c * 2, w * c * 2, h, _sh())
    return y


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
        # config 5 (fp8): only modules that opt in run on e4m3 operands -- the UNet's ResBlocks set this on their two 3x3 convs (the layer
        # policy is a precision budget: ff/fp8.py).  Everything else (VAE, 1x1 and up / down-sampling convs) stays fp16
        self._fp8_ok = False

    def fold_1x1(self, proj):
        """(w, bias) of this conv with the 1x1 conv ``proj`` folded in as extra K columns (see __call__'s ``extra``): packed
        once on the device, rebuilt when either module's weights are replaced."""
        key = (self.weight.wkey, self.bias.wkey, proj.weight.wkey, proj.bias.wkey)
        if self._cache.get("fold_key") != key:
            k, c, r, s = self.weight.shape
            kc, ce = r * s * c, proj.weight.shape[1]
            assert proj.weight.shape[0] == k and tuple(proj.weight.shape[2:]) == (1, 1)
            wp = DeviceArray.empty((k, kc + ce), self.weight.dtype, "row")
            hip.tf_memcpy_2d_async(wp.ptr, (kc + ce) * 2, self.weight.ptr, kc * 2, kc * 2, k, _sh())
            hip.tf_memcpy_2d_async(wp.ptr + kc * 2, (kc + ce) * 2, proj.weight.ptr, ce * 2, ce * 2, k, _sh())
            bp = DeviceArray.empty((k,), self.bias.dtype, "row")
            hip.tf_add_16(dtag(self.bias.dtype), bp.ptr, self.bias.ptr, proj.bias.ptr, k, _sh())
            self._cache["fold_key"], self._cache["fold"] = key, (wp, bp)
        return self._cache["fold"]

    def __call__(self, x, bias_nc=None, residual=None, upsample=False, gn=0, extra=None, gn_in=None, out_norm=None):
        """gn = G: also emit the statistics of the output for the GroupNorm(G) that reads it next (y.gn).
        extra = (proj, x3): add ``proj(x3)`` (a Conv2d 1x1; x3 a tensor or a concat pair) inside this conv's GEMM.
        gn_in = (GroupNorm, silu): this conv reads GroupNorm(x) [-> SiLU] (x is the RAW tensor): one launch where possible."""
        x0 = x[0] if isinstance(x, (tuple, list)) else x
        if len(self.padding) == 4:                         # [left, right, top, bottom]: pad explicitly, then an unpadded conv
            assert extra is None and gn_in is None and not upsample and not isinstance(x, (tuple, list))
            x = pad_image(x, *self.padding)
            return _conv(x, self.weight, self.bias, [0, 0], self.stride, self.dilation, bias_nc, residual, False, gn, out_norm=out_norm)
        if extra is not None:
            proj, x3 = extra
            wp, bp = self.fold_1x1(proj)
            k, _, r, s = self.weight.shape
            assert not isinstance(x, (tuple, list)) and x.shape[1] % 8 == 0
            return _conv(x, wp, bp, self.padding, self.stride, self.dilation, bias_nc, residual, upsample, gn,
                         {"x": x3, "cout": k, "r": r, "s": s}, gn_in=gn_in, out_norm=out_norm)
        cin = (x[0].shape[1] + x[1].shape[1]) if isinstance(x, (tuple, list)) else x.shape[1]
        from ..ff import fp8
        if self._fp8_ok and gn_in is not None and extra is None and fp8.conv_ok((x[0].shape[0], cin) + tuple(x[0].shape[2:]) if isinstance(x, (tuple, list)) else x.shape,
                                                                                 self.weight.shape, self.stride, self.padding, upsample):
            # config 5: e4m3 operands.  The activation operand comes straight out of the GroupNorm apply as a block-scaled tensor (the concat of the
            # output path arrives materialised); bias / time embedding / residual / statistics as in the fp16 conv
            w8, wsc = fp8.pack_weight(self.weight, self._cache)
            x8 = fp8.group_norm_mx(x, gn_in[0], gn_in[1])
            return fp8.conv2d_mx(x8, w8, wsc, self.bias, self.weight.shape, self.padding, bias_nc, residual, gn)
        if cin % 8 != 0:
            assert bias_nc is None and residual is None and not upsample
            if gn_in is not None:
                x = gn_in[0](x, silu=gn_in[1])
            return _conv_small_c(x, self.weight, self.bias, self.padding, self.stride, self._cache, gn)
        return _conv(x, self.weight, self.bias, self.padding, self.stride, self.dilation, bias_nc, residual, upsample, gn, gn_in=gn_in, out_norm=out_norm)
