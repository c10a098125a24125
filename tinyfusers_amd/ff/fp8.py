"""fp8 (OCP e4m3) conv / linear path for BASELINE config 5 ("SD1.5 768x768 batch 32 on 8 GPUs, fp8 MFMA conv/linear path") -- the
precision variant of vision/conv2d.py:9-58 and ff/linear.py:112-121, selected with ``config.set_dtype("fp8")``.

Operands (round 4): weights e4m3 with one fp32 scale per output channel (tf_pack_weight_fp8, packed once per weight set); ACTIVATIONS
block-scaled e4m3 ("mx8": every 32 consecutive channels of a pixel / token share one power-of-two E8M0 scale, amax / 2^e <= 448 -- so any
tensor may be quantised, not only a normalised one), written directly by the kernel that produces them (GroupNorm apply, LayerNorm, the
GEGLU epilogue) or by one quantise pass (``quantize_mx``); the GEMM (k_igemm_pp<F8>) hands the scale bytes to the hardware's scale operand
(v_mfma_scale_f32_16x16x128_f8f6f4).  An mx8 tensor is ONE buffer: codes, then the scale bytes (include/tinyfusers_hip.h).

Which layers (measured on the CPU oracle with the same quantisers, tests/fp8_policy_study.py; BASELINE.md section 4 gate: UNet rel-L2 <= 0.1):
e4m3 rounding noise adds up in quadrature whatever the scales -- every conv and linear in e4m3 gives 0.15 with block scales and with a
fixed scale alike -- so the policy is a precision budget, spent where the FLOPs are: the two 3x3 convs of every ResBlock (0.048), the
FeedForward pair (0.042 + 0.029) -- 0.068 together -- and, optionally (``ATTENTION``, off: slower, see below), the attention projections
q|k|v / q / to_out (0.018 + 0.010 + 0.022): 0.076 for all of them at 32 x 32, 0.094 at 64 x 64.  The 1x1 convs on the residual path (proj_in 0.060, proj_out 0.055, skip 0.092) and the up / down-sampling convs (0.056) stay fp16,
as do conv_in / conv_out, the time-embedding GEMVs and everything outside the UNet.  A layer also stays fp16 where the block-scaled kernel
cannot take its shape (tf_mx8_*_supported: it is the 192- / 256-row ping-pong kernel, i.e. launches that fill the chip -- config 5's
regime) or where the fp16 kernels are faster (K = 320: ``MIN_K``).

The fixed-scale entries of round 2 (``quantize``, ``conv2d_fp8``, ``linear_fp8``, ``group_norm_fp8``, ``layer_norm_fp8``: scale 1, k_igemm8)
remain as an op-level API; the model no longer uses them.
"""
import ctypes

import numpy as np

from .. import config
from ..native import hip, lib
from ..storage.tensor import DeviceArray, _sh
from .linear import workspace


def enabled():
    return config.dtype == "fp8"


def pack_weight(w, cache, tag="fp8"):
    """(N, K...) fp16 weight -> (e4m3 bytes of the same layout, fp32 scale per row); cached on ``cache`` by the weight's content key."""
    key = w.wkey
    hit = cache.get(tag)
    if hit is None or hit[0] != key:
        n = w.shape[0]
        k = w.size // n
        w8 = DeviceArray.empty((n, k), np.uint8, "row")
        sc = DeviceArray.empty((n,), np.float32, "row")
        hip.tf_pack_weight_fp8(w8.ptr, sc.ptr, w.ptr, n, k, _sh())
        cache[tag] = hit = (key, w8, sc)
    return hit[1], hit[2]


def quantize(x):
    """fp16 DeviceArray -> e4m3 DeviceArray of the same logical shape and layout (scale 1, saturating)."""
    y = DeviceArray.empty(x.shape, np.uint8, x.layout)
    hip.tf_quantize_fp8_f16(y.ptr, x.ptr, x.size, 1.0, _sh())
    return y


def group_norm_fp8(x, norm, silu):
    """GroupNorm(x) [-> SiLU] written as e4m3: one apply launch when the statistics came with x (and its concat partner), the fp16
    GroupNorm followed by a quantise pass otherwise."""
    x2 = None
    if isinstance(x, (tuple, list)):
        x, x2 = x
    n, c1, h, w = x.shape
    c2 = x2.shape[1] if x2 is not None else 0
    G = norm.num_groups
    gm = norm.weight.ptr if norm.weight is not None else None
    bt = norm.bias.ptr if norm.bias is not None else None
    if x2 is None and x.gn is not None and x.gn[2] == G:
        y = DeviceArray.empty((n, c1, h, w), np.uint8, "nhwc")
        hip.tf_group_norm_apply_fp8(y.ptr, x.ptr, None, gm, bt, x.gn[0].ptr, x.gn[1], G, None, 0, 0, n, h * w, c1, 0, G, float(norm.eps), 1 if silu else 0, _sh())
        return y
    if x2 is not None and x.gn is not None and x2.gn is not None and config.concat_stats:
        g1, g2 = x.gn[2], x2.gn[2]
        cpg = (c1 + c2) // G
        if c1 % g1 == 0 and c2 % g2 == 0 and c1 // g1 == c2 // g2 and cpg % (c1 // g1) == 0 and cpg // (c1 // g1) <= 8:
            y = DeviceArray.empty((n, c1 + c2, h, w), np.uint8, "nhwc")
            hip.tf_group_norm_apply_fp8(y.ptr, x.ptr, x2.ptr, gm, bt, x.gn[0].ptr, x.gn[1], g1, x2.gn[0].ptr, x2.gn[1], g2, n, h * w, c1, c2, G,
                                        float(norm.eps), 1 if silu else 0, _sh())
            return y
    return quantize(norm((x, x2) if x2 is not None else x, silu=silu))


def conv_eligible(weight_shape, cin_parts, extra):
    k, c, r, s = weight_shape
    return enabled() and extra is None and r == 3 and s == 3 and k >= 64 and all(cp % 64 == 0 for cp in cin_parts)


def conv2d_fp8(x8, w8, wscale, bias, weight_shape, padding, stride, bias_nc=None, residual=None, upsample=False, gn=0):
    """e4m3 activations (one array, or the concat pair) x e4m3 weights -> fp16 NHWC output; same epilogue as the fp16 conv."""
    x82 = None
    if isinstance(x8, (tuple, list)):
        x8, x82 = x8
    n, c1, h, wd = x8.shape
    c2 = x82.shape[1] if x82 is not None else 0
    k, c, r, s = weight_shape
    assert c == c1 + c2
    up = 1 if upsample else 0
    ho = ((h << up) + 2 * padding[0] - r) // stride[0] + 1
    wo = ((wd << up) + 2 * padding[1] - s) // stride[1] + 1
    y = DeviceArray.empty((n, k, ho, wo), np.float16, "nhwc")
    nb = hip.tf_conv2d_fp8_workspace(n, h, wd, c1, c2, k, r, s, stride[0], padding[0], up)
    ws = workspace(nb)
    bnc_stride = 0
    if bias_nc is not None:
        bnc_stride = k if bias_nc.size // k > 1 else 0
    part, pb, chunks = None, 0, ctypes.c_int(0)
    if gn:
        pb = hip.tf_conv2d_gn_partial_bytes(n, gn)
        part = workspace(pb)
    hip.tf_conv2d_fp8(y.ptr, x8.ptr, x82.ptr if x82 is not None else None, w8.ptr, wscale.ptr, bias.ptr if bias is not None else None,
                      bias_nc.ptr if bias_nc is not None else None, bnc_stride, residual.ptr if residual is not None else None,
                      n, h, wd, c1, c2, k, r, s, stride[0], padding[0], up, ws.ptr if ws else None, nb,
                      part.ptr if part is not None else None, pb, gn, ctypes.byref(chunks), _sh())
    if chunks.value > 0:
        y.gn = (part, chunks.value, gn)
    return y


def linear_fp8(x8, w8, wscale, bias, residual=None, act=0, out_features=None, out_fp8=False):
    """y = act(x8 . w8^T * wscale + bias) + residual; x8 (..., K) e4m3, w8 (N, K) e4m3; y fp16, or e4m3 (out_fp8) for the next fp8 GEMM."""
    K = x8.shape[-1]
    rows = x8.size // K
    n_out = out_features if out_features is not None else w8.shape[0]
    y = DeviceArray.empty(x8.shape[:-1] + (n_out,), np.uint8 if out_fp8 else np.float16, "row")
    nb = 0 if out_fp8 else hip.tf_linear_workspace(rows, n_out, K, act)
    ws = workspace(nb)
    hip.tf_linear_fp8(y.ptr, x8.ptr, w8.ptr, wscale.ptr, bias.ptr if bias is not None else None, residual.ptr if residual is not None else None,
                      rows, n_out, K, act, 1 if out_fp8 else 0, ws.ptr if ws else None, nb, _sh())
    return y


def layer_norm_fp8(x, ln):
    c = x.shape[-1]
    y = DeviceArray.empty(x.shape, np.uint8, x.layout)
    hip.tf_layer_norm_fp8(y.ptr, x.ptr, ln.weight.ptr if ln.weight is not None else None, ln.bias.ptr if ln.bias is not None else None,
                          x.size // c, c, float(np.asarray(ln.eps).reshape(-1)[0]), _sh())
    return y


# ---- block-scaled e4m3 ("mx8") -------------------------------------------------------------------------------------------------------
import os
# The attention projections (q|k|v, q, to_out at K >= MIN_K) on e4m3 operands as well: within the precision budget and built (tests/test_gpu_mx8.py),
# but MEASURED SLOWER at config 5 (profiles/r04_ab.txt: 23.78 against 23.59 ms per step): the LayerNorm can no longer be folded into the
# projection (one launch and one pass more per block) and to_out needs a quantise pass, which costs more than the shorter GEMMs give back.  Off.
ATTENTION = os.environ.get("TF_FP8_ATTN", "0") not in ("0", "")
MIN_K = int(os.environ.get("TF_FP8_MIN_K", "640"))        # below this the fp16 kernels win (LayerNorm folded into the persistent short-K kernel: 215 + 90 us against 293 + 74 us for the
                   # e4m3 FeedForward pair at config 5's first level) -- K = 320 layers stay fp16
_supported = {}


def mx_empty(shape, layout=None):
    """An mx8 tensor of the logical shape `shape` (channels = shape[1] for an NHWC image, shape[-1] for rows): codes + scale bytes in one
    allocation; the returned uint8 array views the codes, the scale bytes sit right behind them."""
    n = int(np.prod(shape, dtype=np.int64))
    c = shape[1] if len(shape) == 4 else shape[-1]
    assert c % 32 == 0, shape
    buf = DeviceArray.empty((n + n // 32,), np.uint8, "row")
    return buf.view(tuple(shape), layout or ("nhwc" if len(shape) == 4 else "row"))


def quantize_mx(x):
    """fp16 DeviceArray (NHWC image or rows) -> mx8 tensor of the same logical shape."""
    c = x.shape[1] if x.ndim == 4 else x.shape[-1]
    y = mx_empty(x.shape, x.layout)
    hip.tf_quantize_mx8_f16(y.ptr, x.ptr, x.size // c, c, _sh())
    return y


def linear_ok(rows, n, k, act=0, out_mx=False):
    """Does the block-scaled kernel take (and pay for) this Linear?"""
    if not enabled() or k < MIN_K:
        return False
    key = ("lin", rows, n, k, act, out_mx)
    if key not in _supported:
        _supported[key] = bool(lib.tf_mx8_gemm_supported(rows, n, k, act, 1 if out_mx else 0))
    return _supported[key]


def conv_ok(x_shape, weight_shape, stride, padding, upsample):
    n, c, h, w = x_shape
    k, cw, r, s = weight_shape
    if not enabled() or upsample or stride[0] != 1 or stride[1] != 1 or r != s or c != cw or c < 64 or k < 64:
        return False
    key = ("conv", n, h, w, c, k, r, padding[0])
    if key not in _supported:
        _supported[key] = bool(lib.tf_mx8_conv_supported(n, h, w, c, 0, k, r, s, 1, padding[0], 0))
    return _supported[key]


def group_norm_mx(x, norm, silu):
    """GroupNorm(x) [-> SiLU] written as an mx8 tensor: one apply launch when the statistics came with x (and its concat partner), the fp16
    GroupNorm followed by a quantise pass otherwise."""
    x2 = None
    if isinstance(x, (tuple, list)):
        x, x2 = x
    n, c1, h, w = x.shape
    c2 = x2.shape[1] if x2 is not None else 0
    G = norm.num_groups
    gm = norm.weight.ptr if norm.weight is not None else None
    bt = norm.bias.ptr if norm.bias is not None else None
    if x2 is None and x.gn is not None and x.gn[2] == G:
        y = mx_empty((n, c1, h, w), "nhwc")
        hip.tf_group_norm_apply_mx8(y.ptr, x.ptr, None, gm, bt, x.gn[0].ptr, x.gn[1], G, None, 0, 0, n, h * w, c1, 0, G, float(norm.eps), 1 if silu else 0, _sh())
        return y
    if x2 is not None and x.gn is not None and x2.gn is not None and config.concat_stats:
        g1, g2 = x.gn[2], x2.gn[2]
        cpg = (c1 + c2) // G
        if c1 % g1 == 0 and c2 % g2 == 0 and c1 // g1 == c2 // g2 and cpg % (c1 // g1) == 0 and cpg // (c1 // g1) <= 8:
            y = mx_empty((n, c1 + c2, h, w), "nhwc")
            hip.tf_group_norm_apply_mx8(y.ptr, x.ptr, x2.ptr, gm, bt, x.gn[0].ptr, x.gn[1], g1, x2.gn[0].ptr, x2.gn[1], g2, n, h * w, c1, c2, G,
                                        float(norm.eps), 1 if silu else 0, _sh())
            return y
    return quantize_mx(norm((x, x2) if x2 is not None else x, silu=silu))


def layer_norm_mx(x, ln):
    c = x.shape[-1]
    y = mx_empty(x.shape, x.layout)
    hip.tf_layer_norm_mx8(y.ptr, x.ptr, ln.weight.ptr if ln.weight is not None else None, ln.bias.ptr if ln.bias is not None else None,
                          x.size // c, c, float(np.asarray(ln.eps).reshape(-1)[0]), _sh())
    return y


def conv2d_mx(x8, w8, wscale, bias, weight_shape, padding, bias_nc=None, residual=None, gn=0):
    """mx8 activations (one tensor: a concat arrives materialised from group_norm_mx) x e4m3 weights -> fp16 NHWC output; stride 1; same
    epilogue as the fp16 conv (bias, time embedding, residual, the statistics of the output)."""
    n, c1, h, wd = x8.shape
    k, c, r, s = weight_shape
    assert c == c1
    ho, wo = h + 2 * padding[0] - r + 1, wd + 2 * padding[1] - s + 1
    y = DeviceArray.empty((n, k, ho, wo), np.float16, "nhwc")
    nb = hip.tf_conv2d_fp8_workspace(n, h, wd, c1, 0, k, r, s, 1, padding[0], 0)
    ws = workspace(nb)
    bnc_stride = 0
    if bias_nc is not None:
        bnc_stride = k if bias_nc.size // k > 1 else 0
    part, pb, chunks = None, 0, ctypes.c_int(0)
    if gn:
        pb = hip.tf_conv2d_gn_partial_bytes(n, gn)
        part = workspace(pb)
    hip.tf_conv2d_mx8(y.ptr, x8.ptr, None, w8.ptr, wscale.ptr, bias.ptr if bias is not None else None,
                      bias_nc.ptr if bias_nc is not None else None, bnc_stride, residual.ptr if residual is not None else None,
                      n, h, wd, c1, 0, k, r, s, 1, padding[0], ws.ptr if ws else None, nb,
                      part.ptr if part is not None else None, pb, gn, ctypes.byref(chunks), _sh())
    if chunks.value > 0:
        y.gn = (part, chunks.value, gn)
    return y


def linear_mx(x8, w8, wscale, bias, residual=None, act=0, out_features=None, out_mx=False):
    """y = act(x8 . w8^T * wscale + bias) + residual; x8 (..., K) mx8, w8 (N, K) e4m3; y fp16, or an mx8 tensor (out_mx: the GEGLU output that
    feeds the next block-scaled GEMM)."""
    K = x8.shape[-1]
    rows = x8.size // K
    n_out = out_features if out_features is not None else w8.shape[0]
    shape = x8.shape[:-1] + (n_out,)
    y = mx_empty(shape, "row") if out_mx else DeviceArray.empty(shape, np.float16, "row")
    nb = 0 if out_mx else hip.tf_linear_workspace(rows, n_out, K, act)
    ws = workspace(nb)
    hip.tf_linear_mx8(y.ptr, x8.ptr, w8.ptr, wscale.ptr, bias.ptr if bias is not None else None, residual.ptr if residual is not None else None,
                      rows, n_out, K, act, 1 if out_mx else 0, ws.ptr if ws else None, nb, _sh())
    return y
