"""fp8 (OCP e4m3) conv / linear path for BASELINE config 5 ("SD1.5 768x768 batch 32 on 8 GPUs, fp8 MFMA conv/linear path") -- the
precision variant of vision/conv2d.py:9-58 and ff/linear.py:112-121, selected with ``config.set_dtype("fp8")``.

Which layers run in fp8 (measured on the CPU oracle with e4m3 emulation, BASELINE.md section 4 gate: UNet rel-L2 <= 0.1):
every conv and linear in e4m3 gives 0.16; the 3x3 convolutions (Cin, Cout >= 64) + the FeedForward linears give 0.088 and hold
86 % of the conv / linear FLOPs.  The policy is the part of that set whose activation operand is a NORMALISED tensor: the two 3x3
convs of every ResBlock (they read GroupNorm + SiLU outputs; opt-in through ``Conv2d._fp8_ok``) and the FeedForward linears of width >= 640 behind a
LayerNorm.  The 1x1 projections, the attention projections, conv_in / conv_out, the time-embedding GEMVs, the six up / down-sampling
convs (their input is the raw residual stream: e4m3 at a fixed scale of 1 saturates at 448 and flushes everything below 2e-3) and
every module outside the UNet (the VAE's convs) stay fp16.

Weights: e4m3 with one fp32 scale per output channel (tf_pack_weight_fp8, packed once per weight set).  Activations: e4m3 with
scale 1 -- they are normalised tensors (GroupNorm + SiLU, LayerNorm, GEGLU output) -- written directly by the kernel that produces
them.  ``quantize`` (tf_quantize_fp8_f16) remains for callers that hold a normalised fp16 tensor and for the op-level tests.
"""
import ctypes

import numpy as np

from .. import config
from ..native import hip
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
