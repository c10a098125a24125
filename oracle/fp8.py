"""CPU emulation of the fp8 (OCP e4m3) conv / linear path of BASELINE config 5 -- TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference has no fp8 path (it runs fp32 throughout, example/sd1.py:33); config 5 asks for one, gated at UNet rel-L2 <= 0.1 against
the fp32 oracle.  This module restates the quantisation the HIP path applies so that (a) single ops can be
checked tightly -- same e4m3 operands on both sides, fp32 accumulate -- and (b) the layer policy can be evaluated on the CPU:
e4m3 weights with one scale per output channel (max|w| / 448); activations block-scaled e4m3 (quant_act_mx: one power-of-two E8M0 scale
per 32 consecutive channels of a pixel / token -- round 4; quant_act is round 2's fixed scale of 1, kept for the op-level entries that still
take it); the layer policy of tinyfusers_amd/ff/fp8.py: the two 3x3 convolutions of every ResBlock, the FeedForward pair and the attention
projections q|k|v / q / to_out, each only where the block-scaled kernel takes the shape (mx_gemm_supported restates the host rule) and
K >= 640; everything else as in oracle.unet."""
import numpy as np
import torch

from . import ops

F8 = torch.float8_e4m3fn


def quant_act(t):
    """activation operand as the HIP kernels store it: e4m3, scale 1, saturating at +-448, round to nearest even."""
    return ops.as_t(t).clamp(-448.0, 448.0).to(F8).to(torch.float32)


MX_BLOCK = 32          # elements that share one E8M0 scale (the block of v_mfma_scale_f32_16x16x128_f8f6f4: 32 consecutive K values of a lane)


def mx_scale_exponent(amax):
    """Power-of-two block scale 2^e with amax / 2^e <= 448 (the largest e4m3 value): e = ceil(log2(amax / 448)), exact on the fp32 bits
    (frexp), clamped to the E8M0 range [-127, 127]; a block of zeros gets e = -127.  The E8M0 byte is e + 127."""
    a = ops.as_t(amax)
    m, ex = torch.frexp(a / 448.0)                          # a / 448 = m * 2^ex, m in [0.5, 1)
    e = torch.where(m == 0.5, ex - 1, ex)                   # exact powers of two need no round-up
    e = torch.where(a / 448.0 >= 2.0 ** -126, e, torch.full_like(e, -127))     # a zero or subnormal quotient: 2^-127 (the device reads the exponent field)
    return e.clamp(-127, 127)


def quant_act_mx(t, channel_dim=-1):
    """activation operand of the block-scaled e4m3 GEMMs as the HIP producers store it: along the channel axis, every MX_BLOCK consecutive
    channels of one pixel / token share one power-of-two scale 2^e (mx_scale_exponent of the block's largest magnitude); elements are
    e4m3(x / 2^e), round to nearest even -- no saturation can occur.  Returns the dequantised tensor (fp32)."""
    t = ops.as_t(t)
    x = t.movedim(channel_dim, -1)
    sh = x.shape
    assert sh[-1] % MX_BLOCK == 0, sh
    xb = x.reshape(*sh[:-1], sh[-1] // MX_BLOCK, MX_BLOCK)
    e = mx_scale_exponent(xb.abs().amax(-1, keepdim=True))
    s = torch.ldexp(torch.ones_like(xb[..., :1]), e)
    q = (xb / s).to(F8).to(torch.float32) * s
    return q.reshape(sh).movedim(-1, channel_dim)


def quant_act_mx_codes(t, channel_dim=-1):
    """(e4m3 codes as uint8, E8M0 scale bytes as uint8) of quant_act_mx, channel axis last: what the device tensors hold."""
    t = ops.as_t(t)
    x = t.movedim(channel_dim, -1)
    sh = x.shape
    xb = x.reshape(*sh[:-1], sh[-1] // MX_BLOCK, MX_BLOCK)
    e = mx_scale_exponent(xb.abs().amax(-1, keepdim=True))
    s = torch.ldexp(torch.ones_like(xb[..., :1]), e)
    codes = (xb / s).to(F8).view(torch.uint8).reshape(sh)
    return codes.numpy(), (e.squeeze(-1) + 127).to(torch.uint8).numpy()


def quant_weight(w):
    """(dequantised weight, scale per output channel): scale = max|w[n]| / 448, w8 = e4m3(w / scale)."""
    w = ops.as_t(w)
    s = w.reshape(w.shape[0], -1).abs().amax(1)
    s = torch.where(s > 0, s / 448.0, torch.ones_like(s))
    sh = (-1,) + (1,) * (w.dim() - 1)
    return (w / s.reshape(sh)).to(F8).to(torch.float32) * s.reshape(sh), s


def decode_e4m3(raw_u8):
    """uint8 array of e4m3 bytes (as downloaded from the device) -> float32 array."""
    return torch.from_numpy(np.ascontiguousarray(raw_u8, dtype=np.uint8)).view(F8).to(torch.float32).numpy()


MIN_K = 640            # tinyfusers_amd/ff/fp8.py::MIN_K


def mx_gemm_supported(M, N, K, act=0, out_mx=False, howo=None, c_parts=None):
    """Restatement of the host rule (csrc/gemm.hip: mx_shape_ok / pp_ok): the block-scaled GEMM is the 192- / 256-row ping-pong kernel and is
    offered where one of its tiles gives the launch at least 128 blocks.  N = output width (GEGLU: the width AFTER the gate), howo = pixels
    per image of a convolution that carries a time-embedding bias (its tile must not span more than two images), c_parts = channel counts of
    the sources (a count off the 128 grid excludes the 256 x 160 tile)."""
    if M <= 256 or K % 64 or N % 8:
        return False
    n_eff = 2 * N if act == 1 else N
    h2 = any(c % 128 for c in (c_parts or (K,)))
    for bm, bn in ((192, 160), (192, 128), (256, 128), (256, 160)):
        if act == 1 and bn % 64:
            continue
        if out_mx and not (act == 1 and bn == 128):
            continue
        if bm == 256 and bn == 160 and h2:
            continue
        if howo is not None and howo < bm:
            continue
        if -(-M // bm) * -(-n_eff // bn) >= 128:
            return True
    return False


class policy:
    """``with oracle.fp8.policy():`` makes oracle.unet_forward evaluate the fp8 layer policy of tinyfusers_amd/ff/fp8.py on block-scaled e4m3
    activations and per-channel-scaled e4m3 weights: the ResBlocks' 3x3 convs, the FeedForward pair, the attention projections -- each where
    mx_gemm_supported says the kernel takes the shape (``assume_supported=True``: at every shape -- the precision budget of the policy as such,
    for a CPU study at a small batch).  Restores the fp32 functions on exit."""

    def __init__(self, assume_supported=False, attention=False):
        self.assume_supported, self.attention = assume_supported, attention      # attention: tinyfusers_amd/ff/fp8.py::ATTENTION (off by default)

    def __enter__(self):
        mx_ok = (lambda *a, **k: True) if self.assume_supported else mx_gemm_supported
        from . import unet as U
        self._saved = (ops.conv2d_bias, U.feed_forward, U.resblock, U.cross_attention)
        conv0, lin0, geglu0, res0 = ops.conv2d_bias, ops.linear, ops.geglu, U.resblock
        inside = [0]                                       # > 0 while a ResBlock body runs

        def conv(x, w, b, padding=(0, 0), stride=(1, 1), dilation=(1, 1)):
            w_, x_ = ops.as_t(w), ops.as_t(x)
            n, c, h, wd = x_.shape
            if inside[0] and w_.shape[-1] == 3 and tuple(stride) == (1, 1) and w_.shape[0] >= 64 and c >= 64 and c % 64 == 0 \
                    and mx_ok(n * h * wd, w_.shape[0], 9 * c, howo=h * wd, c_parts=(c,)):
                return conv0(quant_act_mx(x_, 1), quant_weight(w_)[0], b, padding, stride, dilation)
            return conv0(x, w, b, padding, stride, dilation)

        def res(*a, **k):
            inside[0] += 1
            try:
                return res0(*a, **k)
            finally:
                inside[0] -= 1

        ff0, ca0 = U.feed_forward, U.cross_attention

        def ff(x, W, p):
            x_ = ops.as_t(x)
            rows, k = x_.numel() // x_.shape[-1], x_.shape[-1]
            hid = ops.as_t(W[p + ".net.2.weight"]).shape[1]
            if k < MIN_K or not (mx_ok(rows, hid, k, 1, True) and mx_ok(rows, k, hid)):
                return ff0(x, W, p)
            h = geglu0(quant_act_mx(x_), quant_weight(W[p + ".net.0.proj.weight"])[0], W[p + ".net.0.proj.bias"])
            return lin0(quant_act_mx(h), quant_weight(W[p + ".net.2.weight"])[0], W[p + ".net.2.bias"])

        def ca(x, context, W, p, n_heads, head_merge="reference_exact"):
            # oracle.unet.cross_attention (attention/attention.py:35-41) with the projections of x and the output projection on e4m3 operands
            x_ = ops.as_t(x)
            b, t, c = x_.shape
            self_attn = context is None
            n_q = 3 * c if self_attn else c                   # the device fuses q | k | v of self-attention into one GEMM
            if not self.attention or c < MIN_K or not mx_ok(b * t, n_q, c):
                return ca0(x, context, W, p, n_heads, head_merge)
            x8 = quant_act_mx(x_)
            q = lin0(x8, quant_weight(W[p + ".to_q.weight"])[0])
            if self_attn:
                k = lin0(x8, quant_weight(W[p + ".to_k.weight"])[0]); v = lin0(x8, quant_weight(W[p + ".to_v.weight"])[0])
            else:
                k = lin0(context, W[p + ".to_k.weight"]); v = lin0(context, W[p + ".to_v.weight"])     # (the context projection stays fp16)
            d = c // n_heads
            q, k, v = [y.reshape(b, -1, n_heads, d).permute(0, 2, 1, 3) for y in (q, k, v)]
            o = ops.scaled_dot_product_attention(q, k, v)
            o = o.reshape(b, -1, n_heads * d) if head_merge == "reference_exact" else o.permute(0, 2, 1, 3).reshape(b, -1, n_heads * d)
            if mx_ok(b * t, c, c):
                return lin0(quant_act_mx(o), quant_weight(W[p + ".to_out.0.weight"])[0], W[p + ".to_out.0.bias"])
            return lin0(o, W[p + ".to_out.0.weight"], W[p + ".to_out.0.bias"])
        ops.conv2d_bias, U.feed_forward, U.resblock, U.cross_attention = conv, ff, res, ca
        return self

    def __exit__(self, *a):
        from . import unet as U
        ops.conv2d_bias, U.feed_forward, U.resblock, U.cross_attention = self._saved
