"""CPU emulation of the fp8 (OCP e4m3) conv / linear path of BASELINE config 5 -- TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference has no fp8 path (it runs fp32 throughout, example/sd1.py:33); config 5 asks for one, gated at UNet rel-L2 <= 0.1 against
the fp32 oracle.  This module restates the quantisation the HIP path applies so that (a) single ops can be
checked tightly -- same e4m3 operands on both sides, fp32 accumulate -- and (b) the layer policy can be evaluated on the CPU:
e4m3 weights with one scale per output channel (max|w| / 448), e4m3 activations with scale 1 (saturating), for the two 3x3 convolutions
of every ResBlock (Cin, Cout >= 64: their input is a GroupNorm + SiLU output) and the FeedForward linears (LayerNorm / GEGLU outputs);
everything else as in oracle.unet -- in particular the up / down-sampling convs, whose input is the raw residual stream and has no
business being cut to e4m3 at a fixed scale of 1 (saturation at 448, nothing below 2e-3)."""
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
    e = torch.where(a > 0, e, torch.full_like(e, -127))
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


class policy:
    """``with oracle.fp8.policy():`` makes oracle.unet_forward evaluate the fp8 layer policy (the ResBlocks' 3x3 convs with
    Cin, Cout >= 64 and the FeedForward linears of width >= 640 on e4m3 operands).  Restores the fp32 functions on exit."""

    def __enter__(self):
        from . import unet as U
        self._conv, self._ff, self._res = ops.conv2d_bias, U.feed_forward, U.resblock
        conv0, lin0, geglu0, res0 = ops.conv2d_bias, ops.linear, ops.geglu, U.resblock
        inside = [0]                                       # > 0 while a ResBlock body runs

        def conv(x, w, b, padding=(0, 0), stride=(1, 1), dilation=(1, 1)):
            w_ = ops.as_t(w)
            if inside[0] and w_.shape[-1] == 3 and w_.shape[0] >= 64 and w_.shape[1] >= 64 and w_.shape[1] % 64 == 0:
                return conv0(quant_act(x), quant_weight(w_)[0], b, padding, stride, dilation)
            return conv0(x, w, b, padding, stride, dilation)

        def res(*a, **k):
            inside[0] += 1
            try:
                return res0(*a, **k)
            finally:
                inside[0] -= 1
        U.resblock = res

        ff0 = U.feed_forward

        def ff(x, W, p):
            if ops.as_t(x).shape[-1] < 640:                # the K = 320 FeedForward stays fp16 (tinyfusers_amd/ff/nn.py: faster there, and exact)
                return ff0(x, W, p)
            h = geglu0(quant_act(x), quant_weight(W[p + ".net.0.proj.weight"])[0], W[p + ".net.0.proj.bias"])
            return lin0(quant_act(h), quant_weight(W[p + ".net.2.weight"])[0], W[p + ".net.2.bias"])
        ops.conv2d_bias, U.feed_forward = conv, ff
        return self

    def __exit__(self, *a):
        from . import unet as U
        ops.conv2d_bias, U.feed_forward, U.resblock = self._conv, self._ff, self._res
