"""GEGLU / FeedForward -- mirrors tinyfusers/ff/nn.py:5-23.  The projection, the split and a * gelu(gate) are
one GEMM launch: the (2*dim_out, dim_in) weight is re-packed once on the device into 16-row blocks
alternating value / gate so that both halves of a pair land in the same MFMA lane (tf_linear_f16 act=1)."""
import numpy as np

from ..native import hip
from ..storage.tensor import DeviceArray, Tensor, _sh
from .linear import Linear, fold_layer_norm, linear_any, linear_f16, linear_ln_f16


def pack_geglu(weight, bias):
    """(2N, K) [values ; gates] -> 16-row blocks v0 g0 v1 g1 ... (and the bias likewise), on the device."""
    two_n, k = weight.shape
    n = two_n // 2
    assert n % 16 == 0, "GEGLU width must be a multiple of 16"
    wp = DeviceArray.empty((two_n, k), weight.dtype, "row")     # (2-byte copies: fp16 and bfloat16 alike)
    blk = 16 * k * 2
    hip.tf_memcpy_2d_async(wp.ptr, 2 * blk, weight.ptr, blk, blk, n // 16, _sh())
    hip.tf_memcpy_2d_async(wp.ptr + blk, 2 * blk, weight.ptr + n * k * 2, blk, blk, n // 16, _sh())
    bp = DeviceArray.empty((two_n,), bias.dtype, "row")
    hip.tf_memcpy_2d_async(bp.ptr, 64, bias.ptr, 32, 32, n // 16, _sh())
    hip.tf_memcpy_2d_async(bp.ptr + 32, 64, bias.ptr + n * 2, 32, 32, n // 16, _sh())
    return wp, bp


class GEGLU:
    def __init__(self, dim_in, dim_out, init=True):
        self.proj = Linear(dim_in, dim_out * 2, init=init)
        self.dim_out = dim_out
        self._packed = None

    def _pack(self):
        key = (self.proj.weight.wkey, self.proj.bias.wkey)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key,) + pack_geglu(self.proj.weight, self.proj.bias)
        return self._packed[1], self._packed[2]

    def _pack_ln(self, ln):
        key = (self.proj.weight.wkey, self.proj.bias.wkey, ln.weight.wkey, ln.bias.wkey)
        if getattr(self, "_packed_ln", None) is None or self._packed_ln[0] != key:
            wf, bf, cs = fold_layer_norm(self.proj.weight, self.proj.bias, ln)       # fold first ...
            wp, bp = pack_geglu(wf, bf)                                            # ... then interleave value | gate blocks
            n = self.dim_out
            csp = DeviceArray.empty((2 * n,), np.float32, "row")
            hip.tf_memcpy_2d_async(csp.ptr, 128, cs.ptr, 64, 64, n // 16, _sh())
            hip.tf_memcpy_2d_async(csp.ptr + 64, 128, cs.ptr + n * 4, 64, 64, n // 16, _sh())
            self._packed_ln = (key, (wp, bp, csp))
        return self._packed_ln[1]

    def __call__(self, x, ln=None):
        if ln is not None:
            assert self.dim_out % 16 == 0 and self.proj.bias is not None and x.shape[-1] % 64 == 0
            return linear_ln_f16(x, self._pack_ln(ln), ln.eps, act=1, out_features=self.dim_out)
        if self.dim_out % 16 == 0 and self.proj.bias is not None:
            wp, bp = self._pack()
            return linear_any(x, wp, bp, None, act=1, out_features=self.dim_out)
        h = self.proj(x)                                   # unfused fallback shape (still HIP): split + a*gelu(gate)
        assert h.dtype == np.float16, "unfused GEGLU fallback: fp16 only"
        y = DeviceArray.empty(x.shape[:-1] + (self.dim_out,), np.float16, "row")
        hip.tf_geglu_f16(y.ptr, h.ptr, h.size // h.shape[-1], self.dim_out, _sh())
        return y


class FeedForward:
    def __init__(self, dim, mult=4, init=True):
        self.net = [
            GEGLU(dim, dim * mult, init=init),
            lambda x: x,  # dropout slot: keeps list indices aligned with checkpoint keys (ff/nn.py:18)
            Linear(dim * mult, dim, init=init),
        ]

    def __call__(self, x, residual=None, ln=None):
        from . import fp8
        g, l2 = self.net[0], self.net[2]
        rows = x.size // x.shape[-1]
        if ln is not None and g.dim_out % 64 == 0 and g.proj.bias is not None and fp8.linear_ok(rows, g.dim_out, x.shape[-1], 1, True) \
                and fp8.linear_ok(rows, l2.weight.shape[0], g.dim_out):
            # config 5: LayerNorm -> block-scaled e4m3, GEGLU projection with a block-scaled e4m3 output, second Linear (+ bias + residual, fp16 out);
            # K >= fp8.MIN_K only (at K = 320 the fp16 pair -- LayerNorm folded into the persistent short-K kernel -- is faster, and exact)
            h8 = fp8.layer_norm_mx(x, ln)
            if getattr(g, "_cache8", None) is None:
                g._cache8, l2._cache8 = {}, {}
            wp, bp = g._pack()                              # value | gate rows interleaved in 16-row blocks, then quantised row by row
            w8, wsc = fp8.pack_weight(wp, g._cache8)
            hid8 = fp8.linear_mx(h8, w8, wsc, bp, act=1, out_features=g.dim_out, out_mx=True)
            w28, wsc2 = fp8.pack_weight(l2.weight, l2._cache8)
            return fp8.linear_mx(hid8, w28, wsc2, l2.bias, residual=residual)
        h = self.net[0](x, ln=ln) if ln is not None else self.net[0](x)
        return self.net[2](h, residual=residual)


class CLIPMLP:
    """ff/nn.py:25-34 -- fc1 -> quick_gelu -> fc2.  ``ln``: the LayerNorm in front of the block, folded into fc1;
    ``residual`` is added in fc2's epilogue (vae/encoder.py:62-65)."""

    def __init__(self, init=True):
        self.fc1 = Linear(768, 3072, init=init)
        self.fc2 = Linear(3072, 768, init=init)

    def _folded(self, ln):
        key = (self.fc1.weight.wkey, self.fc1.bias.wkey, ln.weight.wkey, ln.bias.wkey)
        if getattr(self, "_ln_fold", None) is None or self._ln_fold[0] != key:
            self._ln_fold = (key, fold_layer_norm(self.fc1.weight, self.fc1.bias, ln))
        return self._ln_fold[1]

    def __call__(self, hidden_states, residual=None, ln=None):
        h = linear_ln_f16(hidden_states, self._folded(ln), ln.eps) if ln is not None else self.fc1(hidden_states)
        return self.fc2(Tensor.quick_gelu(h), residual=residual)
