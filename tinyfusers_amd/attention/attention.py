"""CrossAttention / BasicTransformerBlock / SpatialTransformer -- mirrors tinyfusers/attention/attention.py:26-76.

MI355X-first differences (results identical to the reference within fp16 tolerance):
  * to_q/to_k/to_v of self-attention run as ONE GEMM over a device-side concatenated (3C, C) weight, to_k/to_v
    of cross-attention as one (2C, ctx) GEMM; the SDPA kernel reads heads through strides, so the
    reshape+transpose(0,2,1,3) of attention.py:38 never materialises;
  * the head merge after SDPA is an output-stride choice (config.head_merge);
  * every residual add is fused into the producing GEMM's epilogue.
"""
import numpy as np

from .. import config
from ..ff.group_norm import GroupNorm
from ..ff.layer_norm import LayerNorm
from ..ff.linear import Linear, fold_layer_norm, linear_any, linear_f16, linear_ln_f16
from ..ff.nn import FeedForward
from ..native import hip
from ..storage.tensor import DeviceArray, _sh, is_bfloat16
from ..vision.conv2d import Conv2d, _conv
from .sdpa import sdpa_strided


def _concat_rows(ws):
    k = ws[0].shape[1]
    n = sum(w.shape[0] for w in ws)
    out = DeviceArray.empty((n, k), ws[0].dtype, "row")
    off = 0
    for w in ws:
        hip.tf_memcpy_async(out.ptr + off, w.ptr, w.nbytes, 3, _sh())
        off += w.nbytes
    return out


class AttnBlock:
    """attention/attention.py:10-24 (VAE mid block).  The reference hands the NCHW q/k/v straight to
    scaled_dot_product_attention, which reads them as (B, NH, T, HS) = (b, c, h, w): one 'head' per channel attending
    over the h x w matrix of that channel (not the LDM (b, hw, c) single-head attention).  config.head_merge ==
    'reference_exact' reproduces exactly that (NHWC -> NCHW re-layout around the fused SDPA kernel); 'intended'
    runs the LDM form."""

    def __init__(self, in_channels, init=True):
        self.norm = GroupNorm(32, in_channels, init=init)
        self.q = Conv2d(in_channels, in_channels, kernel_size=[1, 1], init=init)
        self.k = Conv2d(in_channels, in_channels, kernel_size=[1, 1], init=init)
        self.v = Conv2d(in_channels, in_channels, kernel_size=[1, 1], init=init)
        self.proj_out = Conv2d(in_channels, in_channels, kernel_size=[1, 1], init=init)

    def __call__(self, x):
        b, c, h, w = x.shape
        h_ = self.norm(x)
        q, k, v = self.q(h_), self.k(h_), self.v(h_)
        if config.head_merge == "reference_exact":
            def nchw(t):
                o = DeviceArray.empty((b, c, h, w), np.float16, "row")
                hip.tf_nhwc_to_nchw_f16(o.ptr, t.ptr, b, c, h, w, _sh())
                return o
            qn, kn, vn = nchw(q), nchw(k), nchw(v)
            on = DeviceArray.empty((b, c, h, w), np.float16, "row")
            st = (c * h * w, h * w, w)
            sdpa_strided(on, qn, kn, vn, b, c, h, h, w, st, st, st, st)
            o = DeviceArray.empty((b, c, h, w), np.float16, "nhwc")
            hip.tf_nchw_to_nhwc_f16(o.ptr, on.ptr, b, c, h, w, _sh())
        else:
            # LDM form: ONE head of size c over the h*w tokens; NHWC q/k/v are already (b, hw, c) token matrices.  Head size 512 is
            # beyond the flash kernel's registers: the unfused matmul / softmax / matmul path (once per image, not per step).
            from .sdpa import sdpa_unfused
            t = h * w
            o = sdpa_unfused(q.view((b, 1, t, c), "row"), k.view((b, 1, t, c), "row"), v.view((b, 1, t, c), "row")).view((b, c, h, w), "nhwc")
        return self.proj_out(o, residual=x)


class CrossAttention:
    def __init__(self, query_dim, context_dim, n_heads, d_head, init=True):
        self.to_q = Linear(query_dim, n_heads * d_head, bias=False, init=init)
        self.to_k = Linear(context_dim, n_heads * d_head, bias=False, init=init)
        self.to_v = Linear(context_dim, n_heads * d_head, bias=False, init=init)
        self.num_heads = n_heads
        self.head_size = d_head
        self.to_out = [Linear(n_heads * d_head, query_dim, init=init)]
        self._fused = None

    def _fused_weights(self, self_attn):
        key = (self.to_q.weight.wkey, self.to_k.weight.wkey, self.to_v.weight.wkey, self_attn)
        if self._fused is None or self._fused[0] != key:
            ws = [self.to_q.weight, self.to_k.weight, self.to_v.weight] if self_attn else [self.to_k.weight, self.to_v.weight]
            self._fused = (key, _concat_rows(ws))
        return self._fused[1]

    def _folded(self, ln, self_attn):
        """LayerNorm folded into the fused q|k|v weight (self-attention) or into to_q (cross-attention)."""
        w = self._fused_weights(True) if self_attn else self.to_q.weight
        key = (w.wkey, ln.weight.wkey, ln.bias.wkey)
        if getattr(self, "_ln_fold", None) is None or self._ln_fold[0] != key:
            self._ln_fold = (key, fold_layer_norm(w, None, ln))
        return self._ln_fold[1]

    def project_kv(self, context):
        """(b, tk, 2C) fused K|V projection of the context (computed once per UNet call by the model)."""
        return linear_any(context, self._fused_weights(False))

    def __call__(self, x, context=None, residual=None, kv=None, ln=None):
        """ln: a LayerNorm to apply to x first, folded into the q (or q|k|v) projection (x is then the RAW input)."""
        b, t, _ = x.shape
        nh, hs = self.num_heads, self.head_size
        c = nh * hs
        from ..ff import fp8
        # config 5: the projections on block-scaled e4m3 operands where the kernel takes the shape (ff/fp8.py: K >= 640 levels) -- LayerNorm written
        # as an mx8 tensor instead of folded, the attention output quantised in one pass in front of to_out
        mx_in = fp8.ATTENTION and ln is not None and fp8.linear_ok(b * t, 3 * c if (context is None and kv is None) else c, x.shape[-1])
        if mx_in:
            if getattr(self, "_cache8", None) is None:
                self._cache8 = {"qkv": {}, "q": {}, "out": {}}
            x8 = fp8.layer_norm_mx(x, ln)
        if context is None and kv is None:
            if mx_in:
                w8, wsc = fp8.pack_weight(self._fused_weights(True), self._cache8["qkv"])
                qkv = fp8.linear_mx(x8, w8, wsc, None)
            elif ln is not None:
                qkv = linear_ln_f16(x, self._folded(ln, True), ln.eps)
            else:
                qkv = linear_any(x, self._fused_weights(True))        # (b, t, 3C): q | k | v
            q, k, v = qkv, qkv.view((b, t, 3 * c), "row", c), qkv.view((b, t, 3 * c), "row", 2 * c)
            tk, qs, ks = t, (t * 3 * c, hs, 3 * c), (t * 3 * c, hs, 3 * c)
        else:
            if mx_in:
                w8, wsc = fp8.pack_weight(self.to_q.weight, self._cache8["q"])
                q = fp8.linear_mx(x8, w8, wsc, None)
            else:
                q = linear_ln_f16(x, self._folded(ln, False), ln.eps) if ln is not None else linear_any(x, self.to_q.weight)
            if kv is None:
                kv = self.project_kv(context)
            if hasattr(kv, "ld"):                          # column slice of the UNet's step-level K|V GEMM
                tk, ld = kv.arr.shape[1], kv.ld
                k = kv.arr.view(kv.arr.shape, "row", kv.off)
                v = kv.arr.view(kv.arr.shape, "row", kv.off + c)
            else:
                tk, ld = kv.shape[1], 2 * c
                k, v = kv, kv.view(kv.shape, "row", c)
            qs, ks = (t * c, hs, c), (tk * ld, hs, ld)
        o = DeviceArray.empty((b, t, c), x.dtype, "row")       # (attention in the step's own 16-bit type: bfloat16 q / k / v run the bf16 kernels, no fp16 hop)
        if config.head_merge == "reference_exact":
            os_ = (nh * t * hs, t * hs, hs)      # (b,h,t,d) contiguous, then read as (b, t, h*d): attention.py:38-39
        else:
            os_ = (t * c, hs, c)                 # LDM-intended merge
        sdpa_strided(o, q, k, v, b, nh, t, tk, hs, qs, ks, ks, os_)
        lo = self.to_out[0]
        if fp8.ATTENTION and fp8.linear_ok(b * t, lo.weight.shape[0], c):
            if getattr(self, "_cache8", None) is None:
                self._cache8 = {"qkv": {}, "q": {}, "out": {}}
            w8, wsc = fp8.pack_weight(lo.weight, self._cache8["out"])
            return fp8.linear_mx(fp8.quantize_mx(o), w8, wsc, lo.bias, residual=residual)
        return lo(o, residual=residual)


class BasicTransformerBlock:
    def __init__(self, dim, context_dim, n_heads, d_head, init=True):
        self.attn1 = CrossAttention(dim, dim, n_heads, d_head, init=init)
        self.ff = FeedForward(dim, init=init)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head, init=init)
        self.norm1 = LayerNorm(dim, init=init)
        self.norm2 = LayerNorm(dim, init=init)
        self.norm3 = LayerNorm(dim, init=init)

    def __call__(self, x, context=None, kv=None, defer_ff2=False):
        """defer_ff2: stop after GEGLU and return (hidden (b, t, 4*dim), x): the caller folds ff.net[2] into what follows."""
        if config.fuse_layer_norm and x.shape[-1] % 64 == 0:
            x = self.attn1(x, residual=x, ln=self.norm1)
            x = self.attn2(x, context=context, residual=x, kv=kv, ln=self.norm2)
            if defer_ff2:
                return self.ff.net[0](x, ln=self.norm3), x
            x = self.ff(x, residual=x, ln=self.norm3)
            return x
        x = self.attn1(self.norm1(x), residual=x)
        x = self.attn2(self.norm2(x), context=context, residual=x, kv=kv)
        if defer_ff2:
            return self.ff.net[0](self.norm3(x)), x
        x = self.ff(self.norm3(x), residual=x)
        return x


class SpatialTransformer:
    def __init__(self, channels, context_dim, n_heads, d_head, init=True):
        self.norm = GroupNorm(32, channels, init=init)
        assert channels == n_heads * d_head
        self.proj_in = Conv2d(channels, n_heads * d_head, kernel_size=[1, 1], init=init)
        self.transformer_blocks = [BasicTransformerBlock(channels, context_dim, n_heads, d_head, init=init)]
        self.proj_out = Conv2d(n_heads * d_head, channels, kernel_size=[1, 1], init=init)
        self._fold = None

    def _ff2_proj_out(self):
        """ff.net[2] (Linear 4C -> C, + residual x2) followed by proj_out (1x1 conv C -> C) has nothing non-linear in between:
        proj_out(h W2^T + b2 + x2) = [h | x2] [Wp W2 | Wp]^T + (Wp b2 + bp).  One GEMM with K = 5C over the pair (h, x2) --
        the same FLOPs as the two it replaces, one launch and one activation round trip less.  Folded once per weight set
        on the host in fp32 (first eager call; cached by weight pointers)."""
        ff2, po = self.transformer_blocks[-1].ff.net[2], self.proj_out
        key = (ff2.weight.wkey, ff2.bias.wkey, po.weight.wkey, po.bias.wkey)
        if self._fold is None or self._fold[0] != key:
            c = po.weight.shape[0]
            w2, b2 = ff2.weight.numpy(), ff2.bias.numpy()
            wp, bp = po.weight.numpy().reshape(c, -1), po.bias.numpy()
            wf = np.concatenate((wp @ w2, wp), axis=1).astype(np.float32)
            bf = (wp @ b2 + bp).astype(np.float32)
            self._fold = (key, DeviceArray.from_numpy(wf.reshape(c, -1, 1, 1), po.weight.dtype), DeviceArray.from_numpy(bf, po.bias.dtype, layout="row"))
        return self._fold[1], self._fold[2]

    def __call__(self, x, context=None, kv=None, out_gn=0, out_norm=None):
        b, c, h, w = x.shape
        x_in = x
        x = self.proj_in(x, gn_in=(self.norm, False))    # GroupNorm -> 1x1 conv (attention.py:66-68) as one launch where the statistics came with x
        x = x.tokens()                                   # (b, hw, c): free re-view of NHWC (attention.py:71)
        if config.fold_proj_out and config.dtype != "fp8" and c % 8 == 0 and self.proj_out.weight.shape[0] == c:   # (fp8: the FeedForward runs in e4m3, proj_out stays fp16)
            for block in self.transformer_blocks[:-1]:
                x = block(x, context=context, kv=kv)
            hid, x2 = self.transformer_blocks[-1](x, context=context, kv=kv, defer_ff2=True)
            wf, bf = self._ff2_proj_out()
            pair = (hid.image(b, hid.shape[-1], h, w), x2.image(b, c, h, w))
            return _conv(pair, wf, bf, [0, 0], [1, 1], [1, 1], residual=x_in, gn=out_gn, out_norm=out_norm)
        for block in self.transformer_blocks:
            x = block(x, context=context, kv=kv)
        x = x.image(b, c, h, w)                          # attention.py:74
        return self.proj_out(x, residual=x_in, gn=out_gn, out_norm=out_norm)   # + x_in fused (attention.py:75)


class CLIPAttention:
    """attention/attention.py:78-104 -- 12-head causal self-attention of the CLIP text encoder.  q/k/v run as one
    (3*768, 768) GEMM with the LayerNorm in front folded in; heads are read through strides; unlike CrossAttention
    the reference merges the heads back WITH the transpose (:100), i.e. the standard (b, t, h*d) layout."""

    def __init__(self, init=True):
        self.embed_dim = 768
        self.num_heads = 12
        self.head_dim = self.embed_dim // self.num_heads
        self.k_proj = Linear(self.embed_dim, self.embed_dim, init=init)
        self.v_proj = Linear(self.embed_dim, self.embed_dim, init=init)
        self.q_proj = Linear(self.embed_dim, self.embed_dim, init=init)
        self.out_proj = Linear(self.embed_dim, self.embed_dim, init=init)
        self._fused = None

    def _qkv(self, ln):
        ps = [self.q_proj, self.k_proj, self.v_proj]
        key = tuple(p.weight.wkey for p in ps) + tuple(p.bias.wkey for p in ps) + ((ln.weight.wkey, ln.bias.wkey) if ln is not None else ())
        if self._fused is None or self._fused[0] != key:
            w = _concat_rows([p.weight for p in ps])
            b = _concat_rows([p.bias.view((p.bias.shape[0], 1), "row") for p in ps]).view((3 * self.embed_dim,), "row")
            self._fused = (key, fold_layer_norm(w, b, ln) if ln is not None else (w, b), (w, b))
        return self._fused[1]

    def __call__(self, hidden_states, causal_attention_mask=None, residual=None, ln=None):
        """causal_attention_mask: None, the causal mask of vae/encoder.py:79 (host array) -- both on the fused flash kernel -- or any
        other boolean / additive mask broadcastable to (b, heads, t, t) (attention/attention.py:94 hands it to sdpa.py:67-68 unchanged),
        which runs the unfused matmul / softmax / matmul path on heads gathered to the contiguous (b, heads, t, d) layout."""
        from .sdpa import _additive_mask, _is_causal_mask, sdpa_unfused
        b, t, c = hidden_states.shape
        nh, hs = self.num_heads, self.head_dim
        causal = causal_attention_mask is not None
        general = causal and not _is_causal_mask(causal_attention_mask, t, t)
        if ln is not None:
            qkv = linear_ln_f16(hidden_states, self._qkv(ln), ln.eps)
        else:
            w, bias = self._qkv(None)
            qkv = linear_f16(hidden_states, w, bias)
        o = DeviceArray.empty((b, t, c), np.float16, "row")
        if general:
            def heads_of(off):                                     # columns off .. off + c of qkv -> (b, heads, t, d), contiguous
                out = DeviceArray.empty((b, nh, t, hs), np.float16, "row")
                for i in range(b * nh):
                    hip.tf_memcpy_2d_async(out.ptr + i * t * hs * 2, hs * 2, qkv.ptr + ((i // nh) * t * 3 * c + off + (i % nh) * hs) * 2, 3 * c * 2,
                                           hs * 2, t, _sh())
                return out
            oh = sdpa_unfused(heads_of(0), heads_of(c), heads_of(2 * c), _additive_mask(causal_attention_mask, b, nh, t, t))
            for i in range(b * nh):                                # heads back WITH the transpose (attention.py:96): (b, t, heads * d)
                hip.tf_memcpy_2d_async(o.ptr + ((i // nh) * t * c + (i % nh) * hs) * 2, c * 2, oh.ptr + i * t * hs * 2, hs * 2, hs * 2, t, _sh())
            return self.out_proj(o, residual=residual)
        q, k, v = qkv, qkv.view((b, t, 3 * c), "row", c), qkv.view((b, t, 3 * c), "row", 2 * c)
        st = (t * 3 * c, hs, 3 * c)
        sdpa_strided(o, q, k, v, b, nh, t, t, hs, st, st, st, (t * c, hs, c), causal)
        return self.out_proj(o, residual=residual)
