"""scaled_dot_product_attention -- mirrors tinyfusers/attention/sdpa.py:53-77.
One fused flash-style launch (tf_sdpa_f16); the (B,NH,Tq,Tk) score matrix never exists."""
import numpy as np

from ..native import hip
from ..storage.tensor import DeviceArray, _sh


def sdpa_strided(o, q, k, v, B, NH, Tq, Tk, HS, qs, ks, vs, os_, causal=False):
    """Raw launch: q/k/v/o are DeviceArrays (or views), *s = (batch, head, token) element strides."""
    hip.tf_sdpa_f16(o.ptr, q.ptr, k.ptr, v.ptr, B, NH, Tq, Tk, HS, *qs, *ks, *vs, *os_, 1 if causal else 0, _sh())
    return o


def _is_causal_mask(m, tq, tk):
    m = np.asarray(m)
    if m.shape[-2:] != (tq, tk):
        return False
    m2 = m.reshape(-1, tq, tk)[0]
    tri = np.tril(np.ones((tq, tk), dtype=bool))
    if m.dtype == np.bool_:
        return bool((m2 == tri).all())
    return bool(np.isneginf(m2[~tri]).all() and (m2[tri] == 0).all())


def scaled_dot_product_attention(q_cp, k_cp, v_cp, attn_mask=None):
    """q,k,v: (B, NH, T, HS) contiguous DeviceArrays -> (B, NH, Tq, HS).  attn_mask: None or a causal mask
    (the only mask the reference ever passes, attention/attention.py:94)."""
    B, NH, Tq, HS = q_cp.shape
    Tk = k_cp.shape[-2]
    causal = False
    if attn_mask is not None:
        if not _is_causal_mask(attn_mask, Tq, Tk):
            raise NotImplementedError("tf_sdpa_f16 supports attn_mask=None or a causal mask")
        causal = True
    o = DeviceArray.empty((B, NH, Tq, HS), np.float16, "row")
    st = lambda T: (NH * T * HS, T * HS, HS)
    return sdpa_strided(o, q_cp, k_cp, v_cp, B, NH, Tq, Tk, HS, st(Tq), st(Tk), st(Tk), st(Tq), causal)
