"""scaled_dot_product_attention -- mirrors tinyfusers/attention/sdpa.py:53-77.
One fused flash-style launch (tf_sdpa_f16); the (B,NH,Tq,Tk) score matrix never exists."""
import numpy as np

from ..native import hip
from ..storage.tensor import DeviceArray, _sh, dtag


def sdpa_strided(o, q, k, v, B, NH, Tq, Tk, HS, qs, ks, vs, os_, causal=False):
    """Raw launch: q/k/v/o are DeviceArrays (or views), *s = (batch, head, token) element strides."""
    assert dtag(q.dtype) == dtag(k.dtype) == dtag(v.dtype) == dtag(o.dtype), "sdpa: q / k / v / o must hold the same 16-bit type"
    hip.tf_sdpa_16(dtag(q.dtype), o.ptr, q.ptr, k.ptr, v.ptr, B, NH, Tq, Tk, HS, *qs, *ks, *vs, *os_, 1 if causal else 0, _sh())
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


def _additive_mask(attn_mask, B, NH, Tq, Tk):
    """attention/sdpa.py:67-68: a boolean mask keeps where True (-inf elsewhere), any other dtype is added.  Returns the fp32
    additive mask as a device array of shape (rows, Tk) with rows = Tq (broadcast over batch and heads) or B*NH*Tq."""
    m = np.asarray(attn_mask.numpy() if isinstance(attn_mask, DeviceArray) else attn_mask)
    if m.dtype == np.bool_:
        m = np.where(m, np.float32(0.0), np.float32(-np.inf))
    m = m.astype(np.float32)
    if m.ndim < 2 or m.shape[-2:] != (Tq, Tk):
        m = np.broadcast_to(m, (Tq, Tk)) if m.ndim <= 2 else np.broadcast_to(m, m.shape[:-2] + (Tq, Tk))
    if m.ndim > 2 and int(np.prod(m.shape[:-2])) > 1:
        m = np.broadcast_to(m, (B, NH, Tq, Tk)).reshape(B * NH * Tq, Tk)
    else:
        m = m.reshape(Tq, Tk)
    return DeviceArray.from_numpy(np.ascontiguousarray(m), np.float32, "row")


def sdpa_unfused(q, k, v, mask=None, scale=None):
    """The reference's own op sequence (attention/sdpa.py:63-76): scale * (q k^T) [+ mask] -> row softmax -> . v, per (batch, head)
    on the MFMA GEMM kernel.  The scores stay in fp32 between the GEMM and the softmax as they do in the reference (its `preatt` is an
    fp32 CuPy array): an fp16 score would be rounded at 1 / scale times the size the exp sees (22.6x at HS = 512).
    q (B,NH,Tq,HS), k / v (B,NH,Tk,HS) contiguous row-major; any HS % 8 == 0, any mask.  The fused flash kernel (tf_sdpa_f16) is the hot
    path; this one covers masks other than causal and head sizes beyond 160.  Everything that does not depend on the head -- padding
    the keys to a multiple of 8, transposing V -- is done once for all heads; the per-head loop is GEMM, softmax, GEMM."""
    B, NH, Tq, HS = q.shape
    Tk = k.shape[-2]
    assert HS % 8 == 0, "sdpa_unfused: head size must be a multiple of 8"
    scale = float(1.0 / np.sqrt(HS)) if scale is None else float(scale)
    Tkp = (Tk + 7) // 8 * 8
    heads = B * NH
    if Tkp != Tk:                                                  # zero value rows beyond Tk: they meet zero probabilities (pad columns)
        vp = DeviceArray.zeros((heads, Tkp, HS), np.float16, "row")
        hip.tf_memcpy_2d_async(vp.ptr, Tkp * HS * 2, v.ptr, Tk * HS * 2, Tk * HS * 2, heads, _sh())
    else:
        vp = v
    vt = DeviceArray.empty((heads, HS, Tkp), np.float16, "row")
    hip.tf_nhwc_to_nchw_f16(vt.ptr, vp.ptr, heads, HS, 1, Tkp, _sh())         # (Tkp, HS) -> (HS, Tkp) for every head
    o = DeviceArray.empty((B, NH, Tq, HS), np.float16, "row")
    # one head's scores / probabilities, reused (stream order) -- tiled over query rows so that the fp32 score buffer stays below 64 MiB
    # whatever the sequence length (a 96 x 96 latent's single-head attention would otherwise hold 9216 x 9216 x 4 B = 340 MB)
    RB = max(8, min(Tq, (64 << 20) // (4 * max(Tk, 1)) // 8 * 8))
    s32 = DeviceArray.empty((RB, Tk), np.float32, "row")
    pr = DeviceArray.empty((RB, Tkp), np.float16, "row")
    mrows = mask.shape[0] if mask is not None else 1
    for bh in range(heads):
        for r0 in range(0, Tq, RB):
            rows = min(RB, Tq - r0)
            hip.tf_linear_f32out_f16(s32.ptr, q.ptr + (bh * Tq + r0) * HS * 2, k.ptr + bh * Tk * HS * 2, rows, Tk, HS, _sh())
            mk = None
            if mask is not None:
                mk = mask.ptr + ((bh * Tq if mrows != Tq else 0) + r0) * Tk * 4
            hip.tf_softmax_mask_rows_f32in_f16(pr.ptr, Tkp, s32.ptr, Tk, mk, rows, Tk, scale, rows, _sh())
            hip.tf_linear_f16(o.ptr + (bh * Tq + r0) * HS * 2, pr.ptr, vt.ptr + bh * HS * Tkp * 2, None, None, rows, HS, Tkp, 0, None, 0, _sh())
    return o


def scaled_dot_product_attention(q_cp, k_cp, v_cp, attn_mask=None):
    """q,k,v: (B, NH, T, HS) contiguous DeviceArrays -> (B, NH, Tq, HS).  attn_mask as in attention/sdpa.py:67-68: None, a boolean
    mask (attend where True) or an additive mask, broadcastable to (B, NH, Tq, Tk).  No mask and the causal mask (the only one the
    reference ever passes, attention/attention.py:94) run on the fused flash kernel; any other mask, or a head size beyond 160, on the
    unfused matmul / softmax / matmul path."""
    B, NH, Tq, HS = q_cp.shape
    Tk = k_cp.shape[-2]
    causal = False
    if attn_mask is not None:
        if not _is_causal_mask(attn_mask.numpy() if isinstance(attn_mask, DeviceArray) else attn_mask, Tq, Tk):
            return sdpa_unfused(q_cp, k_cp, v_cp, _additive_mask(attn_mask, B, NH, Tq, Tk))
        causal = True
    if HS > 160 or HS % 8 != 0:
        m = None
        if causal:
            m = _additive_mask(np.tril(np.ones((Tq, Tk), dtype=bool)), B, NH, Tq, Tk)
        return sdpa_unfused(q_cp, k_cp, v_cp, m)
    o = DeviceArray.empty((B, NH, Tq, HS), np.float16, "row")
    st = lambda T: (NH * T * HS, T * HS, HS)
    return sdpa_strided(o, q_cp, k_cp, v_cp, B, NH, Tq, Tk, HS, st(Tq), st(Tk), st(Tk), st(Tq), causal)
