"""Linear -- mirrors tinyfusers/ff/linear.py:112-121 (live branch: cp.dot(x, W^T) + b, fp32 cuBLAS SGEMM via CuPy).
Here: one MFMA implicit-GEMM launch (tf_linear_f16) with bias / residual fused, or the weight-streaming
GEMV (tf_gemv_f16) for <= 8 rows (time-embedding MLP, ResBlock emb_layers)."""
import numpy as np

from ..native import hip
from ..storage.tensor import DeviceArray, _sh, asarray


def workspace(nbytes):
    return DeviceArray.empty((nbytes,), np.uint8, "row") if nbytes else None


def linear_f16(x, w, b=None, residual=None, act=0, out_features=None):
    """y = act(x . w^T + b) + residual on raw DeviceArrays (x (..., K) row-major, w (N, K))."""
    K = x.shape[-1]
    rows = x.size // K
    n_out = out_features if out_features is not None else w.shape[0]
    y = DeviceArray.empty(x.shape[:-1] + (n_out,), np.float16, "row")
    nb = hip.tf_linear_workspace(rows, n_out, K, act)
    ws = workspace(nb)
    hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, b.ptr if b is not None else None, residual.ptr if residual is not None else None,
                      rows, n_out, K, act, ws.ptr if ws else None, nb, _sh())
    return y


def gemv_f16(x, w, b=None, silu_input=False):
    K = x.shape[-1]
    rows = x.size // K
    y = DeviceArray.empty(x.shape[:-1] + (w.shape[0],), np.float16, "row")
    hip.tf_gemv_f16(y.ptr, x.ptr, w.ptr, b.ptr if b is not None else None, rows, w.shape[0], K, 1 if silu_input else 0, _sh())
    return y


def fold_layer_norm(w, b, ln):
    """(w', bias', colsum) of Linear(LayerNorm(.)) for tf_linear_ln_f16; w (N,K), b (N,) or None, ln a LayerNorm."""
    n, k = w.shape
    wf = DeviceArray.empty((n, k), np.float16, "row")
    bf = DeviceArray.empty((n,), np.float16, "row")
    cs = DeviceArray.empty((n,), np.float32, "row")
    hip.tf_ln_fold_weights_f16(wf.ptr, bf.ptr, cs.ptr, w.ptr, b.ptr if b is not None else None, ln.weight.ptr, ln.bias.ptr, n, k, _sh())
    return wf, bf, cs


def linear_ln_f16(x, folded, eps, residual=None, act=0, out_features=None):
    """y = act(Linear(LayerNorm(x))) + residual in one launch; ``folded`` from fold_layer_norm."""
    wf, bf, cs = folded
    K = x.shape[-1]
    rows = x.size // K
    n_out = out_features if out_features is not None else wf.shape[0]
    y = DeviceArray.empty(x.shape[:-1] + (n_out,), np.float16, "row")
    hip.tf_linear_ln_f16(y.ptr, x.ptr, wf.ptr, bf.ptr, cs.ptr, residual.ptr if residual is not None else None, rows, n_out, K, act,
                         float(np.asarray(eps).reshape(-1)[0]), _sh())
    return y


class Linear:
    def __init__(self, in_features, out_features, bias=True, init=True):
        self.in_features, self.out_features = in_features, out_features
        # reference init: all-ones weight and bias (ff/linear.py:114-115)
        self.weight = asarray(np.ones((out_features, in_features), dtype=np.float16)) if init else None
        self.bias = (asarray(np.ones((out_features,), dtype=np.float16)) if init else None) if bias else None
        self._has_bias = bias

    def __call__(self, x, residual=None, silu_input=False):
        assert x.layout == "row" and x.shape[-1] == self.weight.shape[1], (x.shape, self.weight.shape)
        rows = x.size // x.shape[-1]
        if rows <= 8 and residual is None:
            return gemv_f16(x, self.weight, self.bias, silu_input)
        assert not silu_input
        return linear_f16(x, self.weight, self.bias, residual)
