"""Linear -- mirrors tinyfusers/ff/linear.py:112-121 (live branch: cp.dot(x, W^T) + b, fp32 cuBLAS SGEMM via CuPy).
Here: one MFMA implicit-GEMM launch (tf_linear_f16) with bias / residual fused, or the weight-streaming
GEMV (tf_gemv_f16) for <= 8 rows (time-embedding MLP, ResBlock emb_layers)."""
import numpy as np

from ..native import hip
from ..storage.tensor import DeviceArray, _sh, asarray, dtag, is_bfloat16


def workspace(nbytes):
    return DeviceArray.empty((nbytes,), np.uint8, "row") if nbytes else None


def linear_f16(x, w, b=None, residual=None, act=0, out_features=None):
    """y = act(x . w^T + b) + residual on raw DeviceArrays (x (..., K) row-major, w (N, K)); float16 or -- every tensor alike -- bfloat16
    (tf_linear_16: the same tuned kernels, the element type is a tag)."""
    K = x.shape[-1]
    rows = x.size // K
    n_out = out_features if out_features is not None else w.shape[0]
    assert dtag(w.dtype) == dtag(x.dtype) and (b is None or dtag(b.dtype) == dtag(x.dtype)) and (residual is None or dtag(residual.dtype) == dtag(x.dtype)), "linear: mixed element types"
    y = DeviceArray.empty(x.shape[:-1] + (n_out,), x.dtype, "row")
    nb = hip.tf_linear_workspace(rows, n_out, K, act)
    ws = workspace(nb)
    hip.tf_linear_16(dtag(x.dtype), y.ptr, x.ptr, w.ptr, b.ptr if b is not None else None, residual.ptr if residual is not None else None,
                     rows, n_out, K, act, ws.ptr if ws else None, nb, _sh())
    return y


def linear_bf16(x, w, b=None, residual=None):
    """y = x . w^T + b + residual with every tensor bfloat16 (tests/linear.py:13 lists the type): one bf16-MFMA launch."""
    K = x.shape[-1]
    rows = x.size // K
    assert is_bfloat16(w.dtype) and (b is None or is_bfloat16(b.dtype)) and (residual is None or is_bfloat16(residual.dtype))
    y = DeviceArray.empty(x.shape[:-1] + (w.shape[0],), x.dtype, "row")
    hip.tf_linear_bf16(y.ptr, x.ptr, w.ptr, b.ptr if b is not None else None, residual.ptr if residual is not None else None,
                       rows, w.shape[0], K, _sh())
    return y


def linear_act_bf16(x, w, b=None, residual=None, act=0, out_features=None):
    """bfloat16 form of linear_f16 (act = 1: GEGLU, w / b packed as for the fp16 kernel)."""
    K = x.shape[-1]
    rows = x.size // K
    n_out = out_features if out_features is not None else w.shape[0]
    y = DeviceArray.empty(x.shape[:-1] + (n_out,), x.dtype, "row")
    hip.tf_linear_act_bf16(y.ptr, x.ptr, w.ptr, b.ptr if b is not None else None, residual.ptr if residual is not None else None, rows, n_out, K, act, _sh())
    return y


def linear_any(x, w, b=None, residual=None, act=0, out_features=None):
    """(round 4 name) linear_f16 takes either element type since round 5."""
    return linear_f16(x, w, b, residual, act, out_features)


def to_f16(x):
    """bfloat16 DeviceArray -> fp16 copy (same shape / layout); fp16 arrays pass through."""
    if not is_bfloat16(x.dtype):
        return x
    y = DeviceArray.empty(x.shape, np.float16, x.layout)
    hip.tf_convert_bf16_to_f16(y.ptr, x.ptr, x.size, _sh())
    return y


def to_bf16(x):
    from ..storage.tensor import bfloat16
    if is_bfloat16(x.dtype):
        return x
    y = DeviceArray.empty(x.shape, bfloat16, x.layout)
    hip.tf_convert_f16_to_bf16(y.ptr, x.ptr, x.size, _sh())
    return y


def gemv_f16(x, w, b=None, silu_input=False):
    K = x.shape[-1]
    rows = x.size // K
    y = DeviceArray.empty(x.shape[:-1] + (w.shape[0],), x.dtype, "row")
    hip.tf_gemv_16(dtag(x.dtype), y.ptr, x.ptr, w.ptr, b.ptr if b is not None else None, rows, w.shape[0], K, 1 if silu_input else 0, _sh())
    return y


def fold_layer_norm(w, b, ln):
    """(w', bias', colsum) of Linear(LayerNorm(.)) for tf_linear_ln_f16; w (N,K), b (N,) or None, ln a LayerNorm."""
    n, k = w.shape
    wf = DeviceArray.empty((n, k), w.dtype, "row")
    bf = DeviceArray.empty((n,), w.dtype, "row")
    cs = DeviceArray.empty((n,), np.float32, "row")
    hip.tf_ln_fold_weights_16(dtag(w.dtype), wf.ptr, bf.ptr, cs.ptr, w.ptr, b.ptr if b is not None else None, ln.weight.ptr, ln.bias.ptr, n, k, _sh())
    return wf, bf, cs


def linear_ln_f16(x, folded, eps, residual=None, act=0, out_features=None):
    """y = act(Linear(LayerNorm(x))) + residual in one launch; ``folded`` from fold_layer_norm."""
    wf, bf, cs = folded
    K = x.shape[-1]
    rows = x.size // K
    n_out = out_features if out_features is not None else wf.shape[0]
    y = DeviceArray.empty(x.shape[:-1] + (n_out,), x.dtype, "row")
    hip.tf_linear_ln_16(dtag(x.dtype), y.ptr, x.ptr, wf.ptr, bf.ptr, cs.ptr, residual.ptr if residual is not None else None, rows, n_out, K, act,
                        float(np.asarray(eps).reshape(-1)[0]), _sh())
    return y


def linear(X_gpu, W_gpu, B_gpu):
    """ff/linear.py:66-80, the reference's fused matmul + bias graph with a HALF output (tests/linear.py:15-60 calls it with torch
    tensors): ``X_gpu`` (1, M, K), ``W_gpu`` (1, K, N) -- stored K-major, i.e. y = X @ W, not X @ W^T -- and ``B_gpu`` (1, 1, N), any of
    torch / numpy / DeviceArray in fp16 / bf16 / fp32; returns the (M, N) fp16 DeviceArray the reference allocates as ``Y_actual``
    (its (M, N) output holds one batch, so batch 1 is what the call means).  One MFMA GEMM launch with the bias in the epilogue; the
    (K, N) weight is turned into the (N, K) rows the kernel streams by one device transpose."""
    def f16(t):
        if isinstance(t, DeviceArray):
            assert t.dtype == np.float16, t.dtype
            return t
        if hasattr(t, "detach"):
            t = t.detach().cpu().float().numpy()
        return DeviceArray.from_numpy(np.ascontiguousarray(np.asarray(t, dtype=np.float32)), np.float16, "row")
    xs, ws, bs = tuple(X_gpu.shape), tuple(W_gpu.shape), tuple(B_gpu.shape)
    assert len(xs) == 3 and len(ws) == 3 and xs[0] == 1 and ws[0] == 1 and xs[2] == ws[1] and bs[-1] == ws[2], (xs, ws, bs)
    M, K, N = xs[1], xs[2], ws[2]
    x, w, b = f16(X_gpu), f16(W_gpu), f16(B_gpu)
    wt = DeviceArray.empty((N, K), np.float16, "row")
    hip.tf_nhwc_to_nchw_f16(wt.ptr, w.ptr, 1, N, 1, K, _sh())      # (K, N) -> (N, K)
    return linear_f16(x.view((M, K), "row"), wt, b.view((N,), "row"))


# cublasOperation_t (native/cublas/ops.py:55-58)
CUBLAS_OP_N, CUBLAS_OP_T, CUBLAS_OP_C = 0, 1, 2


def linear_cublas(weight, x, bias):
    """ff/linear.py:82-110, the reference's fp32 cuBLAS path (tests/linear.py:64-95): ``weight`` (m, k), ``x`` (k, n) and
    ``bias`` (n,) or None are evaluated storage Tensors (fp32, row-major); the result is the flat (m*n) fp32 Tensor that
    cublasSgemm(OP_T, OP_T, m, n, k, weight, lda=k, x, ldb=n, C, ldc=m) leaves behind, i.e. weight @ x stored
    column-major (``res.data.reshape(n, m).T`` on the host); bias[j] is added to column j of it, as Device.add_bias(res, bias, m, n)
    does in the reference (tests/linear.py:77-95 compares with ``torch.matmul(w, x) + bias`` for a (1, N) bias)."""
    from ..storage.tensor import Tensor
    m, k = weight.shape[0], weight.shape[1]
    n = x.shape[1]
    assert x.shape[0] == k, (weight.shape, x.shape)
    res = Tensor.zeros((m * n), dtype=np.float32).eval()
    hip.tf_sgemm_f32(CUBLAS_OP_T, CUBLAS_OP_T, m, n, k, 1.0, weight.dt_ptr, k, x.dt_ptr, n, 0.0, res.dt_ptr, m, None)
    if bias is not None:
        if not bias.dt_ptr:
            bias.eval()
        hip.tf_add_bias_colmajor_f32(res.dt_ptr, bias.dt_ptr, m, n, None)
    return res


def gemm_batch(W, X):
    """ff/linear.py:8-64: batched fp32 GEMM of host Tensors W (B, M, K) and X (B, K, N) through cublasSgemmBatched(OP_T, OP_T):
    returns the list of B device pointers, each holding W[i] @ X[i] stored column-major (M x N, ldc = M), as the reference
    does (tests/linear.py:97-110 copies them back with cudaMemcpy).  The caller owns the buffers (tf_free)."""
    import ctypes
    B, M, K = W.shape
    N = X.shape[2]
    wd, xd = np.ascontiguousarray(W.data, dtype=np.float32), np.ascontiguousarray(X.data, dtype=np.float32)
    ptrs = []
    for arr, nbytes in ((wd, M * K * 4), (xd, K * N * 4), (None, M * N * 4)):
        col = []
        for i in range(B):
            p = ctypes.c_void_p()
            hip.tf_malloc(ctypes.byref(p), nbytes)
            if arr is not None:
                hip.tf_memcpy(p, arr[i].ctypes.data, nbytes, 1)
            col.append(p)
        ptrs.append(col)
    tables = []
    for col in ptrs:                                   # device arrays of device pointers, as cuBLAS wants them
        host = (ctypes.c_void_p * B)(*[c.value for c in col])
        t = ctypes.c_void_p()
        hip.tf_malloc(ctypes.byref(t), ctypes.sizeof(host))
        hip.tf_memcpy(t, ctypes.addressof(host), ctypes.sizeof(host), 1)
        tables.append(t)
    hip.tf_sgemm_batched_f32(CUBLAS_OP_T, CUBLAS_OP_T, M, N, K, 1.0, tables[0], K, tables[1], N, 0.0, tables[2], M, B, None)
    hip.tf_device_sync()
    for t in tables:
        hip.tf_free(t)
    for p in ptrs[0] + ptrs[1]:
        hip.tf_free(p)
    return ptrs[2]


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
