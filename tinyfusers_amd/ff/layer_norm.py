"""layer_norm / LayerNorm -- mirrors tinyfusers/ff/layer_norm.py:8-49 (a cuDNN graph built per call).
Semantics = standard last-dim LayerNorm (torch.nn.functional.layer_norm, the oracle of the reference's own
tests/layer_norm.py:38-41).  One wave per row, row held in registers."""
import numpy as np

from ..native import hip
from ..storage.tensor import DeviceArray, _sh, asarray, is_bfloat16


def layer_norm(x, scale, bias, eps):
    """ff/layer_norm.py:8-32.  Normalises over as many trailing dims as ``scale`` spans (one by default): the reference's own tests
    use a (C, H, W) slab and a 10-element last dim (tests/layer_norm.py:22-71).  x, scale and bias must share their storage order over
    the normalised dims (all 'nhwc' 4-D arrays, or all row-major)."""
    x = asarray(x)                                     # (the reference's test hands torch tensors, tests/layer_norm.py:38-41)
    scale = asarray(scale, x.dtype) if scale is not None else None
    bias = asarray(bias, x.dtype) if bias is not None else None
    c = x.shape[-1]
    if scale is not None and scale.size != c:
        c = scale.size
        tail, n = 1, 0
        while tail < c:
            n += 1
            tail *= x.shape[-n]
        assert tail == c and (n == 1 or (scale.layout == x.layout)), (x.shape, scale.shape)
    rows = x.size // c
    y = DeviceArray.empty(x.shape, x.dtype, x.layout)
    e = float(np.asarray(eps).reshape(-1)[0])
    fn = hip.tf_layer_norm_bf16 if is_bfloat16(x.dtype) else hip.tf_layer_norm_f16
    fn(y.ptr, x.ptr, scale.ptr if scale is not None else None, bias.ptr if bias is not None else None, rows, c, e, _sh())
    return y


class LayerNorm:
    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True, init=True):
        self.normalized_shape = (normalized_shape,) if isinstance(normalized_shape, int) else tuple(normalized_shape)
        self.elementwise_affine = elementwise_affine
        c = int(np.prod(self.normalized_shape))
        self.weight = (asarray(np.ones(c, dtype=np.float16)) if init else None) if elementwise_affine else None
        self.bias = (asarray(np.zeros(c, dtype=np.float16)) if init else None) if elementwise_affine else None
        self.eps = np.full((1, 1, 1, 1), eps, dtype=np.float32)   # same host-side holder as ff/layer_norm.py:40

    def __call__(self, x):
        k = len(self.normalized_shape)
        assert self.normalized_shape == tuple(x.shape[-k:]), f"last dimensions of {x.shape} must match {self.normalized_shape}"
        return layer_norm(x, self.weight, self.bias, self.eps)
