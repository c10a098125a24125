"""layer_norm / LayerNorm -- mirrors tinyfusers/ff/layer_norm.py:8-49 (a cuDNN graph built per call).
Semantics = standard last-dim LayerNorm (torch.nn.functional.layer_norm, the oracle of the reference's own
tests/layer_norm.py:38-41).  One wave per row, row held in registers."""
import numpy as np

from ..native import hip
from ..storage.tensor import DeviceArray, _sh, asarray


def layer_norm(x, scale, bias, eps):
    c = x.shape[-1]
    rows = x.size // c
    y = DeviceArray.empty(x.shape, np.float16, x.layout)
    e = float(np.asarray(eps).reshape(-1)[0])
    hip.tf_layer_norm_f16(y.ptr, x.ptr, scale.ptr if scale is not None else None, bias.ptr if bias is not None else None, rows, c, e, _sh())
    return y


class LayerNorm:
    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True, init=True):
        self.normalized_shape = (normalized_shape,) if isinstance(normalized_shape, int) else tuple(normalized_shape)
        assert len(self.normalized_shape) == 1, "only last-dim LayerNorm is on the UNet path"
        self.elementwise_affine = elementwise_affine
        c = self.normalized_shape[0]
        self.weight = (asarray(np.ones(c, dtype=np.float16)) if init else None) if elementwise_affine else None
        self.bias = (asarray(np.zeros(c, dtype=np.float16)) if init else None) if elementwise_affine else None
        self.eps = np.full((1, 1, 1, 1), eps, dtype=np.float32)   # same host-side holder as ff/layer_norm.py:40

    def __call__(self, x):
        assert self.normalized_shape == tuple(x.shape[-1:]), f"last dimensions of {x.shape} must match {self.normalized_shape}"
        return layer_norm(x, self.weight, self.bias, self.eps)
