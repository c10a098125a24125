"""group_norm / GroupNorm -- mirrors tinyfusers/ff/group_norm.py:3-21 (7 CuPy kernels + a device sync).
Here: two HBM-bound launches (partial statistics; normalise + affine [+ SiLU]) on NHWC fp16, and the
channel concat of vision/unet.py:72 can be folded in by passing a pair ``(x, skip)``."""
import numpy as np

from .. import config
from ..native import hip
from ..storage.tensor import DeviceArray, _sh, asarray, dtag, is_bfloat16
from .linear import workspace


def _gn(x, num_groups, eps, gamma, beta, silu):
    x2 = None
    if isinstance(x, (tuple, list)):
        x, x2 = x
    n, c1, h, w = x.shape
    c2 = x2.shape[1] if x2 is not None else 0
    dt = dtag(x.dtype)
    y = DeviceArray.empty((n, c1 + c2, h, w), x.dtype, "nhwc")
    if x2 is None and x.gn is not None and x.gn[2] == num_groups:
        part, chunks, _ = x.gn                         # statistics came with x from the conv that produced it
        hip.tf_group_norm_apply_16(dt, y.ptr, x.ptr, gamma.ptr if gamma is not None else None, beta.ptr if beta is not None else None,
                                   part.ptr, chunks, n, h * w, c1, num_groups, float(eps), 1 if silu else 0, _sh())
        return y
    if config.concat_stats and x2 is not None and x.gn is not None and x2.gn is not None:
        # concat whose statistics came with its two sources: their partials' sub-groups (equal width) tile the concat's groups --
        # 32 + 32 sub-groups for an equal split, 64 + 32 for the 2:1 splits of the UNet's output path
        g1, g2 = x.gn[2], x2.gn[2]
        cpg = (c1 + c2) // num_groups
        if c1 % g1 == 0 and c2 % g2 == 0 and c1 // g1 == c2 // g2 and cpg % (c1 // g1) == 0 and cpg // (c1 // g1) <= 8:
            hip.tf_group_norm_apply_cat_16(dt, y.ptr, x.ptr, x2.ptr, gamma.ptr if gamma is not None else None, beta.ptr if beta is not None else None,
                                           x.gn[0].ptr, x.gn[1], g1, x2.gn[0].ptr, x2.gn[1], g2, n, h * w, c1, c2, num_groups, float(eps),
                                           1 if silu else 0, _sh())
            return y
    nb = hip.tf_group_norm_workspace(n, h * w, c1 + c2, num_groups)
    ws = workspace(nb)
    (hip.tf_group_norm_bf16 if dt else hip.tf_group_norm_f16)(y.ptr, x.ptr, x2.ptr if x2 is not None else None, gamma.ptr if gamma is not None else None,
                                                              beta.ptr if beta is not None else None, n, h * w, c1, c2, num_groups, float(eps), 1 if silu else 0,
                                                              ws.ptr, nb, _sh())
    return y


def group_norm(x, num_groups, eps):
    """ff/group_norm.py:3-11 -- no affine."""
    return _gn(x, num_groups, eps, None, None, False)


class GroupNorm:
    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True, init=True):
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = (asarray(np.ones(num_channels, dtype=np.float16)) if init else None) if affine else None
        self.bias = (asarray(np.zeros(num_channels, dtype=np.float16)) if init else None) if affine else None

    def __call__(self, x, silu=False):
        nd = getattr(x, "normed", None)
        if nd is not None and nd[0] is self and nd[1] == bool(silu):
            return nd[2]                                 # the conv that produced x already applied this very norm (split-K reduce)
        return _gn(x, self.num_groups, self.eps, self.weight, self.bias, silu)
