"""update_state -- mirrors tinyfusers/storage/state.py:4-23: recursive walk over __dict__ / namedtuple / list /
dict building dotted LDM checkpoint names and replacing each ``weight`` / ``bias`` leaf.  The leaf becomes an
fp16 DeviceArray (4-D conv weights are stored KRSC by the NHWC rule), uploaded once."""
from collections import OrderedDict

import numpy as np

from .tensor import DeviceArray, asarray


def _to_numpy(v):
    if hasattr(v, "numpy") and not isinstance(v, np.ndarray):
        v = v.numpy()                      # torch tensor, as in state.py:20
    return np.asarray(v)


def update_state(obj, state_dict, prefix=''):
    if isinstance(obj, DeviceArray) or isinstance(obj, np.ndarray):
        return
    if hasattr(obj, '__dict__') and not isinstance(obj, type):
        update_state(obj.__dict__, state_dict, f"{prefix}")
    elif hasattr(obj, '_asdict'):
        update_state(obj._asdict(), state_dict, prefix)
    elif isinstance(obj, OrderedDict):
        update_state(dict(obj), state_dict, prefix)
    elif isinstance(obj, (list, tuple)):
        for i, x in enumerate(obj):
            update_state(x, state_dict, f"{prefix}.{str(i)}")
    elif isinstance(obj, dict):
        for k, v in list(obj.items()):
            if k in {"weight", "bias"}:
                if f"{prefix}.{k}" not in state_dict:
                    if v is not None or not prefix.endswith((".to_q", ".to_k", ".to_v")):
                        print(f"skipped: {prefix}.{k}")
                    continue
                obj[k] = asarray(_to_numpy(state_dict[f"{prefix}.{k}"]), np.float16)
            elif k.startswith("_") or v is None or isinstance(v, (int, float, str, bool)):
                continue
            else:
                pre = f"{prefix}.{k}" if prefix != '' else f"{k}"
                update_state(v, state_dict, f"{pre}")


def unet_param_shapes(unet):
    """name -> logical shape of every weight/bias leaf of a (possibly uninitialised) UNetModel, by the same walk."""
    from ..vision.unet import UNetModel  # noqa
    shapes = {}
    cfg = unet.cfg
    emb = cfg.model_channels * 4

    def visit(obj, prefix):
        from ..ff.linear import Linear
        from ..ff.group_norm import GroupNorm
        from ..ff.layer_norm import LayerNorm
        from ..vision.conv2d import Conv2d
        if isinstance(obj, Linear):
            shapes[prefix + ".weight"] = (obj.out_features, obj.in_features)
            if obj._has_bias:
                shapes[prefix + ".bias"] = (obj.out_features,)
        elif isinstance(obj, Conv2d):
            shapes[prefix + ".weight"] = obj._shape
            shapes[prefix + ".bias"] = (obj._shape[0],)
        elif isinstance(obj, GroupNorm):
            shapes[prefix + ".weight"] = (obj.num_channels,); shapes[prefix + ".bias"] = (obj.num_channels,)
        elif isinstance(obj, LayerNorm):
            shapes[prefix + ".weight"] = obj.normalized_shape; shapes[prefix + ".bias"] = obj.normalized_shape
        elif isinstance(obj, (list, tuple)):
            for i, x in enumerate(obj):
                visit(x, f"{prefix}.{i}")
        elif hasattr(obj, "__dict__") and not callable(obj) or (hasattr(obj, "__dict__") and not isinstance(obj, type(lambda: 0))):
            for k, v in obj.__dict__.items():
                if k.startswith("_") or k == "cfg":
                    continue
                visit(v, f"{prefix}.{k}" if prefix else k)
    visit(unet, "")
    return shapes
