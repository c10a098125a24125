"""update_state -- mirrors tinyfusers/storage/state.py:4-23: recursive walk over __dict__ / namedtuple / list /
dict building dotted LDM checkpoint names and replacing each ``weight`` / ``bias`` leaf.  The leaf becomes an
fp16 DeviceArray (4-D conv weights are stored KRSC by the NHWC rule), uploaded once."""
from collections import OrderedDict

import types

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


def param_shapes(obj, prefix=""):
    """name -> logical shape of every weight/bias leaf below ``obj`` (any of the package's modules, lists / namedtuples of
    them, or the whole StableDiffusion), by the same attribute walk as update_state; nothing needs to be initialised."""
    from ..ff.embedding import Embedding
    from ..ff.group_norm import GroupNorm
    from ..ff.layer_norm import LayerNorm
    from ..ff.linear import Linear
    from ..vision.conv2d import Conv2d
    shapes = {}

    def visit(o, pre):
        if o is None or isinstance(o, (DeviceArray, np.ndarray, int, float, str, bool)):
            return
        if isinstance(o, Linear):
            shapes[pre + ".weight"] = (o.out_features, o.in_features)
            if o._has_bias:
                shapes[pre + ".bias"] = (o.out_features,)
        elif isinstance(o, Conv2d):
            shapes[pre + ".weight"] = tuple(o._shape)
            shapes[pre + ".bias"] = (o._shape[0],)
        elif isinstance(o, GroupNorm):
            shapes[pre + ".weight"] = (o.num_channels,); shapes[pre + ".bias"] = (o.num_channels,)
        elif isinstance(o, LayerNorm):
            shapes[pre + ".weight"] = tuple(o.normalized_shape); shapes[pre + ".bias"] = tuple(o.normalized_shape)
        elif isinstance(o, Embedding):
            shapes[pre + ".weight"] = (o.vocab_sz, o.embed_sz)
        elif hasattr(o, "_asdict"):
            for k, v in o._asdict().items():
                visit(v, f"{pre}.{k}" if pre else k)
        elif isinstance(o, (list, tuple)):
            for i, x in enumerate(o):
                visit(x, f"{pre}.{i}")
        elif isinstance(o, dict):
            for k, v in o.items():
                visit(v, f"{pre}.{k}" if pre else str(k))
        elif hasattr(o, "__dict__") and not isinstance(o, (type, types.FunctionType, types.BuiltinFunctionType, types.MethodType)):
            for k, v in o.__dict__.items():
                if k.startswith("_") or k in ("cfg", "alphas_cumprod"):
                    continue
                visit(v, f"{pre}.{k}" if pre else k)
    visit(obj, prefix)
    return shapes


def unet_param_shapes(unet):
    """name -> logical shape of every weight/bias leaf of a (possibly uninitialised) UNetModel, by the same walk."""
    return param_shapes(unet)
