"""update_state -- the weight-name contract of tinyfusers/storage/state.py:4-23: attribute paths of the module tree ARE the
LDM checkpoint keys, and the leaves named ``weight`` / ``bias`` are what a checkpoint fills.  Both entry points of this file
(``update_state``: install tensors; ``param_shapes``: list names and shapes without touching the device) ride on one walker,
``_slots``.  An installed leaf becomes an fp16 DeviceArray (4-D conv weights are stored KRSC by the NHWC rule), uploaded once."""
import types

import numpy as np

from .tensor import DeviceArray, asarray

_LEAVES = ("weight", "bias")
_OPAQUE = (DeviceArray, np.ndarray, int, float, str, bool, bytes, type, types.FunctionType, types.BuiltinFunctionType, types.MethodType)


def _children(node):
    """(name, child, owner) of everything below ``node`` that can hold parameters, in the reference's visiting order: attributes
    of an object, fields of a namedtuple, items of a list / tuple (named by index), entries of a dict.  ``owner`` is the dict a
    leaf of that name must be written back into (None where the container cannot be assigned through: tuples)."""
    if isinstance(node, dict):
        return [(str(k), v, node) for k, v in list(node.items())]
    if hasattr(node, "_asdict"):
        return [(k, v, None) for k, v in node._asdict().items()]
    if isinstance(node, (list, tuple)):
        return [(str(i), v, None) for i, v in enumerate(node)]
    if hasattr(node, "__dict__"):
        return [(k, v, node.__dict__) for k, v in list(node.__dict__.items())]
    return []


def _slots(root, prefix=""):
    """Yield (dotted name, owner dict, key, current value, module) for every ``weight`` / ``bias`` slot below ``root``.  Private
    attributes (``_x``), scalars and arrays that are not such leaves are not descended into; ``module`` is the object the slot
    belongs to (``param_shapes`` reads the shape off it)."""
    # Depth first, siblings in declaration order.  Only CYCLES are cut (a public back-reference: a module holding its parent, a compiled graph
    # reachable from the root): a node is skipped when it is one of its own ancestors on the current path.  A module reachable under two names
    # (shared between two parents) is walked under BOTH, so update_state fills it from either key and param_shapes lists both names, as the
    # reference's recursive walk does (storage/state.py:4-23).
    def walk(pre, node, path):
        if id(node) in path:
            return
        path = path | {id(node)}
        for name, child, owner in _children(node):
            dotted = f"{pre}.{name}" if pre else name
            if name in _LEAVES and owner is not None:
                yield dotted, owner, name, child, node
            elif name.startswith("_") or child is None or isinstance(child, _OPAQUE):
                continue
            else:
                yield from walk(dotted, child, path)
    yield from walk(prefix, root, frozenset())


def _to_numpy(v):
    if hasattr(v, "numpy") and not isinstance(v, np.ndarray):
        v = v.numpy()                                      # a torch tensor, as the reference's loader hands over (state.py:20)
    return np.asarray(v)


def update_state(obj, state_dict, prefix=''):
    """Install ``state_dict[name]`` into every weight / bias slot below ``obj`` (storage/state.py:4-23: same names, same
    ``skipped: <name>`` line for a key the checkpoint lacks -- except the bias probes of the bias-free q / k / v projections,
    which the reference prints on every load)."""
    for name, owner, key, cur, _ in _slots(obj, prefix):
        if name not in state_dict:
            if cur is not None or not name.endswith((".to_q.bias", ".to_k.bias", ".to_v.bias")):
                print(f"skipped: {name}")
            continue
        from .. import config
        from .tensor import bfloat16
        owner[key] = asarray(_to_numpy(state_dict[name]), bfloat16 if config.is_bf16() else np.float16)


def _leaf_shape(module, key):
    from ..ff.embedding import Embedding
    from ..ff.group_norm import GroupNorm
    from ..ff.layer_norm import LayerNorm
    from ..ff.linear import Linear
    from ..vision.conv2d import Conv2d
    if isinstance(module, Linear):
        if key == "weight":
            return (module.out_features, module.in_features)
        return (module.out_features,) if module._has_bias else None
    if isinstance(module, Conv2d):
        return tuple(module._shape) if key == "weight" else (module._shape[0],)
    if isinstance(module, GroupNorm):
        return (module.num_channels,)
    if isinstance(module, LayerNorm):
        return tuple(module.normalized_shape)
    if isinstance(module, Embedding):
        return (module.vocab_sz, module.embed_sz) if key == "weight" else None
    return None


def param_shapes(obj, prefix=""):
    """name -> logical shape of every weight / bias leaf below ``obj`` (any of the package's modules, lists / namedtuples of
    them, or the whole StableDiffusion), by the walk update_state makes; nothing needs to be initialised."""
    shapes = {}
    for name, _, key, _, module in _slots(obj, prefix):
        shape = _leaf_shape(module, key)
        if shape is not None:
            shapes[name] = shape
    return shapes


def unet_param_shapes(unet):
    """name -> logical shape of every weight/bias leaf of a (possibly uninitialised) UNetModel, by the same walk."""
    return param_shapes(unet)
