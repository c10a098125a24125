"""Deterministic synthetic weights / inputs for the SD-1.x UNet path (no checkpoint exists offline).

Build-owned (the reference ships no generator: its default init -- U(-sqrt3,sqrt3) conv weights,
all-ones Linear, vision/conv2d.py:52, ff/linear.py:114 -- overflows fp16 within a few layers).
Every tensor is drawn from its own Philox stream keyed by (seed, crc32(name)), so any subset can be
regenerated independently and both the HIP path and the CPU oracle see bit-identical values.
Values are rounded to fp16 once here; the oracle uses those fp16 values widened to fp32.
"""
import zlib

import numpy as np

__all__ = ["synth_tensor", "synth_state_dict", "synth_normal"]


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def synth_normal(seed, name, shape, std=1.0, dtype=np.float32):
    """N(0, std^2) tensor from the (seed, name) stream."""
    x = _rng(seed, name).standard_normal(size=shape, dtype=np.float32)
    if std != 1.0:
        x *= np.float32(std)
    return x.astype(dtype, copy=False)


def synth_tensor(seed, name, shape):
    """One UNet parameter by LDM name: conv/linear weight ~ N(0, 1/fan_in); their bias ~ N(0, 0.02^2);
    norm gamma ~ 1 + 0.1 N, beta ~ 0.1 N.  Returned as fp16."""
    shape = tuple(shape)
    is_norm = len(shape) == 1 and (".norm" in name or name.endswith("in_layers.0.weight") or name.endswith("in_layers.0.bias")
                                   or name.endswith("out_layers.0.weight") or name.endswith("out_layers.0.bias")
                                   or name.startswith("out.0.") or "layer_norm" in name)
    x = _rng(seed, name).standard_normal(size=shape, dtype=np.float32)
    if name.endswith(".weight") and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        x *= np.float32(1.0 / np.sqrt(fan_in))
    elif is_norm and name.endswith(".weight"):
        x = np.float32(1.0) + np.float32(0.1) * x
    elif is_norm:
        x *= np.float32(0.1)
    else:
        x *= np.float32(0.02)
    return x.astype(np.float16)


def synth_state_dict(shapes, seed=0, prefix=""):
    """name -> fp16 array for every entry of ``shapes`` (name -> shape)."""
    return {prefix + k: synth_tensor(seed, k, s) for k, s in shapes.items()}
