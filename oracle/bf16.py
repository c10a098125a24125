"""bfloat16 rounding for the bf16 parity tests -- TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference's op tests run bfloat16 next to float16 (tests/group_norm.py:12-19, tests/layer_norm.py:13-27 with atol = rtol = 0.125,
tests/linear.py:13).  The HIP entries (tf_*_bf16) read bfloat16 tensors, compute in fp32 and round the result to bfloat16 once; the
oracle side of a test therefore feeds ``round_bf16(inputs)`` to the fp32 operators of oracle.ops and rounds nothing else."""
import numpy as np
import torch


def round_bf16(x):
    """float array -> float32 array on the bfloat16 grid (round to nearest even: torch's conversion)."""
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
    return t.to(torch.bfloat16).to(torch.float32).numpy()


def bits(x):
    """float array -> the uint16 bit patterns of its bfloat16 rounding (torch's conversion)."""
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
    return t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
