"""Host side of the bfloat16 tensors (no GPU): the numpy bit conversion of storage/tensor.py against torch's."""
import numpy as np

from oracle import bf16 as OB
from tinyfusers_amd.storage.tensor import bf16_bits_to_f32, bfloat16, f32_to_bf16_bits, is_bfloat16


def test_bf16_bits_match_torch_rounding():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * s for s in (1e-3, 1.0, 1e3, 1e30)])
    x = np.concatenate([x, np.array([0.0, -0.0, np.inf, -np.inf, 1.0, 3.3895314e38, 1e-40, 1.00390625, 1.01171875], np.float32)])
    b = f32_to_bf16_bits(x)
    assert b.dtype == np.uint16 and np.array_equal(b, OB.bits(x))                     # round to nearest even, ties included
    assert np.array_equal(bf16_bits_to_f32(b), OB.round_bf16(x))
    assert np.isnan(bf16_bits_to_f32(f32_to_bf16_bits(np.array([np.nan], np.float32))))[0]


def test_bf16_dtype_tag():
    assert is_bfloat16(bfloat16) and not is_bfloat16(np.dtype(np.uint16)) and not is_bfloat16(np.dtype(np.float16))
    assert np.dtype(bfloat16) is bfloat16 and bfloat16.itemsize == 2
