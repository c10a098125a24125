"""bfloat16 forms of the operators the reference tests in that type (tests/group_norm.py:12-19, tests/layer_norm.py:13-27,
tests/linear.py:13): HIP entries tf_*_bf16 against the fp32 oracle on bfloat16-rounded inputs.  The reference's own tolerance for
bfloat16 is atol = rtol = 0.125; one bfloat16 rounding of an fp32-exact result is 2^-8 relative, which is what is asserted here."""
import numpy as np
import pytest

import oracle
from oracle import bf16 as OB

pytestmark = pytest.mark.gpu

BF_TOL = dict(rtol=2 ** -7, atol=2 ** -7)       # one output rounding (2^-8) with headroom; far inside the reference's 0.125


def _bf(x, layout=None):
    from tinyfusers_amd.storage.tensor import DeviceArray, bfloat16
    return DeviceArray.from_numpy(np.asarray(x, dtype=np.float32), bfloat16, layout)


@pytest.mark.parametrize("c", [768, 1024, 1280, 1600])
def test_group_norm_bf16_reference_shapes(c):
    """tests/group_norm.py:21-41 in bfloat16: (2048, C, 2, 2), 2 groups, no affine."""
    from tinyfusers_amd.ff.group_norm import group_norm
    from tinyfusers_amd.storage.tensor import is_bfloat16
    x = OB.round_bf16(np.random.default_rng(c).standard_normal((2048, c, 2, 2)))
    y = group_norm(_bf(x), 2, 1e-5)
    assert is_bfloat16(y.dtype)
    np.testing.assert_allclose(y.numpy(), oracle.group_norm(x, 2, 1e-5).numpy(), **BF_TOL)


@pytest.mark.parametrize("shape,groups,silu", [((2, 320, 64, 64), 32, True), ((3, 640, 17, 9), 32, False), ((1, 64, 5, 7), 8, True)])
def test_group_norm_module_bf16(shape, groups, silu):
    from tinyfusers_amd.ff.group_norm import GroupNorm
    rng = np.random.default_rng(1)
    x = OB.round_bf16(rng.standard_normal(shape) * 2 + 0.5)
    g, b = OB.round_bf16(rng.standard_normal(shape[1])), OB.round_bf16(rng.standard_normal(shape[1]))
    m = GroupNorm(groups, shape[1], init=False)
    m.weight, m.bias = _bf(g), _bf(b)
    ref = oracle.group_norm_affine(x, groups, g, b, 1e-5)
    if silu:
        ref = oracle.silu(ref)
    np.testing.assert_allclose(m(_bf(x), silu=silu).numpy(), ref.numpy(), rtol=2 ** -7, atol=2 ** -6)


@pytest.mark.parametrize("c", [768, 1024, 1280, 1600])
def test_layer_norm_bf16_last_dim(c):
    """tests/layer_norm.py in bfloat16, last-dim form: (2048, C) rows."""
    from tinyfusers_amd.ff.layer_norm import layer_norm
    rng = np.random.default_rng(c)
    x = OB.round_bf16(rng.standard_normal((2048, c)))
    g, b = OB.round_bf16(rng.standard_normal(c)), OB.round_bf16(rng.standard_normal(c))
    y = layer_norm(_bf(x, "row"), _bf(g, "row"), _bf(b, "row"), np.full((1, 1, 1, 1), 1e-3, np.float32))
    np.testing.assert_allclose(y.numpy(), oracle.layer_norm(x, g, b, 1e-3).numpy(), rtol=2 ** -7, atol=2 ** -6)


def test_layer_norm_bf16_slab_and_torch_inputs():
    """tests/layer_norm.py:22-41 verbatim in shape: torch bfloat16 tensors, (N, C, 10, 10) normalised over [C, H, W] (N cut to 64)."""
    import torch
    from tinyfusers_amd.ff.layer_norm import layer_norm
    from tinyfusers_amd.storage.tensor import is_bfloat16
    torch.manual_seed(0)
    n, c, h, w = 64, 768, 10, 10
    x = torch.randn(n, c, h, w, dtype=torch.bfloat16)
    scale, bias = torch.randn(1, c, h, w, dtype=torch.bfloat16), torch.randn(1, c, h, w, dtype=torch.bfloat16)
    y = layer_norm(x, scale, bias, torch.full((1, 1, 1, 1), 1e-3))
    assert is_bfloat16(y.dtype)
    ref = torch.nn.functional.layer_norm(x.float(), [c, h, w], scale.float().squeeze(0), bias.float().squeeze(0), 1e-3)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=2 ** -7, atol=2 ** -6)
    # odd last dim (tests/layer_norm.py:44-71's 10-element rows)
    x2 = OB.round_bf16(np.random.default_rng(2).standard_normal((10, 32, 10, 10)))
    g2, b2 = OB.round_bf16(np.random.default_rng(3).standard_normal(10)), OB.round_bf16(np.random.default_rng(4).standard_normal(10))
    y2 = layer_norm(_bf(x2, "row"), _bf(g2, "row"), _bf(b2, "row"), 1e-3)
    np.testing.assert_allclose(y2.numpy(), oracle.layer_norm(x2, g2, b2, 1e-3).numpy(), rtol=2 ** -7, atol=2 ** -6)


@pytest.mark.parametrize("m,n,k,bias,res", [(4096, 320, 320, True, False), (8192, 1280, 320, True, True), (154, 768, 768, True, False),
                                            (77, 40, 1000, False, False), (1000, 2000, 1000, True, False), (512, 1280, 5120, True, True)])
def test_linear_bf16(m, n, k, bias, res):
    """y = x w^T + b (+ residual), every tensor bfloat16, fp32 accumulation: against float64 on the same bfloat16 values."""
    from tinyfusers_amd.ff.linear import Linear
    rng = np.random.default_rng(m + n + k)
    x, w = OB.round_bf16(rng.standard_normal((m, k))), OB.round_bf16(rng.standard_normal((n, k)) / np.sqrt(k))
    b = OB.round_bf16(rng.standard_normal(n)) if bias else None
    r = OB.round_bf16(rng.standard_normal((m, n))) if res else None
    lin = Linear(k, n, bias=bias, init=False)
    lin.weight, lin.bias = _bf(w, "row"), (_bf(b, "row") if bias else None)
    y = lin(_bf(x, "row"), residual=_bf(r, "row") if res else None).numpy()
    ref = x.astype(np.float64) @ w.astype(np.float64).T + (b if bias else 0.0) + (r if res else 0.0)
    np.testing.assert_allclose(y, ref, rtol=2 ** -7, atol=2 ** -6)


def test_linear_bf16_exact_integers():
    """small integers are exact in bfloat16 operands and in the fp32 accumulator, so the result must be the bfloat16 rounding of the
    exact sum, bit for bit (catches a wrong operand layout, which a tolerance could hide) -- both tiles (64 x 64 and 128 x 128), ragged edges."""
    from tinyfusers_amd.ff.linear import linear_bf16
    rng = np.random.default_rng(5)
    for m, n, k in [(200, 136, 192), (4100, 1030, 64)]:
        x, w = rng.integers(-4, 5, (m, k)).astype(np.float32), rng.integers(-4, 5, (n, k)).astype(np.float32)
        b = rng.integers(-8, 9, n).astype(np.float32)
        ref = x @ w.T + b
        y = linear_bf16(_bf(x, "row"), _bf(w, "row"), _bf(b, "row")).numpy()
        np.testing.assert_array_equal(y, OB.round_bf16(ref))


@pytest.mark.parametrize("n,c,h,w,k,r,stride,pad,bias", [(2, 320, 32, 32, 320, 3, 1, 1, True), (1, 64, 17, 23, 128, 3, 2, 1, False),
                                                         (2, 640, 16, 16, 320, 1, 1, 0, True), (1, 40, 20, 20, 72, 3, 1, 1, True)])
def test_conv2d_bf16(n, c, h, w, k, r, stride, pad, bias):
    """vision/conv2d.py:9-28 on bfloat16 tensors (the last case: channel counts off the 64 grid -> the GENERIC instance)."""
    from tinyfusers_amd.vision.conv2d import conv2d_bf16
    rng = np.random.default_rng(n * 1000 + c)
    x = OB.round_bf16(rng.standard_normal((n, c, h, w)))
    wt = OB.round_bf16(rng.standard_normal((k, c, r, r)) / np.sqrt(c * r * r))
    b = OB.round_bf16(rng.standard_normal(k)) if bias else None
    y = conv2d_bf16(_bf(x), _bf(wt), _bf(b, "row") if bias else None, [pad, pad], [stride, stride], [1, 1]).numpy()
    ref = oracle.conv2d_bias(x, wt, b, (pad, pad), (stride, stride)).numpy()
    np.testing.assert_allclose(y, ref, rtol=2 ** -7, atol=2 ** -6)
