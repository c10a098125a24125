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


# ---- round 5: the bfloat16 step on the TUNED path: every kernel family has bf16 instances, attention is native bfloat16 ----------
def _dev16(x, bf, layout=None):
    from tinyfusers_amd.storage.tensor import DeviceArray, bfloat16
    return DeviceArray.from_numpy(np.asarray(x, dtype=np.float32), bfloat16 if bf else np.float16, layout)


@pytest.mark.parametrize("cfg", [(64, 64, 1, 0), (64, 160, 1, 256), (128, 128, 1, 16), (128, 160, 4, 0), (64, 128, 8, 256), (256, 128, 1, 0), (128, 128, 1, 1024),
                                 (256, 160, 1, 512), (192, 128, 2, 512), (256, 256, 1, 512), (256, 128, 1, 16384)])
def test_bf16_linear_is_exact_on_small_integers_on_every_kernel(cfg):
    """Small-integer operands: every product and partial sum is exact in bf16 x bf16 -> fp32, so each kernel family's bf16 instance (deep / wide /
    ALL8 rings, split-K with fp32 slabs + the bf16 reducers, the 256-row tile, k_gemm_c4, k_igemm_pp incl. 256 x 256) must reproduce the integer
    result bit for bit; bias and residual ride in the epilogue."""
    from tinyfusers_amd.ff.linear import linear_f16
    from tinyfusers_amd.native import lib
    bm, bn, split, flag = cfg
    M, N, K = 2048, 1280, 1024
    rng = np.random.default_rng(sum(cfg))
    x, w = rng.integers(-3, 4, (M, K)).astype(np.float32), rng.integers(-3, 4, (N, K)).astype(np.float32)
    b, r = rng.integers(-8, 9, N).astype(np.float32), rng.integers(-8, 9, (M, N)).astype(np.float32)
    want = x @ w.T + b + r
    assert np.abs(want).max() < 2 ** 24                  # every partial sum is an exact fp32 integer: the result is ONE bfloat16 rounding of the exact value
    try:
        lib.tf_gemm_force_config(bm, bn, split); lib.tf_gemm_debug(flag)
        y = linear_f16(_dev16(x, True, "row"), _dev16(w, True, "row"), _dev16(b, True, "row"), _dev16(r, True, "row")).numpy()
    finally:
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    if flag & (1024 | 16384):          # the persistent kernels round the tile to 16 bits in front of their row-side residual add (two roundings, as in float16)
        want = OB.round_bf16(x @ w.T + b) + r
    np.testing.assert_array_equal(y, OB.round_bf16(want))


@pytest.mark.parametrize("m,n,k", [(4608, 960, 320), (33000, 392, 320), (1000, 400, 256)])
def test_bf16_linear_is_exact_on_small_integers_on_the_activation_resident_kernel(m, n, k):
    """k_gemm_ar<..., BF> (K = 256 / 320, no residual): the same exact-integer argument; runs that cross panel seams and ragged edges."""
    from tinyfusers_amd.ff.linear import linear_f16
    from tinyfusers_amd.native import lib
    rng = np.random.default_rng(m + n + k)
    x, w = rng.integers(-3, 4, (m, k)).astype(np.float32), rng.integers(-3, 4, (n, k)).astype(np.float32)
    b = rng.integers(-8, 9, n).astype(np.float32)
    try:
        lib.tf_gemm_force_config(128, 128, 1); lib.tf_gemm_debug(32768)
        y = linear_f16(_dev16(x, True, "row"), _dev16(w, True, "row"), _dev16(b, True, "row")).numpy()
    finally:
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    np.testing.assert_array_equal(y, OB.round_bf16(x @ w.T + b))


@pytest.mark.parametrize("shape", [(2, 320, 64, 64, 320), (2, 1280, 8, 8, 1280), (8, 640, 48, 48, 640)])
def test_bf16_conv3x3_all_fusions_against_the_oracle(shape):
    """conv 3x3 with bias + time embedding + residual + the statistics of the next GroupNorm, then that GroupNorm (+ SiLU): the fused entries in bfloat16
    (tf_conv2d_fused_norm_16 / tf_conv2d_gn_16 / the split-K reduce that applies the norm) against the fp32 oracle on bfloat16-rounded inputs; the last
    shape runs the patch form of the ping-pong kernel (k_igemm_pp3<..., BF>), the middle one split-K."""
    from oracle import ops as O
    from tinyfusers_amd.ff.group_norm import GroupNorm
    from tinyfusers_amd.vision.conv2d import Conv2d
    n, c, h, w_, co = shape
    rng = np.random.default_rng(c + h)
    rb = lambda a: OB.round_bf16(a)
    x, wt, b = rb(rng.standard_normal((n, c, h, w_))), rb(rng.standard_normal((co, c, 3, 3)) * (9 * c) ** -0.5), rb(rng.standard_normal(co) * 0.1)
    emb, res = rb(rng.standard_normal((n, co)) * 0.1), rb(rng.standard_normal((n, co, h, w_)))
    g, gb = rb(1 + 0.1 * rng.standard_normal(co)), rb(0.1 * rng.standard_normal(co))
    conv = Conv2d(c, co, [3, 3], padding=[1, 1], init=False)
    conv.weight, conv.bias = _dev16(wt, True, "nhwc"), _dev16(b, True, "row")
    norm = GroupNorm(32, co, init=False)
    norm.weight, norm.bias = _dev16(g, True, "row"), _dev16(gb, True, "row")
    y = conv(_dev16(x, True, "nhwc"), bias_nc=_dev16(emb, True, "row"), residual=_dev16(res, True, "nhwc"), gn=32, out_norm=(norm, True))
    z = norm(y, silu=True)
    want = O.conv2d_bias(x, wt, b, (1, 1)).numpy() + emb[:, :, None, None] + res
    np.testing.assert_allclose(y.numpy(), want, rtol=2 ** -6, atol=2 ** -5)
    zw = O.silu(O.group_norm_affine(y.numpy(), 32, g, gb)).numpy()          # (the norm of what the device stored)
    np.testing.assert_allclose(z.numpy(), zw, rtol=2 ** -6, atol=2 ** -5)


@pytest.mark.parametrize("b,t,nh,hs,tk", [(2, 4096, 8, 40, 4096), (2, 1024, 8, 80, 77), (2, 256, 8, 160, 256), (8, 9216 // 4, 8, 40, 9216 // 4), (1, 77, 12, 64, 77)])
def test_bf16_sdpa_matches_the_oracle(b, t, nh, hs, tk):
    """tf_sdpa_16(bfloat16): the SD head sizes on the LDS-DMA kernels (incl. the eight-wave d = 40 form) against attention/sdpa.py:53-77 restated."""
    from oracle import ops as O
    from tinyfusers_amd.attention.sdpa import sdpa_strided
    from tinyfusers_amd.storage.tensor import DeviceArray, bfloat16
    rng = np.random.default_rng(hs + t)
    q, k, v = (OB.round_bf16(rng.standard_normal((b, nh, n_, hs))) for n_ in (t, tk, tk))
    o = DeviceArray.empty((b, nh, t, hs), bfloat16, "row")
    st = lambda n_: (nh * n_ * hs, n_ * hs, hs)
    sdpa_strided(o, _dev16(q, True, "row"), _dev16(k, True, "row"), _dev16(v, True, "row"), b, nh, t, tk, hs, st(t), st(tk), st(tk), st(t))
    np.testing.assert_allclose(o.numpy(), O.scaled_dot_product_attention(q, k, v).numpy(), rtol=2 ** -6, atol=2 ** -6)


def test_bf16_attention_block_survives_activations_beyond_the_fp16_range():
    """Why one picks bfloat16: CrossAttention (attention/attention.py:26-41) with a value projection whose outputs reach ~3e5 -- inf in float16, where
    round 4's bf16 step sent q / k / v through the fp16 attention kernel -- stays finite and matches the fp32 oracle on bfloat16-rounded operands."""
    from oracle import ops as O
    from tinyfusers_amd import config
    from tinyfusers_amd.attention.attention import CrossAttention
    C, nh, hs, b, t = 320, 8, 40, 2, 1024
    rng = np.random.default_rng(3)
    rb = OB.round_bf16
    x = rb(rng.standard_normal((b, t, C)))
    wq, wk = rb(rng.standard_normal((C, C)) * C ** -0.5), rb(rng.standard_normal((C, C)) * C ** -0.5)
    wv, wo, bo = rb(rng.standard_normal((C, C)) * C ** -0.5 * 1e5), rb(rng.standard_normal((C, C)) * C ** -0.5 * 1e-5), rb(rng.standard_normal(C) * 0.1)
    att = CrossAttention(C, C, nh, hs, init=False)
    att.to_q.weight, att.to_k.weight, att.to_v.weight = (_dev16(w_, True, "row") for w_ in (wq, wk, wv))
    att.to_out[0].weight, att.to_out[0].bias = _dev16(wo, True, "row"), _dev16(bo, True, "row")
    saved = config.head_merge
    try:
        config.head_merge = "intended"
        y = att(_dev16(x, True, "row")).numpy()
    finally:
        config.head_merge = saved
    heads = lambda a: a.reshape(b, t, nh, hs).transpose(0, 2, 1, 3)
    q, k, v = rb(x @ wq.T), rb(x @ wk.T), rb(x @ wv.T)
    assert np.abs(v).max() > 65504 * 2                          # far beyond fp16
    o = rb(O.scaled_dot_product_attention(heads(q), heads(k), heads(v)).numpy().transpose(0, 2, 1, 3).reshape(b, t, C))
    want = o @ wo.T + bo
    assert np.isfinite(y).all()
    assert float(np.linalg.norm(y - want) / np.linalg.norm(want)) < 2e-2
