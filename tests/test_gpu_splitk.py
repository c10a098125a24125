"""Split-K partial slabs in fp16 (round 4, tf_gemm_splitk_partials(16), the default) against fp32 slabs and against the oracle: the reducer
accumulates in fp32 in split order either way; a slab element carries one extra fp16 rounding (2^-11 relative) of a PARTIAL sum.  Reference
ops: vision/conv2d.py:9-28 (the convs that run split-K are the 32 x 32 ... 8 x 8 levels of the UNet), ff/linear.py:112-121."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    import tinyfusers_amd.storage.tensor as T
    T.ensure_init(0)
    return T


@pytest.mark.parametrize("n,c,hw,cout,force,gn", [(2, 1280, 8, 1280, (64, 128, 16), 32), (2, 640, 16, 1280, (128, 160, 8), 32), (2, 320, 32, 640, (128, 128, 4), 0),
                                                   (2, 1280, 16, 1280, (64, 160, 8), 32)])
def test_conv_split_k_with_fp16_and_fp32_partials(tf, n, c, hw, cout, force, gn):
    from oracle import ops as O
    from tinyfusers_amd.native import hip, lib
    from tinyfusers_amd.storage.synth import synth_normal
    from tinyfusers_amd.vision.conv2d import Conv2d
    from tinyfusers_amd.ff.group_norm import GroupNorm
    r = lambda name, shape, std=1.0: synth_normal(41, name, shape, std).astype(np.float16).astype(np.float32)
    x, w, b = r("sk.x", (n, c, hw, hw)), r("sk.w", (cout, c, 3, 3), (9 * c) ** -0.5), r("sk.b", (cout,), 0.1)
    conv = Conv2d(c, cout, [3, 3], padding=[1, 1], init=False)
    conv.weight = tf.DeviceArray.from_numpy(w, np.float16, "nhwc"); conv.bias = tf.DeviceArray.from_numpy(b, np.float16, "row")
    g = GroupNorm(32, cout, init=False)
    g.weight = tf.DeviceArray.from_numpy(1 + r("sk.g", (cout,), 0.1), np.float16, "row"); g.bias = tf.DeviceArray.from_numpy(r("sk.gb", (cout,), 0.1), np.float16, "row")
    xd = tf.DeviceArray.from_numpy(x, np.float16, "nhwc")
    want = O.conv2d_bias(x, w, b, (1, 1)).numpy()
    outs = {}
    try:
        for bits in (32, 16):
            hip.tf_gemm_splitk_partials(bits)
            lib.tf_gemm_force_config(*force)
            y = conv(xd, gn=gn, out_norm=(g, True) if gn else None)
            lib.tf_gemm_force_config(0, 0, 0)
            outs[bits] = (y.numpy(), g(y, silu=True).numpy() if gn else None)
    finally:
        lib.tf_gemm_force_config(0, 0, 0)
        hip.tf_gemm_splitk_partials(16)
    for bits in (32, 16):
        np.testing.assert_allclose(outs[bits][0], want, rtol=1e-2, atol=1e-2)
        if gn:
            zw = O.silu(O.group_norm_affine(torch.from_numpy(outs[bits][0]), 32, g.weight.numpy(), g.bias.numpy())).numpy()
            np.testing.assert_allclose(outs[bits][1], zw, rtol=1e-2, atol=1e-2)
    # the two slab types agree to a few output roundings: the slab rounding (2^-11 of a partial) is of the size of the output's own rounding
    d = np.abs(outs[16][0] - outs[32][0])
    assert d.max() <= 4e-3 * (1 + np.abs(want).max()) and (d > 0).mean() < 0.6
    rel = lambda a: float(np.linalg.norm(a - want) / np.linalg.norm(want))
    assert rel(outs[16][0]) <= 1.5 * rel(outs[32][0]) + 1e-4, (rel(outs[16][0]), rel(outs[32][0]))


@pytest.mark.parametrize("split,big", [(8, False), (32, False), (8, True)])
def test_large_cancelling_partials_fp16_slabs_against_fp32_slabs(tf, split, big):
    """a split-K PARTIAL may be large where the full sum is small (cancellation along K), and an fp16 slab rounds every partial to 11
    bits before the reduce.  A linear whose K range is +big in its first half and -big in its second (exact small-integer products, so the fp32
    slabs give the exact result): the fp16-slab result stays finite and within splits x 2^-11 x max|partial| of it; partials beyond fp16's
    range saturate at +-65504 instead of turning the output into inf / NaN (the stated price of the half-size seam)."""
    from tinyfusers_amd.native import hip, lib
    from tinyfusers_amd.ff.linear import linear_f16
    M, N, K = 128, 256, 64 * split * 4
    rng = np.random.default_rng(7)
    x = rng.integers(-4, 5, (M, K)).astype(np.float32)
    w = rng.integers(-4, 5, (N, K)).astype(np.float32)
    # rows of +-16: every split's partial of column 0 is +16 * 16 * (K / split) ... cancelling pairwise over the splits
    sign = np.repeat(np.where(np.arange(split) % 2 == 0, 1.0, -1.0), K // split).astype(np.float32)
    x[:, :] = np.where((np.arange(K)[None, :] % 2 == 0) | big, 16.0 * sign[None, :], x)
    w[0, :] = np.where((np.arange(K) % 2 == 0) | big, 16.0, w[0, :])                      # big: column 0's partials are +-65536 > 65504
    want = x.astype(np.float64) @ w.astype(np.float64).T
    per_split = np.abs((x[:, :K // split].astype(np.float64) @ w[:, :K // split].astype(np.float64).T)).max()
    xd, wd = tf.DeviceArray.from_numpy(x, np.float16, "row"), tf.DeviceArray.from_numpy(w, np.float16, "row")
    outs = {}
    try:
        for bits in (32, 16):
            hip.tf_gemm_splitk_partials(bits)
            lib.tf_gemm_force_config(64, 64, split)
            outs[bits] = linear_f16(xd, wd).numpy().astype(np.float64)
    finally:
        lib.tf_gemm_force_config(0, 0, 0)
        hip.tf_gemm_splitk_partials(16)
    h16 = lambda a: a.astype(np.float16).astype(np.float64)
    np.testing.assert_array_equal(outs[32], h16(want))               # fp32 slabs: exact integers, one output rounding
    assert np.isfinite(outs[16]).all()
    assert (per_split > 65504) == big
    if per_split <= 65504:
        bound = split * 2.0 ** -11 * per_split + np.abs(want) * 2.0 ** -10 + 1e-6
        assert (np.abs(outs[16] - want) <= bound).all(), float(np.abs(outs[16] - want).max())
    else:
        # saturated partials: finite, wrong only in the columns whose partials left the range
        bad = np.abs(outs[16] - h16(want)) > split * 2.0 ** -11 * 65504
        assert bad[:, 1:].mean() < 0.5
