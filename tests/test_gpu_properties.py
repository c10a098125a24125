"""Size-independent properties at the sizes of BASELINE config 5 (96 x 96 latents, 9216 tokens), where the CPU oracle would take minutes
per case: exact ones where the arithmetic allows (translation equivariance of the convolution, integer linearity of the GEMM, softmax rows
summing to one), tight tolerances elsewhere (GroupNorm invariances, key permutations in attention).  They exercise the same C-ABI entries
as the parity tests of test_gpu_ops.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    import tinyfusers_amd.storage.tensor as T
    T.ensure_init(0)
    return T


def _h(x):
    return np.asarray(x, dtype=np.float16).astype(np.float32)


@pytest.mark.parametrize("c,k,hw", [(320, 320, 96), (640, 320, 48)])
def test_conv3x3_is_translation_equivariant_bit_for_bit(tf, c, k, hw):
    """conv(shift(x)) == shift(conv(x)) away from the border, EXACTLY: every output pixel runs the same K loop in the same order wherever
    it sits in the image, in whichever tile (vision/conv2d.py:9-28 is a plain cross-correlation).  Any position-dependent defect -- a
    halo row fetched from the wrong place, a tile edge, a split-K seam -- breaks the equality."""
    from tinyfusers_amd.vision.conv2d import Conv2d
    rng = np.random.default_rng(c)
    x = _h(rng.standard_normal((2, c, hw, hw)))
    conv = Conv2d(c, k, [3, 3], padding=[1, 1], init=False)
    conv.weight = tf.DeviceArray.from_numpy(_h(rng.standard_normal((k, c, 3, 3)) / np.sqrt(9 * c)))
    conv.bias = tf.DeviceArray.from_numpy(_h(rng.standard_normal(k)), layout="row")
    xs = np.zeros_like(x)
    xs[:, :, 1:, 2:] = x[:, :, :-1, :-2]                  # down by one row, right by two columns
    y, ys = conv(tf.DeviceArray.from_numpy(x)).numpy(), conv(tf.DeviceArray.from_numpy(xs)).numpy()
    assert np.isfinite(y).all() and np.abs(y).max() > 0.5
    # ys[p, q] == y[p - 1, q - 2] wherever neither side touches a border or the rows / columns the shift dropped
    assert np.array_equal(ys[:, :, 2:-1, 4:-1], y[:, :, 1:-2, 2:-3])


def test_linear_is_exactly_linear_on_integers(tf):
    """small integers: every product and partial sum is exact in fp16 operands / fp32 accumulation, so Linear(x1 + x2) == Linear(x1) +
    Linear(x2) - b exactly, at the GEGLU-sized GEMM of config 5's top level (73728 x 2560 x 320) -- whatever tile or split the tuner picked."""
    from tinyfusers_amd.ff.linear import linear_f16
    rng = np.random.default_rng(0)
    m, n, k = 73728, 2560, 320
    x1, x2 = rng.integers(-2, 3, (m, k)).astype(np.float32), rng.integers(-2, 3, (m, k)).astype(np.float32)
    w = rng.integers(-1, 2, (n, k)).astype(np.float32)
    W = tf.DeviceArray.from_numpy(w, layout="row")
    y12 = linear_f16(tf.DeviceArray.from_numpy(x1 + x2, layout="row"), W).numpy()
    y1 = linear_f16(tf.DeviceArray.from_numpy(x1, layout="row"), W).numpy()
    y2 = linear_f16(tf.DeviceArray.from_numpy(x2, layout="row"), W).numpy()
    assert np.abs(y12).max() <= 2048                       # (still exact in the fp16 output)
    assert np.array_equal(y12, y1 + y2)
    rows = rng.integers(0, m, 64)                          # and it is the right sum: spot rows against numpy
    assert np.array_equal(y1[rows], x1[rows] @ w.T)


def test_attention_rows_sum_to_one_at_9216_tokens(tf):
    """V = 1 everywhere: softmax(QK^T / sqrt(d)) V must be 1 (the row sums -- taken from V's padding column at d = 40 -- divide the
    accumulated row exactly); attention/sdpa.py:53-77 at config 5's self-attention size (4 images x CFG, 8 heads, 9216 tokens, d = 40)."""
    from tinyfusers_amd.attention.sdpa import scaled_dot_product_attention
    rng = np.random.default_rng(1)
    b, nh, t, d = 2, 8, 9216, 40
    q = tf.DeviceArray.from_numpy(_h(rng.standard_normal((b, nh, t, d)) * 2), layout="row")
    k = tf.DeviceArray.from_numpy(_h(rng.standard_normal((b, nh, t, d)) * 2), layout="row")
    v = tf.DeviceArray.from_numpy(np.ones((b, nh, t, d), np.float32), layout="row")
    o = scaled_dot_product_attention(q, k, v).numpy()
    np.testing.assert_allclose(o, 1.0, rtol=0, atol=2e-3)


def test_attention_is_invariant_to_a_permutation_of_the_keys(tf):
    """the same (key, value) pairs in another order give the same output up to the summation order (fp32 accumulation, fp16 P)."""
    from tinyfusers_amd.attention.sdpa import scaled_dot_product_attention
    rng = np.random.default_rng(2)
    b, nh, tq, tk, d = 1, 8, 1024, 9216, 40
    q = _h(rng.standard_normal((b, nh, tq, d)))
    k, v = _h(rng.standard_normal((b, nh, tk, d))), _h(rng.standard_normal((b, nh, tk, d)))
    perm = rng.permutation(tk)
    D = lambda a: tf.DeviceArray.from_numpy(a, layout="row")
    o1 = scaled_dot_product_attention(D(q), D(k), D(v)).numpy()
    o2 = scaled_dot_product_attention(D(q), D(np.ascontiguousarray(k[:, :, perm])), D(np.ascontiguousarray(v[:, :, perm]))).numpy()
    np.testing.assert_allclose(o1, o2, rtol=0, atol=2e-3)
    assert np.abs(o1).max() > 0.02


def test_group_norm_invariances_at_96x96(tf):
    """group_norm(a x + b) == group_norm(x) for a > 0 (up to the fp16 rounding of the inputs and eps) and group_norm is idempotent;
    ff/group_norm.py:3-11 at (4, 320, 96, 96), 32 groups."""
    from tinyfusers_amd.ff.group_norm import group_norm
    rng = np.random.default_rng(3)
    x = _h(rng.standard_normal((4, 320, 96, 96)))
    g = group_norm(tf.DeviceArray.from_numpy(x), 32, 1e-5)
    y = g.numpy()
    y2 = group_norm(tf.DeviceArray.from_numpy(_h(4.0 * x + 8.0)), 32, 1e-5).numpy()      # (power-of-two scale: exact in fp16)
    np.testing.assert_allclose(y2, y, rtol=0, atol=6e-3)
    np.testing.assert_allclose(group_norm(g, 32, 1e-5).numpy(), y, rtol=0, atol=4e-3)
    per_group = y.reshape(4, 32, -1)
    np.testing.assert_allclose(per_group.mean(-1), 0.0, atol=2e-3)
    np.testing.assert_allclose(per_group.var(-1), 1.0, atol=4e-3)


def test_empty_inputs_are_accepted_and_bad_ones_rejected(tf):
    """zero-size batches go through every hot entry as a no-op (status 0, nothing launched, nothing written); malformed arguments come
    back as an error status with a message, never as a launch (the reference raises from CuPy / cuDNN in both cases)."""
    import ctypes
    from tinyfusers_amd.native import hip, lib
    buf = tf.DeviceArray.zeros((4096,), np.float16, "row")
    canary = buf.numpy().copy()
    p = buf.ptr
    st = None
    hip.tf_linear_f16(p, p, p, None, None, 0, 320, 320, 0, None, 0, st)                                  # M = 0
    hip.tf_linear_bf16(p, p, p, None, None, 0, 320, 320, st)
    hip.tf_conv2d_f16(p, p, None, p, None, None, 0, None, 0, 8, 8, 64, 0, 64, 3, 3, 1, 1, 0, None, 0, st)   # N = 0
    hip.tf_conv2d_bf16(p, p, None, p, None, None, 0, None, 0, 8, 8, 64, 0, 64, 3, 3, 1, 1, 0, st)
    hip.tf_group_norm_f16(p, p, None, None, None, 0, 64, 64, 0, 8, 1e-5, 0, p, 1 << 20, st)               # N = 0
    hip.tf_group_norm_bf16(p, p, None, None, None, 0, 64, 64, 0, 8, 1e-5, 0, p, 1 << 20, st)
    hip.tf_layer_norm_f16(p, p, None, None, 0, 320, 1e-5, st)                                            # rows = 0
    hip.tf_layer_norm_bf16(p, p, None, None, 0, 320, 1e-5, st)
    hip.tf_sdpa_f16(p, p, p, p, 0, 8, 64, 64, 40, 0, 0, 40, 0, 0, 40, 0, 0, 40, 0, 0, 40, 0, st)          # B = 0
    hip.tf_silu_f16(p, p, 0, st)
    hip.tf_device_sync()
    assert np.array_equal(buf.numpy(), canary)
    for call in (lambda: lib.tf_linear_f16(p, p, p, None, None, 16, 320, 321, 0, None, 0, st),           # K not a multiple of 8
                 lambda: lib.tf_linear_bf16(None, p, p, None, None, 16, 320, 320, st),                   # null output
                 lambda: lib.tf_group_norm_f16(p, p, None, None, None, 1, 64, 60, 0, 8, 1e-5, 0, p, 1 << 20, st),   # C not divisible by G / by 8
                 lambda: lib.tf_sdpa_f16(p, p, p, p, 1, 8, 64, 64, 44, 0, 0, 44, 0, 0, 44, 0, 0, 44, 0, 0, 44, 0, st),   # head size not a multiple of 8
                 lambda: lib.tf_layer_norm_f16(p, p, p, None, 4, 320, 1e-5, st)):                        # gamma without beta
        assert call() != 0 and lib.tf_last_error()
    hip.tf_device_sync()
    assert np.array_equal(buf.numpy(), canary)
