"""GPU parity tests (pytest -m gpu) of k_gemm_c4 -- the persistent short-K kernel (two resident 4-wave blocks per CU walk the 128 x 128
tile list, the next tile's first K tile is fetched during the epilogue, the epilogue stays in registers) -- forced through
tf_gemm_force_config(128, 128, 1) + tf_gemm_debug(1024), which fails loudly where the kernel cannot take a launch.  Exact small-integer
GEMMs first (any lane-map, ring, prefetch or tile-walk slip is an O(1) error), then the reference's ops that reach it (ff/linear.py:112-121,
ff/nn.py:5-23, the 1x1 convolutions of vision/conv2d.py:9-58 incl. the concat input) against the oracle."""
import numpy as np
import pytest
import torch

from test_gpu_pp import close, dev, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    import tinyfusers_amd.storage.tensor as T
    T.ensure_init(0)
    return T


KERNELS = {"c4": (128, 128, 1024), "c8": (256, 128, 16384)}      # k_gemm_c4 (two 4-wave blocks per CU, 128 x 128 tiles) / k_gemm_c8 (round 5: one 8-wave block, 256 x 128 tiles, three-slot ring across tiles)
KERNELS["ar"] = (128, 128, 32768)                                # k_gemm_ar (round 5): the 128-row activation panel resident in LDS, K = 256 / 320 only (tests below: test_ar_*)


@pytest.fixture(params=["c4", "c8"])      # ("ar" has tests of its own below: it takes K = 256 / 320 only)
def kern(request):
    return request.param


class forced:
    def __init__(self, kern="c4", extra=0):
        self.bm, self.bn, flag = KERNELS[kern]
        self.flags = flag | extra

    def __enter__(self):
        from tinyfusers_amd.native import lib
        lib.tf_gemm_force_config(self.bm, self.bn, 1); lib.tf_gemm_debug(self.flags)

    def __exit__(self, *a):
        from tinyfusers_amd.native import lib
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)


@pytest.mark.parametrize("flags", [0, 64])       # 64: m-fastest tile order
@pytest.mark.parametrize("m,n,k", [(1000, 400, 64), (1000, 400, 128), (1000, 400, 256), (257, 160, 192), (5000, 320, 320), (8192, 2560, 320), (70000, 128, 128), (129, 8, 1280),
                                   (66000, 136, 192), (140000, 64, 128), (73728, 320, 448), (4608, 1280, 6400)])
def test_c4_linear_exact_integers(tf, kern, flags, m, n, k):
    """1 ... 20 K tiles, 1 ... 1280 output tiles (fewer and more than the 512 resident blocks, up to five tiles per block: the cross-tile
    prefetch, the counted wait that leaves the previous tile's stores in flight), ragged M / N edges, bias + residual, then neither."""
    from tinyfusers_amd.native import hip
    rs = np.random.RandomState(m + n + k)
    x = rs.randint(-3, 4, (m, k)).astype(np.float32); w = rs.randint(-2, 3, (n, k)).astype(np.float32)
    b = rs.randint(-4, 5, (n,)).astype(np.float32); r = rs.randint(-8, 9, (m, n)).astype(np.float32)
    y = tf.DeviceArray.empty((m, n))
    xd, wd, bd, rd = dev(tf, x), dev(tf, w), dev(tf, b), dev(tf, r)
    with forced(kern, flags):
        hip.tf_linear_f16(y.ptr, xd.ptr, wd.ptr, bd.ptr, rd.ptr, m, n, k, 0, None, 0, None)
    np.testing.assert_array_equal(y.numpy(), (x @ w.T + b + r).astype(np.float16).astype(np.float32))
    y2 = tf.DeviceArray.empty((m, n))
    with forced(kern, flags):
        hip.tf_linear_f16(y2.ptr, xd.ptr, wd.ptr, None, None, m, n, k, 0, None, 0, None)
    np.testing.assert_array_equal(y2.numpy(), (x @ w.T).astype(np.float16).astype(np.float32))


def test_c4_refuses_what_it_cannot_run(tf, kern):
    from tinyfusers_amd.native import hip
    for m, n, k in ((512, 256, 200), (512, 250, 128)):     # K off the 64 grid; N off the 8 grid
        y = tf.DeviceArray.empty((m, n))
        x, w = dev(tf, rnd("c4r.x", (m, k))), dev(tf, rnd("c4r.w", (n, k)))
        with forced(kern):
            with pytest.raises(RuntimeError):
                hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, None, None, m, n, k, 0, None, 0, None)


@pytest.mark.parametrize("m,c", [(1000, 128), (4608, 320), (9216, 320)])
def test_c4_geglu(tf, kern, m, c):
    from oracle import ops as O
    from tinyfusers_amd.ff.nn import GEGLU
    x = rnd("c4g.x", (m, c)); w = rnd("c4g.w", (8 * c, c), c ** -0.5); b = rnd("c4g.b", (8 * c,), 0.1)
    g = GEGLU(c, 4 * c, init=False); g.proj.weight = dev(tf, w); g.proj.bias = dev(tf, b)
    with forced(kern):
        got = g(dev(tf, x)).numpy()
    close(got, O.geglu(x, w, b).numpy())


@pytest.mark.parametrize("m,n,k,act", [(4608, 960, 320, 0), (2000, 640, 640, 0), (1100, 1280, 1280, 0), (4608, 1280, 320, 1), (1000, 2560, 640, 1), (300, 128, 128, 0)])
def test_c4_linear_with_folded_layer_norm(tf, kern, m, n, k, act):
    """Linear(LayerNorm(x)) as one GEMM on the raw x (tf_linear_ln_f16): the row statistics come from the fragments the waves multiply,
    the fold rstd (acc - mean colsum) + bias' happens on the accumulators."""
    from oracle import ops as O
    from tinyfusers_amd.ff.layer_norm import LayerNorm
    from tinyfusers_amd.ff.linear import fold_layer_norm, linear_ln_f16
    from tinyfusers_amd.ff.nn import GEGLU
    x = rnd("cln.x", (m, k), 1.5) + 0.7
    g, b = 1 + rnd("cln.g", (k,), 0.1), rnd("cln.b", (k,), 0.1)
    ln = LayerNorm(k); ln.weight = dev(tf, g); ln.bias = dev(tf, b)
    xn = O.layer_norm(x, g, b)
    with forced(kern):
        if act == 0:
            w, bias = rnd("cln.w", (n, k), k ** -0.5), rnd("cln.bias", (n,), 0.1)
            r = rnd("cln.r", (m, n))
            got = linear_ln_f16(dev(tf, x), fold_layer_norm(dev(tf, w), dev(tf, bias), ln), ln.eps, residual=dev(tf, r)).numpy()
            want = (O.linear(xn, w, bias) + torch.from_numpy(r)).numpy()
        else:
            w, bias = rnd("cln.w", (2 * n, k), k ** -0.5), rnd("cln.bias", (2 * n,), 0.1)
            ge = GEGLU(k, n, init=False); ge.proj.weight = dev(tf, w); ge.proj.bias = dev(tf, bias)
            got = ge(dev(tf, x), ln=ln).numpy()
            want = O.geglu(xn, w, bias).numpy()
    close(got, want)


@pytest.mark.parametrize("n,c1,c2,hw,cout", [(2, 128, 64, 32, 320), (2, 320, 0, 32, 320), (3, 128, 0, 24, 128), (2, 64, 64, 32, 192)])
def test_c4_conv1x1(tf, kern, n, c1, c2, hw, cout):
    """A 1x1 / stride 1 convolution is the same GEMM; the concat input is a second K segment with its own row pitch."""
    from oracle import ops as O
    from tinyfusers_amd.vision.conv2d import Conv2d
    xa = rnd("c4c.xa", (n, c1, hw, hw)); xb = rnd("c4c.xb", (n, c2, hw, hw)) if c2 else None
    cin = c1 + c2
    wt = rnd("c4c.w", (cout, cin, 1, 1), cin ** -0.5); b = rnd("c4c.b", (cout,), 0.1); r = rnd("c4c.r", (n, cout, hw, hw))
    m = Conv2d(cin, cout, [1, 1], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    x = (dev(tf, xa), dev(tf, xb)) if c2 else dev(tf, xa)
    xin = torch.from_numpy(np.concatenate((xa, xb), 1) if c2 else xa)
    want = O.conv2d_bias(xin, wt, b, (0, 0)) + torch.from_numpy(r)
    with forced(kern):
        got = m(x, residual=dev(tf, r)).numpy()
    close(got, want.numpy())


def test_c4_agrees_with_the_deep_ring_kernel_at_config5_size(tf, kern):
    """The GEGLU projection of config 5's first level (73728 x 2560 x 320: 11520 tiles, 22-23 per resident block) is too large for the CPU
    oracle inside a test: the persistent kernel against the round-1 two-blocks-per-CU kernel on the same inputs (same products, another
    summation order inside a K tile only), plus the linearity property f(2 x) = 2 f(x) of the plain linear (exact in floating point)."""
    from tinyfusers_amd.native import hip, lib
    m, n, k = 73728, 2560, 320
    x = rnd("c45.x", (m, k), 0.5); w = rnd("c45.w", (n, k), k ** -0.5); b = rnd("c45.b", (n,), 0.1)
    xd, x2d, wd, bd = dev(tf, x), dev(tf, 2.0 * x), dev(tf, w), dev(tf, b)
    y_c4, y_c42, y_ref = tf.DeviceArray.empty((m, n // 2)), tf.DeviceArray.empty((m, n)), tf.DeviceArray.empty((m, n // 2))
    y_lin = tf.DeviceArray.empty((m, n))
    with forced(kern):
        hip.tf_linear_f16(y_c4.ptr, xd.ptr, wd.ptr, bd.ptr, None, m, n // 2, k, 1, None, 0, None)          # GEGLU
        hip.tf_linear_f16(y_lin.ptr, xd.ptr, wd.ptr, None, None, m, n, k, 0, None, 0, None)
        hip.tf_linear_f16(y_c42.ptr, x2d.ptr, wd.ptr, None, None, m, n, k, 0, None, 0, None)
    lib.tf_gemm_force_config(128, 128, 1); lib.tf_gemm_debug(16)
    try:
        hip.tf_linear_f16(y_ref.ptr, xd.ptr, wd.ptr, bd.ptr, None, m, n // 2, k, 1, None, 0, None)
    finally:
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    a, r = y_c4.numpy(), y_ref.numpy()
    assert np.isfinite(a).all()
    np.testing.assert_allclose(a, r, atol=2e-3, rtol=2e-3)
    np.testing.assert_allclose(y_c42.numpy(), 2.0 * y_lin.numpy(), rtol=0, atol=1.2e-7)


# ---- k_gemm_ar: the activation-resident kernel (K = 256 / 320).  Block runs of the tile list that start and end inside a panel, cross one or several panel seams
# (the resident images are replaced K tile by K tile behind the last step that read them), ragged M / N edges, one tile per block, fewer tiles than CUs.
@pytest.mark.parametrize("m,n,k", [(1000, 400, 256), (1000, 400, 320), (257, 160, 320), (5000, 320, 320), (8192, 2560, 320), (8192, 960, 320), (8192, 320, 256), (70000, 128, 256),
                                   (129, 8, 320), (66000, 136, 320), (73728, 320, 320), (36864, 2560, 320), (40000, 1000, 256), (128, 128, 320), (33000, 392, 320)])
def test_ar_linear_exact_integers(tf, m, n, k):
    from tinyfusers_amd.native import hip
    rs = np.random.RandomState(m + n + k)
    x = rs.randint(-3, 4, (m, k)).astype(np.float32); w = rs.randint(-2, 3, (n, k)).astype(np.float32)
    b = rs.randint(-4, 5, (n,)).astype(np.float32)
    y = tf.DeviceArray.empty((m, n))
    xd, wd, bd = dev(tf, x), dev(tf, w), dev(tf, b)
    with forced("ar"):
        hip.tf_linear_f16(y.ptr, xd.ptr, wd.ptr, bd.ptr, None, m, n, k, 0, None, 0, None)
    np.testing.assert_array_equal(y.numpy(), (x @ w.T + b).astype(np.float16).astype(np.float32))
    y2 = tf.DeviceArray.empty((m, n))
    with forced("ar"):
        hip.tf_linear_f16(y2.ptr, xd.ptr, wd.ptr, None, None, m, n, k, 0, None, 0, None)
    np.testing.assert_array_equal(y2.numpy(), (x @ w.T).astype(np.float16).astype(np.float32))


def test_ar_refuses_other_k_and_residuals(tf):
    from tinyfusers_amd.native import hip
    for m, n, k, res in ((512, 256, 128, 0), (512, 256, 640, 0), (512, 250, 320, 0), (512, 256, 320, 1)):     # K of 2 / 10 tiles; N off the 8 grid; a residual (k_gemm_c4's)
        y = tf.DeviceArray.empty((m, n))
        x, w = dev(tf, rnd("arr.x", (m, k))), dev(tf, rnd("arr.w", (n, k)))
        r = dev(tf, rnd("arr.r", (m, n)))
        with forced("ar"):
            with pytest.raises(RuntimeError):
                hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, None, r.ptr if res else None, m, n, k, 0, None, 0, None)


@pytest.mark.parametrize("m,c", [(4608, 320), (9216, 320), (1000, 256)])
def test_ar_geglu(tf, m, c):
    from oracle import ops as O
    from tinyfusers_amd.ff.nn import GEGLU
    x = rnd("arg.x", (m, c)); w = rnd("arg.w", (8 * c, c), c ** -0.5); b = rnd("arg.b", (8 * c,), 0.1)
    g = GEGLU(c, 4 * c, init=False); g.proj.weight = dev(tf, w); g.proj.bias = dev(tf, b)
    with forced("ar"):
        got = g(dev(tf, x)).numpy()
    close(got, O.geglu(x, w, b).numpy())


@pytest.mark.parametrize("m,n,k,act", [(4608, 960, 320, 0), (40000, 320, 320, 0), (4608, 1280, 320, 1), (1000, 2560, 256, 1), (300, 128, 256, 0)])
def test_ar_linear_with_folded_layer_norm(tf, m, n, k, act):
    """Linear(LayerNorm(x)) on the raw x: the row statistics come from the resident panel's fragments during the panel's first tile of a block's run (a run that
    starts inside a panel recomputes them)."""
    from oracle import ops as O
    from tinyfusers_amd.ff.layer_norm import LayerNorm
    from tinyfusers_amd.ff.linear import fold_layer_norm, linear_ln_f16
    from tinyfusers_amd.ff.nn import GEGLU
    x = rnd("aln.x", (m, k), 1.5) + 0.7
    g, b = 1 + rnd("aln.g", (k,), 0.1), rnd("aln.b", (k,), 0.1)
    ln = LayerNorm(k); ln.weight = dev(tf, g); ln.bias = dev(tf, b)
    xn = O.layer_norm(x, g, b)
    with forced("ar"):
        if act == 0:
            w, bias = rnd("aln.w", (n, k), k ** -0.5), rnd("aln.bias", (n,), 0.1)
            got = linear_ln_f16(dev(tf, x), fold_layer_norm(dev(tf, w), dev(tf, bias), ln), ln.eps).numpy()
            want = O.linear(xn, w, bias).numpy()
        else:
            w, bias = rnd("aln.w", (2 * n, k), k ** -0.5), rnd("aln.bias", (2 * n,), 0.1)
            ge = GEGLU(k, n, init=False); ge.proj.weight = dev(tf, w); ge.proj.bias = dev(tf, bias)
            got = ge(dev(tf, x), ln=ln).numpy()
            want = O.geglu(xn, w, bias).numpy()
    close(got, want)


@pytest.mark.parametrize("n,c1,c2,hw,cout", [(2, 320, 0, 32, 320), (2, 192, 128, 32, 192), (3, 256, 0, 24, 128), (2, 64, 192, 32, 320)])
def test_ar_conv1x1(tf, n, c1, c2, hw, cout):
    """A 1x1 / stride 1 convolution is the same GEMM; the concat input is a second K segment with its own row pitch (the panel's K tiles come from two sources)."""
    from oracle import ops as O
    from tinyfusers_amd.vision.conv2d import Conv2d
    xa = rnd("arc.xa", (n, c1, hw, hw)); xb = rnd("arc.xb", (n, c2, hw, hw)) if c2 else None
    cin = c1 + c2
    wt = rnd("arc.w", (cout, cin, 1, 1), cin ** -0.5); b = rnd("arc.b", (cout,), 0.1)
    m = Conv2d(cin, cout, [1, 1], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    x = (dev(tf, xa), dev(tf, xb)) if c2 else dev(tf, xa)
    xin = torch.from_numpy(np.concatenate((xa, xb), 1) if c2 else xa)
    want = O.conv2d_bias(xin, wt, b, (0, 0))
    with forced("ar"):
        got = m(x).numpy()
    close(got, want.numpy())


def test_ar_agrees_with_the_deep_ring_kernel_at_config5_size(tf):
    """73728 x 2560 x 320 (config 5's GEGLU projection: 576 panels, 45 tiles per block, two or three seams per run): against the round-1 kernel on the same inputs,
    and the linearity property f(2 x) = 2 f(x) of the plain linear (exact in floating point)."""
    from tinyfusers_amd.native import hip, lib
    m, n, k = 73728, 2560, 320
    x = rnd("a45.x", (m, k), 0.5); w = rnd("a45.w", (n, k), k ** -0.5); b = rnd("a45.b", (n,), 0.1)
    xd, x2d, wd, bd = dev(tf, x), dev(tf, 2.0 * x), dev(tf, w), dev(tf, b)
    y_ar, y_ar2, y_ref = tf.DeviceArray.empty((m, n // 2)), tf.DeviceArray.empty((m, n)), tf.DeviceArray.empty((m, n // 2))
    y_lin = tf.DeviceArray.empty((m, n))
    with forced("ar"):
        hip.tf_linear_f16(y_ar.ptr, xd.ptr, wd.ptr, bd.ptr, None, m, n // 2, k, 1, None, 0, None)          # GEGLU
        hip.tf_linear_f16(y_lin.ptr, xd.ptr, wd.ptr, None, None, m, n, k, 0, None, 0, None)
        hip.tf_linear_f16(y_ar2.ptr, x2d.ptr, wd.ptr, None, None, m, n, k, 0, None, 0, None)
    lib.tf_gemm_force_config(128, 128, 1); lib.tf_gemm_debug(16)
    try:
        hip.tf_linear_f16(y_ref.ptr, xd.ptr, wd.ptr, bd.ptr, None, m, n // 2, k, 1, None, 0, None)
    finally:
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    a, r = y_ar.numpy(), y_ref.numpy()
    assert np.isfinite(a).all()
    np.testing.assert_allclose(a, r, atol=2e-3, rtol=2e-3)
    np.testing.assert_allclose(y_ar2.numpy(), 2.0 * y_lin.numpy(), rtol=0, atol=1.2e-7)
