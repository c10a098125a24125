"""GPU parity tests (pytest -m gpu) of k_igemm_pp -- the 256-row ping-pong kernel for large problems (all eight waves load and
compute; BASELINE config 5's shapes) -- forced through tf_gemm_force_config(256, BN, split) + tf_gemm_debug(512), which fails loudly
where the kernel cannot take a launch.  Exact small-integer GEMMs (any lane-map, ring or barrier slip is an O(1) error), then the conv
forms of the reference's ops (vision/conv2d.py:9-58, ff/linear.py:112-121, ff/nn.py:5-23) against the oracle: concat, stride 2,
folded up-sampling, the folded 1x1 skip projection, bias / time embedding / residual, GEGLU, split-K, GroupNorm statistics."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-2, atol=1e-2)


@pytest.fixture(scope="module")
def tf():
    import tinyfusers_amd.storage.tensor as T
    T.ensure_init(0)
    return T


def rnd(name, shape, std=1.0, seed=31):
    from tinyfusers_amd.storage.synth import synth_normal
    return synth_normal(seed, name, shape, std).astype(np.float16).astype(np.float32)


def close(got, want, **kw):
    t = dict(TOL); t.update(kw)
    got = np.asarray(got, dtype=np.float32); want = np.asarray(want, dtype=np.float32)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.isfinite(got).all(), "non-finite output"
    np.testing.assert_allclose(got, want, **t)


def dev(tf, x, layout=None, dtype=np.float16):
    return tf.DeviceArray.from_numpy(x, dtype, layout)


class forced:
    """with forced(bn, sk): every GEMM launch inside runs k_igemm_pp<bn> with split-K sk (or raises)."""

    def __init__(self, bn, sk=1, flags=512, bm=256):
        self.cfg, self.flags = (bm, bn, sk), flags

    def __enter__(self):
        from tinyfusers_amd.native import lib
        lib.tf_gemm_force_config(*self.cfg); lib.tf_gemm_debug(self.flags)

    def __exit__(self, *a):
        from tinyfusers_amd.native import lib
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)


@pytest.mark.parametrize("bn,flags", [(128, 512), (160, 512), (256, 512), (128, 512 | 8192), (160, 512 | 8192), (-128, 512), (-160, 512)])     # 8192: one phase per k-step on the 3-slot ring too; negative: the 192-row tile
@pytest.mark.parametrize("m,n,k,sk", [(1000, 400, 64, 1), (1000, 400, 256, 1), (1000, 400, 1600, 1), (257, 160, 128, 1), (2048, 328, 704, 2), (777, 1280, 1088, 3),
                                      (256, 256, 192, 1), (5000, 320, 320, 1)])
def test_pp_linear_exact_integers(tf, bn, flags, m, n, k, sk):
    """1 ... 25 K tiles (the ring's prologue, steady state and drain; 3-slot and 2-slot rings), ragged M / N edges, bias + residual,
    split-K through the shared partial path."""
    from tinyfusers_amd.native import hip
    rs = np.random.RandomState(m + n + k)
    x = rs.randint(-3, 4, (m, k)).astype(np.float32); w = rs.randint(-2, 3, (n, k)).astype(np.float32)
    b = rs.randint(-4, 5, (n,)).astype(np.float32); r = rs.randint(-8, 9, (m, n)).astype(np.float32)
    y = tf.DeviceArray.empty((m, n))
    ws = tf.DeviceArray.empty((sk * m * n * 4 + 16,), np.uint8, "row")
    xd, wd, bd, rd = dev(tf, x), dev(tf, w), dev(tf, b), dev(tf, r)     # (kept alive: the pool would hand a freed block to the next upload)
    with forced(abs(bn), sk, flags, 192 if bn < 0 else 256):
        hip.tf_linear_f16(y.ptr, xd.ptr, wd.ptr, bd.ptr, rd.ptr, m, n, k, 0, ws.ptr, ws.nbytes, None)
    np.testing.assert_array_equal(y.numpy(), (x @ w.T + b + r).astype(np.float16).astype(np.float32))


def test_pp_refuses_what_it_cannot_run(tf):
    """Channel counts off the 64 grid need the per-lane (GENERIC) gather: an explicit request for the ping-pong kernel must fail, not
    silently run something else."""
    from tinyfusers_amd.native import hip
    m, n, k = 512, 256, 200
    y = tf.DeviceArray.empty((m, n))
    x, w = dev(tf, rnd("ppr.x", (m, k))), dev(tf, rnd("ppr.w", (n, k)))
    with forced(160):
        with pytest.raises(RuntimeError):
            hip.tf_linear_f16(y.ptr, x.ptr, w.ptr, None, None, m, n, k, 0, None, 0, None)


@pytest.mark.parametrize("bn", [128, 256])
@pytest.mark.parametrize("m,c", [(1000, 64), (4608, 320)])
def test_pp_geglu(tf, bn, m, c):
    from oracle import ops as O
    from tinyfusers_amd.ff.nn import GEGLU
    x = rnd("ppg.x", (m, c)); w = rnd("ppg.w", (8 * c, c), c ** -0.5); b = rnd("ppg.b", (8 * c,), 0.1)
    g = GEGLU(c, 4 * c, init=False); g.proj.weight = dev(tf, w); g.proj.bias = dev(tf, b)
    with forced(bn):
        got = g(dev(tf, x)).numpy()
    close(got, O.geglu(x, w, b).numpy())


PP_CONV192 = [  # the 192-row tile (wave tiles of 48 rows; statistics in 96-row sub-blocks: image sizes that are multiples of 96 pixels)
    (2, 128, 64, 24, 320, 3, 1, 0, 0, 160, 1, 32), (2, 128, 64, 24, 320, 3, 1, 0, 0, 160, 2, 32), (4, 320, 0, 48, 320, 3, 1, 0, 0, 160, 1, 32),
    (3, 64, 0, 24, 128, 3, 1, 0, 0, 128, 1, 0), (2, 128, 0, 48, 128, 3, 2, 0, 0, 128, 1, 32), (2, 64, 0, 12, 64, 3, 1, 1, 0, 128, 1, 0),
    (2, 128, 0, 24, 256, 3, 1, 0, 192, 128, 1, 32), (2, 320, 0, 24, 320, 1, 1, 0, 0, 160, 1, 0),
]
PP_CONV = [  # n, c1, c2 (concat), hw, cout, k, stride, upsample, c3 (folded 1x1 skip source), bn, sk, gn
    (2, 64, 0, 32, 160, 3, 1, 0, 0, 160, 1, 32),
    (2, 128, 64, 32, 320, 3, 1, 0, 0, 160, 1, 32),     # concat input, two n-tiles, statistics (two 128-row sub-blocks per tile)
    (2, 128, 64, 32, 320, 3, 1, 0, 0, 160, 2, 32),     # ... split-K: statistics come from the reduce kernel
    (3, 64, 0, 24, 128, 3, 1, 0, 0, 128, 1, 0),        # M = 1728: the last tile is ragged, image boundaries inside tiles
    (2, 128, 0, 32, 128, 3, 2, 0, 0, 128, 1, 0),       # stride 2 (Downsample)
    (2, 64, 0, 16, 64, 3, 1, 1, 0, 128, 1, 0),         # nearest-2x upsample folded into the gather
    (2, 128, 0, 32, 256, 3, 1, 0, 192, 256, 1, 32),    # folded skip projection: extra 1x1 K tiles after the nine taps; 2-slot ring
    (2, 128, 0, 32, 256, 3, 1, 0, 64, 128, 3, 0),
    (2, 320, 0, 32, 320, 1, 1, 0, 0, 160, 1, 0),       # 1x1 conv
    (4, 320, 0, 48, 320, 3, 1, 0, 0, 160, 1, 32),      # 96 x 96 / 2: config 5's level-1 geometry at a CPU-checkable size
    (2, 128, 64, 32, 320, 3, 1, 0, 0, -160, 1, 32),    # (negative bn: one phase per k-step, tf_gemm_debug(8192))
    (2, 128, 0, 32, 128, 3, 2, 0, 0, -128, 1, 0),
    (2, 128, 0, 32, 256, 3, 1, 0, 192, -160, 2, 32),
]


@pytest.mark.parametrize("n,c1,c2,hw,cout,k,stride,ups,c3,bn,sk,gn", PP_CONV192)
def test_pp_conv2d_192_row_tile(tf, n, c1, c2, hw, cout, k, stride, ups, c3, bn, sk, gn):
    test_pp_conv2d(tf, n, c1, c2, hw, cout, k, stride, ups, c3, bn, sk, gn, bm=192)


@pytest.mark.parametrize("n,c1,c2,hw,cout,k,stride,ups,c3,bn,sk,gn", PP_CONV)
def test_pp_conv2d(tf, n, c1, c2, hw, cout, k, stride, ups, c3, bn, sk, gn, bm=256):
    from oracle import ops as O
    from tinyfusers_amd.ff.group_norm import GroupNorm
    from tinyfusers_amd.vision.conv2d import Conv2d
    xa = rnd("ppc.xa", (n, c1, hw, hw)); xb = rnd("ppc.xb", (n, c2, hw, hw)) if c2 else None
    cin = c1 + c2
    wt = rnd("ppc.w", (cout, cin, k, k), (cin * k * k) ** -0.5); b = rnd("ppc.b", (cout,), 0.1)
    m = Conv2d(cin, cout, [k, k], stride=[stride, stride], padding=[k // 2, k // 2], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    x = (dev(tf, xa), dev(tf, xb)) if c2 else dev(tf, xa)
    xin = torch.from_numpy(np.concatenate((xa, xb), 1) if c2 else xa)
    if ups:
        xin = O.upsample_nearest2x(xin)
    want = O.conv2d_bias(xin, wt, b, (k // 2, k // 2), (stride, stride))
    ho = want.shape[-1]
    kw = {}
    if c3:
        x3 = rnd("ppc.x3", (n, c3, ho, ho)); ws = rnd("ppc.ws", (cout, c3, 1, 1), c3 ** -0.5); bs = rnd("ppc.bs", (cout,), 0.1)
        proj = Conv2d(c3, cout, [1, 1], init=False); proj.weight = dev(tf, ws); proj.bias = dev(tf, bs)
        kw["extra"] = (proj, dev(tf, x3))
        want = want + O.conv2d_bias(x3, ws, bs, (0, 0))
    else:
        e = rnd("ppc.e", (n, cout), 0.5); r = rnd("ppc.r", (n, cout, ho, ho))
        kw["bias_nc"] = dev(tf, e); kw["residual"] = dev(tf, r)
        want = want + torch.from_numpy(e)[:, :, None, None] + torch.from_numpy(r)
    with forced(abs(bn), sk, 512 | (8192 if bn < 0 else 0), bm):
        y = m(x, gn=gn, upsample=bool(ups), **kw)
        got = y.numpy()
    close(got, want.numpy())
    if gn:
        assert y.gn is not None, "the statistics of the output did not ride on the conv"
        g = GroupNorm(gn, cout, init=False); g.weight = dev(tf, rnd("ppc.g", (cout,), 0.2) + 1.0, "row"); g.bias = dev(tf, rnd("ppc.gb", (cout,), 0.1), "row")
        close(g(y, silu=True).numpy(), O.silu(O.group_norm_affine(torch.from_numpy(got), gn, g.weight.numpy(), g.bias.numpy(), 1e-5)).numpy())


def test_pp_agrees_with_the_deep_ring_kernel_at_config5_size(tf):
    """A config-5 shape too large for the CPU oracle in a test (18432 x 640 x 5760: conv 3x3 640 @ 48^2, UNet batch 8): the ping-pong
    kernel against the round-1 deep-ring kernel on the same inputs (same products, another summation order), plus the linearity
    property conv(a x) = a conv(x) for a power-of-two a (exact in floating point)."""
    from tinyfusers_amd.native import lib
    from tinyfusers_amd.vision.conv2d import Conv2d
    n, c, hw = 8, 640, 48
    x = rnd("pp5.x", (n, c, hw, hw)); wt = rnd("pp5.w", (c, c, 3, 3), (c * 9) ** -0.5); b = rnd("pp5.b", (c,), 0.1)
    m = Conv2d(c, c, [3, 3], padding=[1, 1], init=False); m.weight = dev(tf, wt); m.bias = None
    xd = dev(tf, x)
    with forced(160):
        y_pp = m(xd).numpy()
        y_pp2 = m(dev(tf, 2.0 * x)).numpy()
    lib.tf_gemm_force_config(128, 128, 1); lib.tf_gemm_debug(8)
    try:
        y_ref = m(xd).numpy()
    finally:
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    assert np.isfinite(y_pp).all()
    np.testing.assert_allclose(y_pp, y_ref, atol=4e-3, rtol=4e-3)
    np.testing.assert_allclose(y_pp2, 2.0 * y_pp, rtol=0, atol=1.2e-7)      # exact up to the fp16 subnormal spacing (2^-24) of outputs below 2^-14


@pytest.mark.parametrize("m,n,k,act,bm,bn", [(4608, 960, 320, 0, 256, 160), (4608, 960, 320, 0, 192, 160), (2000, 640, 640, 0, 256, 128), (1100, 1280, 1280, 0, 192, 128),
                                             (4608, 1280, 320, 1, 256, 128), (4608, 1280, 320, 1, 256, 256), (1000, 2560, 640, 1, 192, 128), (300, 128, 64, 0, 256, 128)])
def test_pp_linear_with_folded_layer_norm(tf, m, n, k, act, bm, bn):
    """Linear(LayerNorm(x)) as one GEMM on the raw x (tf_linear_ln_f16; ff/layer_norm.py:34-49 followed by ff/linear.py:112-121 / ff/nn.py:5-12)
    on the ping-pong kernel: the row statistics come from the fragments the waves multiply (8 v_dot2 per fragment), the fold
    rstd (acc - mean colsum) happens in the shared epilogue.  Non-zero row means exercise the cancellation."""
    from oracle import ops as O
    from tinyfusers_amd.ff.layer_norm import LayerNorm
    from tinyfusers_amd.ff.linear import fold_layer_norm, linear_ln_f16
    from tinyfusers_amd.ff.nn import GEGLU
    x = rnd("pln.x", (m, k), 1.5) + 0.7
    g, b = 1 + rnd("pln.g", (k,), 0.1), rnd("pln.b", (k,), 0.1)
    ln = LayerNorm(k); ln.weight = dev(tf, g); ln.bias = dev(tf, b)
    xn = O.layer_norm(x, g, b)
    with forced(bn, 1, 512, bm):
        if act == 0:
            w, bias = rnd("pln.w", (n, k), k ** -0.5), rnd("pln.bias", (n,), 0.1)
            r = rnd("pln.r", (m, n))
            got = linear_ln_f16(dev(tf, x), fold_layer_norm(dev(tf, w), dev(tf, bias), ln), ln.eps, residual=dev(tf, r)).numpy()
            want = (O.linear(xn, w, bias) + torch.from_numpy(r)).numpy()
        else:
            w, bias = rnd("pln.w", (2 * n, k), k ** -0.5), rnd("pln.bias", (2 * n,), 0.1)
            ge = GEGLU(k, n, init=False); ge.proj.weight = dev(tf, w); ge.proj.bias = dev(tf, bias)
            got = ge(dev(tf, x), ln=ln).numpy()
            want = O.geglu(xn, w, bias).numpy()
    close(got, want)
