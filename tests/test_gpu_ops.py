"""GPU parity tests (run on the MI355X box: pytest -m gpu): every HIP op, called through the C-ABI, against
the CPU oracle on the same seeded inputs; per-op tolerance atol = rtol = 1e-2, the reference's own
(tests/conv2d.py:33, tests/group_norm.py:25-28, tests/layer_norm.py:25-28, tests/sdpa.py:100)."""
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


def rnd(name, shape, std=1.0, seed=11):
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


# ------------------------------------------------------------------------------------------------
def test_device_attributes(tf):
    import ctypes
    from tinyfusers_amd.native import hip
    v = ctypes.c_int()
    hip.tf_device_attr(ctypes.byref(v), 0, 0); assert v.value == 256
    hip.tf_device_attr(ctypes.byref(v), 2, 0); assert v.value == 64
    buf = ctypes.create_string_buffer(64)
    hip.tf_device_arch(buf, 64, 0); assert buf.value.decode().startswith("gfx950")


@pytest.mark.parametrize("n", [0, 1, 7, 8, 1000, 2 * 320 * 64 * 64 + 3])
def test_activations(tf, n):
    from oracle import ops as O
    x = rnd("act", (n,), 3.0)
    d = dev(tf, x)
    for name in ("silu", "sigmoid", "gelu", "quick_gelu", "swish"):
        got = getattr(tf.Tensor, name)(d).numpy()
        want = getattr(O, "silu" if name == "swish" else name)(x).numpy()
        close(got, want, atol=4e-3)


def test_add_cast_layout(tf):
    a, b = rnd("a", (3, 1003)), rnd("b", (3, 1003))
    close((dev(tf, a) + dev(tf, b)).numpy(), a + b)
    x = rnd("img", (2, 37, 9, 11))
    d = dev(tf, x)                       # host-side NHWC packing
    close(d.numpy(), x, atol=0, rtol=0)
    # device-side converters
    from tinyfusers_amd.native import hip
    src = tf.DeviceArray.from_numpy(x, np.float32, "row")
    dst = tf.DeviceArray.empty(x.shape, np.float16, "nhwc")
    hip.tf_nchw_f32_to_nhwc_f16(dst.ptr, src.ptr, 2, 37, 9, 11, None)
    close(dst.numpy(), x, atol=0, rtol=0)
    back = tf.DeviceArray.empty(x.shape, np.float32, "row")
    hip.tf_nhwc_f16_to_nchw_f32(back.ptr, dst.ptr, 2, 37, 9, 11, None)
    close(back.numpy(), x, atol=0, rtol=0)
    close(src.astype(np.float16).astype(np.float32).numpy(), x, atol=0, rtol=0)


# (16 * 128, {768, 1024, 1280, 1600}) are the shapes of the reference's own tests/layer_norm.py:13-27
@pytest.mark.parametrize("rows,c", [(1, 8), (5, 64), (154, 768), (2 * 4096, 320), (2 * 256, 1280), (3, 2560),
                                    (2048, 768), (2048, 1024), (2048, 1280), (2048, 1600)])
def test_layer_norm(tf, rows, c):
    from oracle import ops as O
    from tinyfusers_amd.ff.layer_norm import LayerNorm
    x = rnd("ln.x", (rows, c), 1.5) + 0.25
    m = LayerNorm(c); m.weight = dev(tf, 1 + rnd("ln.w", (c,), 0.1)); m.bias = dev(tf, rnd("ln.b", (c,), 0.1))
    close(m(dev(tf, x)).numpy(), O.layer_norm(x, m.weight.numpy(), m.bias.numpy()).numpy())


# (2048, {768, 1024, 1280, 1600}, 2, 2) with 2 groups are the shapes of the reference's own tests/group_norm.py:13-33
@pytest.mark.parametrize("n,c,h,w,g", [(2, 64, 5, 3, 32), (2, 320, 64, 64, 32), (2, 1280, 8, 8, 32), (2, 2560, 16, 16, 32),
                                       (2, 1920, 16, 16, 32), (2, 960, 32, 32, 32), (64, 768, 2, 2, 2), (3, 32, 1, 1, 32),
                                       (2048, 768, 2, 2, 2), (2048, 1024, 2, 2, 2), (2048, 1280, 2, 2, 2), (2048, 1600, 2, 2, 2)])
def test_group_norm(tf, n, c, h, w, g):
    from oracle import ops as O
    from tinyfusers_amd.ff.group_norm import GroupNorm, group_norm
    x = rnd("gn.x", (n, c, h, w), 2.0) + 0.5
    close(group_norm(dev(tf, x), g, 1e-5).numpy(), O.group_norm(x, g, 1e-5).numpy())
    m = GroupNorm(g, c); m.weight = dev(tf, 1 + rnd("gn.w", (c,), 0.1)); m.bias = dev(tf, rnd("gn.b", (c,), 0.1))
    want = O.group_norm_affine(x, g, m.weight.numpy(), m.bias.numpy())
    close(m(dev(tf, x)).numpy(), want.numpy())
    close(m(dev(tf, x), silu=True).numpy(), O.silu(want).numpy())


@pytest.mark.parametrize("c1,c2", [(1280, 1280), (1280, 640), (640, 320), (64, 8)])
def test_group_norm_concat(tf, c1, c2):
    from oracle import ops as O
    from tinyfusers_amd.ff.group_norm import GroupNorm
    a, b = rnd("gnc.a", (2, c1, 8, 8), 2.0), rnd("gnc.b", (2, c2, 8, 8), 0.5) - 1
    g = 32 if (c1 + c2) % 32 == 0 else 8
    m = GroupNorm(g, c1 + c2); m.weight = dev(tf, 1 + rnd("gn.w", (c1 + c2,), 0.1)); m.bias = dev(tf, rnd("gn.b", (c1 + c2,), 0.1))
    want = O.silu(O.group_norm_affine(np.concatenate((a, b), 1), g, m.weight.numpy(), m.bias.numpy()))
    close(m((dev(tf, a), dev(tf, b)), silu=True).numpy(), want.numpy())


LIN_SHAPES = [(1, 1280, 320), (2, 320, 1280), (8, 640, 1280), (154, 320, 768), (154, 1280, 768), (128, 1280, 1280),
              (512, 1280, 1280), (2048, 640, 640), (8192, 320, 320), (8192, 320, 1280), (100, 48, 64), (77, 6, 8), (300, 200, 72)]


@pytest.mark.parametrize("m,n,k", LIN_SHAPES)
def test_linear(tf, m, n, k):
    from oracle import ops as O
    from tinyfusers_amd.ff.linear import Linear
    x = rnd("lin.x", (m, k)); w = rnd("lin.w", (n, k), k ** -0.5); b = rnd("lin.b", (n,), 0.1); r = rnd("lin.r", (m, n))
    lin = Linear(k, n, init=False); lin.weight = dev(tf, w); lin.bias = dev(tf, b)
    close(lin(dev(tf, x)).numpy(), O.linear(x, w, b).numpy())
    if m > 8:
        close(lin(dev(tf, x), residual=dev(tf, r)).numpy(), (O.linear(x, w, b) + torch.from_numpy(r)).numpy())
    lin.bias = None
    close(lin(dev(tf, x)).numpy(), O.linear(x, w).numpy())


@pytest.mark.parametrize("bm,bn,sk", [(128, 160, 1), (64, 160, 1), (128, 128, 1), (64, 128, 1), (128, 64, 1), (64, 64, 1),
                                      (128, 160, 3), (64, 64, 4), (64, 160, 2)])
def test_linear_every_tile_config(tf, bm, bn, sk):
    """Each template instantiation + split-K, on exact small-integer data (any lane-map slip is an O(1) error)
    with ragged M / N / K edges."""
    from tinyfusers_amd.native import hip, lib
    m, n, k = 333, 276, 200
    rs = np.random.RandomState(5)
    x = rs.randint(-3, 4, (m, k)).astype(np.float32); w = rs.randint(-2, 3, (n, k)).astype(np.float32)
    b = rs.randint(-4, 5, (n,)).astype(np.float32)
    want = x @ w.T + b
    y = tf.DeviceArray.empty((m, n))
    ws = tf.DeviceArray.empty((sk * m * n * 4 + 16,), np.uint8, "row")
    lib.tf_gemm_force_config(bm, bn, sk)
    try:
        hip.tf_linear_f16(y.ptr, dev(tf, x).ptr, dev(tf, w).ptr, dev(tf, b).ptr, None, m, n, k, 0, ws.ptr, ws.nbytes, None)
    finally:
        lib.tf_gemm_force_config(0, 0, 0)
    close(y.numpy(), want, atol=0.5, rtol=1e-3)   # |values| up to ~100: fp16 output rounding only


@pytest.mark.parametrize("m,c", [(8192, 320), (77, 64), (128, 1280)])
def test_geglu_fused(tf, m, c):
    from oracle import ops as O
    from tinyfusers_amd.ff.nn import GEGLU, FeedForward
    x = rnd("gg.x", (m, c)); w = rnd("gg.w", (8 * c, c), c ** -0.5); b = rnd("gg.b", (8 * c,), 0.1)
    g = GEGLU(c, 4 * c, init=False); g.proj.weight = dev(tf, w); g.proj.bias = dev(tf, b)
    close(g(dev(tf, x)).numpy(), O.geglu(x, w, b).numpy())


CONV_CASES = [  # n, cin, h, w, cout, k, stride, pad
    (2, 8, 9, 7, 6, 3, 1, 1), (2, 8, 9, 7, 6, 3, 2, 1), (2, 8, 9, 7, 6, 1, 1, 0), (1, 16, 50, 40, 8, 2, 1, 0),
    (2, 320, 64, 64, 320, 3, 1, 1), (2, 320, 64, 64, 320, 3, 2, 1), (2, 640, 32, 32, 640, 1, 1, 0),
    (2, 1280, 8, 8, 1280, 3, 1, 1), (2, 320, 64, 64, 4, 3, 1, 1), (2, 4, 64, 64, 320, 3, 1, 1), (1, 4, 9, 7, 16, 3, 2, 1),
    (2, 1280, 16, 16, 640, 3, 1, 1),
]


@pytest.mark.parametrize("n,cin,h,w,cout,k,stride,pad", CONV_CASES)
def test_conv2d(tf, n, cin, h, w, cout, k, stride, pad):
    from oracle import ops as O
    from tinyfusers_amd.vision.conv2d import Conv2d, conv_2d
    x = rnd("conv.x", (n, cin, h, w)); wt = rnd("conv.w", (cout, cin, k, k), (cin * k * k) ** -0.5); b = rnd("conv.b", (cout,), 0.1)
    close(conv_2d(dev(tf, x), dev(tf, wt), [pad, pad], [stride, stride], [1, 1]).numpy(), O.conv_2d(x, wt, (pad, pad), (stride, stride), (1, 1)).numpy())
    m = Conv2d(cin, cout, [k, k], stride=[stride, stride], padding=[pad, pad], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    want = O.conv2d_bias(x, wt, b, (pad, pad), (stride, stride))
    close(m(dev(tf, x)).numpy(), want.numpy())
    if cin % 8 == 0:
        r = rnd("conv.r", tuple(want.shape)); e = rnd("conv.e", (n, cout))
        got = m(dev(tf, x), bias_nc=dev(tf, e), residual=dev(tf, r)).numpy()
        close(got, (want + torch.from_numpy(e)[:, :, None, None] + torch.from_numpy(r)).numpy())
        e1 = rnd("conv.e1", (1, cout))
        close(m(dev(tf, x), bias_nc=dev(tf, e1)).numpy(), (want + torch.from_numpy(e1)[:, :, None, None]).numpy())


@pytest.mark.parametrize("c1,c2,cout,hw,k", [(1280, 1280, 1280, 8, 3), (1280, 640, 1280, 16, 1), (640, 320, 320, 32, 3), (64, 8, 24, 5, 3)])
def test_conv2d_concat_and_upsample(tf, c1, c2, cout, hw, k):
    from oracle import ops as O
    from tinyfusers_amd.vision.conv2d import Conv2d
    from tinyfusers_amd.vision.unet import Upsample
    a, b = rnd("cc.a", (2, c1, hw, hw)), rnd("cc.b", (2, c2, hw, hw))
    wt = rnd("cc.w", (cout, c1 + c2, k, k), ((c1 + c2) * k * k) ** -0.5); bs = rnd("cc.bias", (cout,), 0.1)
    m = Conv2d(c1 + c2, cout, [k, k], padding=[k // 2, k // 2], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, bs)
    close(m((dev(tf, a), dev(tf, b))).numpy(), O.conv2d_bias(np.concatenate((a, b), 1), wt, bs, (k // 2, k // 2)).numpy())
    up = Upsample(c1, init=False); wu = rnd("up.w", (c1, c1, 3, 3), (c1 * 9) ** -0.5)
    up.conv.weight = dev(tf, wu); up.conv.bias = dev(tf, rnd("up.b", (c1,), 0.1))
    close(up(dev(tf, a)).numpy(), O.conv2d_bias(O.upsample_nearest2x(a), wu, up.conv.bias.numpy(), (1, 1)).numpy())


GN_FUSED_CASES = [  # n, cin, hw, cout, k, groups, forced (bm, bn, splitk) or None
    (2, 64, 16, 320, 3, 32, (64, 160, 1)), (2, 64, 16, 320, 3, 32, (64, 128, 1)), (2, 64, 16, 320, 3, 32, (64, 64, 1)),
    (2, 64, 16, 320, 3, 32, (128, 160, 1)), (2, 64, 16, 320, 3, 32, (128, 128, 1)), (2, 128, 16, 640, 3, 32, (128, 64, 1)),
    (2, 320, 16, 1280, 3, 32, (64, 160, 4)), (2, 640, 8, 1280, 3, 32, (64, 64, 8)), (2, 320, 32, 640, 1, 32, (64, 128, 2)),
    (2, 64, 8, 64, 3, 32, None), (1, 64, 16, 96, 1, 8, (64, 64, 1)), (2, 320, 16, 320, 3, 32, None), (2, 1280, 8, 1280, 3, 32, None),
    (2, 64, 8, 128, 3, 32, (128, 64, 1)),      # tile straddles two images: statistics cannot ride along (chunks = 0)
    (1, 64, 16, 2560, 1, 32, (64, 160, 1)),    # groups of 80 channels: too wide for the epilogue fold (chunks = 0)
]


@pytest.mark.parametrize("n,cin,hw,cout,k,groups,force", GN_FUSED_CASES)
def test_conv2d_emits_group_norm_statistics(tf, n, cin, hw, cout, k, groups, force):
    """conv -> GroupNorm(+SiLU) with the statistics produced by the conv's epilogue / split-K reduce (tf_conv2d_fused_f16 +
    tf_group_norm_apply_f16) against the oracle's conv followed by its own group_norm (vision/resnet.py:11,17-18), and
    bit-identical to the unfused two-kernel GroupNorm on the same conv output."""
    from oracle import ops as O
    from tinyfusers_amd.native import lib
    from tinyfusers_amd.ff.group_norm import GroupNorm
    from tinyfusers_amd.vision.conv2d import Conv2d
    x = rnd("gf.x", (n, cin, hw, hw)); wt = rnd("gf.w", (cout, cin, k, k), (cin * k * k) ** -0.5); b = rnd("gf.b", (cout,), 0.1)
    e = rnd("gf.e", (n, cout), 0.5); r = rnd("gf.r", (n, cout, hw, hw))
    gam = 1.0 + rnd("gf.g", (cout,), 0.1); bet = rnd("gf.bt", (cout,), 0.1)
    m = Conv2d(cin, cout, [k, k], padding=[k // 2, k // 2], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    g = GroupNorm(groups, cout, init=False); g.weight = dev(tf, gam, "row"); g.bias = dev(tf, bet, "row")
    if force:
        lib.tf_gemm_force_config(*force)
    try:
        y = m(dev(tf, x), bias_nc=dev(tf, e), residual=dev(tf, r), gn=groups)
    finally:
        lib.tf_gemm_force_config(0, 0, 0)
    expect_fused = cout // groups <= 64 and not (force and hw * hw % force[0] != 0)
    if force:
        assert (y.gn is not None) == expect_fused
    fused = g(y, silu=True).numpy()
    want_y = O.conv2d_bias(x, wt, b, (k // 2, k // 2)) + torch.from_numpy(e)[:, :, None, None] + torch.from_numpy(r)
    close(y.numpy(), want_y.numpy())
    yq = torch.from_numpy(y.numpy())                       # GroupNorm of the fp16 tensor the device holds
    want = O.silu(O.group_norm_affine(yq, groups, gam, bet, 1e-5)).numpy()
    close(fused, want)
    y.gn = None
    plain = g(y, silu=True).numpy()                        # stand-alone statistics pass on the same tensor
    np.testing.assert_allclose(fused, plain, atol=2e-3, rtol=2e-3)


FOLD_CASES = [  # n, hw, c_mid (= cout), c3, c4, forced (bm, bn, splitk) or None, gn
    (2, 16, 128, 64, 0, None, 0), (2, 16, 128, 64, 64, (64, 64, 1), 32), (2, 8, 1280, 1280, 1280, None, 32), (2, 16, 64, 24, 8, None, 0),
    (1, 16, 320, 640, 320, (64, 160, 2), 32), (2, 8, 128, 192, 64, (64, 64, 4), 32), (2, 32, 320, 640, 0, (128, 160, 1), 32),
]


@pytest.mark.parametrize("n,hw,cm,c3,c4,force,gn", FOLD_CASES)
def test_conv2d_with_folded_skip_projection(tf, n, hw, cm, c3, c4, force, gn):
    """ResBlock tail (vision/resnet.py:24, :31): conv3x3(h) + skip_connection(x), x possibly the concat pair of the UNet's
    output path, as ONE GEMM with the 1x1 projection riding as extra K columns (tf_conv2d_fused_f16) -- vs the oracle's two
    convs and their sum, and (gn) with the GroupNorm statistics of the sum emitted on the way."""
    from oracle import ops as O
    from tinyfusers_amd.native import lib
    from tinyfusers_amd.ff.group_norm import GroupNorm
    from tinyfusers_amd.vision.conv2d import Conv2d
    h = rnd("fs.h", (n, cm, hw, hw)); xa = rnd("fs.xa", (n, c3, hw, hw)); xb = rnd("fs.xb", (n, c4, hw, hw)) if c4 else None
    w2 = rnd("fs.w2", (cm, cm, 3, 3), (cm * 9) ** -0.5); b2 = rnd("fs.b2", (cm,), 0.1)
    ws = rnd("fs.ws", (cm, c3 + c4, 1, 1), (c3 + c4) ** -0.5); bs = rnd("fs.bs", (cm,), 0.1)
    conv = Conv2d(cm, cm, [3, 3], padding=[1, 1], init=False); conv.weight = dev(tf, w2); conv.bias = dev(tf, b2)
    proj = Conv2d(c3 + c4, cm, [1, 1], init=False); proj.weight = dev(tf, ws); proj.bias = dev(tf, bs)
    xs = (dev(tf, xa), dev(tf, xb)) if c4 else dev(tf, xa)
    if force:
        lib.tf_gemm_force_config(*force)
    try:
        y = conv(dev(tf, h), gn=gn, extra=(proj, xs))
    finally:
        lib.tf_gemm_force_config(0, 0, 0)
    xcat = np.concatenate((xa, xb), 1) if c4 else xa
    want = O.conv2d_bias(h, w2, b2, (1, 1)) + O.conv2d_bias(xcat, ws, bs, (0, 0))
    close(y.numpy(), want.numpy())
    if gn:
        assert y.gn is not None
        gam = 1.0 + rnd("fs.g", (cm,), 0.1); bet = rnd("fs.bt", (cm,), 0.1)
        g = GroupNorm(gn, cm, init=False); g.weight = dev(tf, gam, "row"); g.bias = dev(tf, bet, "row")
        close(g(y, silu=True).numpy(), O.silu(O.group_norm_affine(torch.from_numpy(y.numpy()), gn, gam, bet, 1e-5)).numpy())


SDPA_CASES = [  # b, nh, tq, tk, hs
    (2, 2, 16, 16, 8), (2, 2, 16, 5, 8), (1, 3, 100, 77, 40), (2, 8, 256, 256, 160), (2, 8, 64, 77, 160), (2, 8, 1024, 1024, 80),
    (2, 8, 4096, 77, 40), (1, 2, 4096, 4096, 40), (2, 2, 130, 130, 32), (1, 12, 77, 77, 64), (1, 1, 200, 333, 128), (1, 2, 70, 70, 96),
    (1, 2, 200, 200, 40), (1, 2, 130, 130, 80), (1, 1, 192, 192, 64), (1, 1, 256, 256, 128), (1, 1, 129, 65, 160), (1, 1, 1, 1, 40),
    (1, 2, 250, 250, 56), (1, 1, 64, 300, 48),
    (35, 12, 1024, 1024, 64),        # the dims of the reference's own tests/sdpa.py:13-20
    # d = 40 / 80 with >= 256 blocks of 256 queries: the eight-wave form of k_sdpa_dma (the last two also run the causal mask through it)
    (16, 8, 1024, 300, 40), (8, 16, 1000, 200, 40), (16, 8, 512, 512, 80), (8, 32, 300, 200, 80), (16, 16, 256, 256, 40), (32, 8, 200, 200, 80),
    (4, 8, 1000, 1000, 40), (2, 8, 2048, 200, 40), (2, 16, 1030, 64, 40),      # d = 40 with >= 256 query blocks (32-query waves), ragged Tq / Tk
]


@pytest.mark.parametrize("b,nh,tq,tk,hs", SDPA_CASES)
def test_sdpa(tf, b, nh, tq, tk, hs):
    from oracle import ops as O
    from tinyfusers_amd.attention.sdpa import scaled_dot_product_attention
    q, k, v = rnd("sd.q", (b, nh, tq, hs)), rnd("sd.k", (b, nh, tk, hs)), rnd("sd.v", (b, nh, tk, hs))
    got = scaled_dot_product_attention(dev(tf, q, "row"), dev(tf, k, "row"), dev(tf, v, "row")).numpy()
    close(got, O.scaled_dot_product_attention(q, k, v).numpy())
    if tq == tk and tq <= 256:
        mask = np.tril(np.ones((tq, tk), dtype=bool))
        got = scaled_dot_product_attention(dev(tf, q, "row"), dev(tf, k, "row"), dev(tf, v, "row"), attn_mask=mask).numpy()
        close(got, O.scaled_dot_product_attention(q, k, v, mask).numpy())


def test_sdpa_online_softmax_rescale(tf):
    """Force the running-max rescale branch: one huge score late in the key sequence (guide rule 26)."""
    from oracle import ops as O
    from tinyfusers_amd.attention.sdpa import scaled_dot_product_attention
    b, nh, t, hs = 1, 1, 256, 64
    q, k, v = rnd("rs.q", (b, nh, t, hs), 0.5), rnd("rs.k", (b, nh, t, hs), 0.5), rnd("rs.v", (b, nh, t, hs))
    k[0, 0, 200] = q[0, 0, 17] * 6.0           # query 17 . key 200 spikes in the 4th tile
    k[0, 0, 70] = q[0, 0, 33] * 4.0
    got = scaled_dot_product_attention(dev(tf, q, "row"), dev(tf, k, "row"), dev(tf, v, "row")).numpy()
    close(got, O.scaled_dot_product_attention(q, k, v).numpy())


@pytest.mark.parametrize("hs", [40, 96])
def test_sdpa_first_tile_far_below_zero(tf, hs):
    """Every score of the first key tile is hugely negative (and later tiles huge positive): the reference maximum is
    adopted from the first tile whatever its sign, then raised; nothing may underflow to an all-zero row."""
    from oracle import ops as O
    from tinyfusers_amd.attention.sdpa import scaled_dot_product_attention
    b, nh, t = 1, 2, 192
    q = rnd("ft.q", (b, nh, t, hs), 1.0)
    k = rnd("ft.k", (b, nh, t, hs), 0.05)
    v = rnd("ft.v", (b, nh, t, hs))
    k[:, :, :64] -= 3.0 * np.sign(q.mean(axis=2, keepdims=True)) * np.abs(q).mean()   # push tile 0 scores down
    q = q + 2.0 * np.sign(q.mean(axis=2, keepdims=True))                              # common direction for all queries
    k[:, :, 128:] += 1.5 * np.sign(q.mean(axis=2, keepdims=True))                     # tile 2 far above tile 0
    q, k = q.astype(np.float16).astype(np.float32), k.astype(np.float16).astype(np.float32)
    got = scaled_dot_product_attention(dev(tf, q, "row"), dev(tf, k, "row"), dev(tf, v, "row")).numpy()
    ref = O.scaled_dot_product_attention(q, k, v).numpy()
    assert np.isfinite(got).all()
    close(got, ref)


def test_sdpa_rescale_and_far_below_zero_at_256_query_blocks(tf):
    """The two online-softmax corner cases above at a size that runs the 32-queries-per-wave kernel (d = 40, 256 query blocks): a score
    spike late in the key sequence (rescale branch, the -m_run accumulator initialiser) and a first tile far below the later ones."""
    from oracle import ops as O
    from tinyfusers_amd.attention.sdpa import scaled_dot_product_attention
    b, nh, t, hs = 8, 8, 512, 40
    q, k, v = rnd("xr.q", (b, nh, t, hs), 0.7), rnd("xr.k", (b, nh, t, hs), 0.7), rnd("xr.v", (b, nh, t, hs))
    k[0, 0, 300] = q[0, 0, 17] * 8.0           # query 17 . key 300 spikes in the 5th tile
    k[3, 5, 70] = q[3, 5, 133] * 6.0
    k[5, :, :64] -= 4.0 * np.sign(q[5].mean(axis=1, keepdims=True)) * np.abs(q[5]).mean()
    q[5] = q[5] + 2.0 * np.sign(q[5].mean(axis=1, keepdims=True))
    q, k = q.astype(np.float16).astype(np.float32), k.astype(np.float16).astype(np.float32)
    got = scaled_dot_product_attention(dev(tf, q, "row"), dev(tf, k, "row"), dev(tf, v, "row")).numpy()
    assert np.isfinite(got).all()
    close(got, O.scaled_dot_product_attention(q, k, v).numpy())


RTC_SRC = r"""
// written for this test: the semantics of the reference's scale_tensor_func.cu:5-10 (in-place scalar multiply) and a row-sum that uses LDS
extern "C" __global__ void scale_kernel(float* inp, float scale, int N) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) inp[i] *= scale;
}
extern "C" __global__ void row_sum(float* out, const float* inp, int C) {
  extern __shared__ float part[];
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) s += inp[(size_t)blockIdx.x * C + c];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[blockIdx.x] = part[0];
}
"""


def test_device_load_func_compiles_and_launches_at_run_time(tf):
    """storage/device.py:31-77 + :104-127 (tests/device.py:8-22 of the reference): a kernel given as SOURCE is compiled for the device at run
    time (hiprtc), cached per Device, and launched with the reference's geometry (block 8, grid ceil(N / 8)); a kernel with dynamic
    shared memory; a compilation error is a RuntimeError that carries the compiler's message."""
    import ctypes
    from tinyfusers_amd.storage.device import Device
    d = Device("hip")
    x = rnd("rtc.x", (13, 87), 2.0)
    t = tf.Tensor.from_np(x.copy()).eval()
    fn = d.load_func(RTC_SRC, "scale_kernel")
    assert d.load_func(RTC_SRC, "scale_kernel") is fn      # cached
    n = x.size
    d.launch_func(fn, ((n + 7) // 8,), (8,), [t, ctypes.c_float(2.5), ctypes.c_int(n)])
    rs = d.load_func(RTC_SRC, "row_sum")
    out = tf.Tensor.zeros((13,), np.float32).eval()
    d.launch_func(rs, (13,), (64,), [out, t, 87], shared_mem=64 * 4)
    close(out.to("cpu").data, (2.5 * x).sum(axis=1), rtol=1e-5, atol=1e-4)
    close(t.to("cpu").data, 2.5 * x, atol=1e-6)           # (to("cpu") moves the tensor: its device memory is gone afterwards)
    with pytest.raises(RuntimeError) as e:
        d.load_func("extern \"C\" __global__ void broken(float* p) { p[0] = undeclared_symbol; }", "broken")
    assert "undeclared_symbol" in str(e.value)
    with pytest.raises(RuntimeError):
        d.load_func(RTC_SRC, "no_such_kernel")


def test_softmax_rows_and_own_runtime_kernels(tf):
    from oracle import ops as O
    from tinyfusers_amd.storage.device import Device
    d = Device("hip")
    x = rnd("sm", (13, 87), 2.0)
    t_in = tf.Tensor.from_np(x).eval(); t_out = tf.Tensor.zeros(x.shape, np.float32).eval()
    d.softmax(t_out, t_in)
    close(t_out.to("cpu").data, O.softmax_rows(x).numpy(), atol=1e-5)
    t2 = tf.Tensor.from_np(x.copy()).eval()
    d.scale_tensor(t2.dt_ptr, 2.5, 1, 13, 1, 87)
    close(t2.to("cpu").data, 2.5 * x, atol=1e-6)
    for shape, axes in (((13, 87), (1, 0)), ((30, 7, 30), (1, 2, 0)), ((7, 13, 13, 5), (1, 2, 0, 3))):
        a = rnd("tr", shape)
        src = tf.Tensor.from_np(a).eval(); dst = tf.Tensor.zeros(shape, np.float32).eval()
        (d.transpose4d if len(shape) == 4 else d.transpose)(dst, src, axes)
        out = dst.to("cpu").data.reshape([shape[i] for i in axes])
        close(out, np.transpose(a, axes), atol=0, rtol=0)
    # Tensor.T (storage/tensor.py:90-92 of the reference, driven as its tests/device.py:60-67 does): default axes (1, 0), then a 3-D permutation
    a = rnd("trT", (13, 87))
    a_t = tf.Tensor.from_np(a).eval(); out_t = tf.Tensor.zeros((13, 87), np.float32).eval()
    a_t.T(out_t)
    assert tuple(out_t.shape) == (87, 13)
    got = out_t.to("cpu").data
    assert got.shape == (87, 13)
    close(got, a.T, atol=0, rtol=0)
    a3 = rnd("trT3", (5, 7, 9))
    o3 = tf.Tensor.zeros((5, 7, 9), np.float32).eval()
    tf.Tensor.from_np(a3).eval().T(o3, axes=(2, 0, 1))
    close(o3.to("cpu").data, np.transpose(a3, (2, 0, 1)), atol=0, rtol=0)


@pytest.mark.parametrize("m,n,k,act", [(8192, 960, 320, 0), (2048, 640, 640, 0), (512, 1280, 1280, 0), (300, 128, 64, 0), (8192, 1280, 320, 1), (154, 64, 128, 1)])
def test_linear_with_folded_layer_norm(tf, m, n, k, act):
    """Linear(LayerNorm(x)) as one GEMM on the raw x (tf_linear_ln_f16) vs the oracle's LN followed by Linear / GEGLU."""
    from oracle import ops as O
    from tinyfusers_amd.ff.layer_norm import LayerNorm
    from tinyfusers_amd.ff.linear import fold_layer_norm, linear_ln_f16
    from tinyfusers_amd.ff.nn import GEGLU
    x = rnd("lnf.x", (m, k), 1.5) + 0.7          # non-zero mean: exercises the mean * colsum cancellation
    g, b = 1 + rnd("lnf.g", (k,), 0.1), rnd("lnf.b", (k,), 0.1)
    ln = LayerNorm(k); ln.weight = dev(tf, g); ln.bias = dev(tf, b)
    xn = O.layer_norm(x, g, b)
    if act == 0:
        w, bias = rnd("lnf.w", (n, k), k ** -0.5), rnd("lnf.bias", (n,), 0.1)
        r = rnd("lnf.r", (m, n))
        got = linear_ln_f16(dev(tf, x), fold_layer_norm(dev(tf, w), dev(tf, bias), ln), ln.eps, residual=dev(tf, r)).numpy()
        close(got, (O.linear(xn, w, bias) + torch.from_numpy(r)).numpy())
        got = linear_ln_f16(dev(tf, x), fold_layer_norm(dev(tf, w), None, ln), ln.eps).numpy()
        close(got, O.linear(xn, w).numpy())
    else:
        w, bias = rnd("lnf.w", (2 * n, k), k ** -0.5), rnd("lnf.bias", (2 * n,), 0.1)
        ge = GEGLU(k, n, init=False); ge.proj.weight = dev(tf, w); ge.proj.bias = dev(tf, bias)
        close(ge(dev(tf, x), ln=ln).numpy(), O.geglu(xn, w, bias).numpy())


@pytest.mark.parametrize("M,N,K,dt", [(500, 50, 8, torch.float16), (1000, 2000, 104, torch.float16), (64, 768, 768, torch.bfloat16), (37, 24, 40, torch.float32)])
def test_linear_matmul_bias_half_output(tf, M, N, K, dt):
    """ff/linear.py:66-80 `linear(X, W, B)` driven as tests/linear.py:15-60 drives it: torch tensors X (1, M, K), W (1, K, N), B (1, 1, N);
    compared with torch.matmul(X, W) + B at the reference's atol = rtol = 1e-2."""
    from tinyfusers_amd.ff.linear import linear
    g = torch.Generator().manual_seed(0)
    mat1 = torch.randn(1, M, K, generator=g).to(dt); mat2 = (torch.randn(1, K, N, generator=g) / K ** 0.5).to(dt); bias1 = torch.randn(1, 1, N, generator=g).to(dt)
    got = linear(mat1, mat2, bias1)
    assert got.shape == (M, N) and got.dtype == np.float16
    h = lambda t: t.to(torch.float16).float()              # the operands as the fp16 kernel holds them
    want = (torch.matmul(h(mat1).squeeze(0), h(mat2).squeeze(0)) + h(bias1).squeeze()).numpy()
    close(got.numpy(), want)


def test_linear_cublas_and_gemm_batch_fp32(tf):
    """The reference's fp32 cuBLAS paths (ff/linear.py:8-110) behind the same names and memory conventions, checked the way
    tests/linear.py:64-110 checks them: column-major results against numpy."""
    import ctypes
    from tinyfusers_amd.ff.linear import gemm_batch, linear_cublas
    from tinyfusers_amd.native import hip
    rng = np.random.default_rng(3)
    for m, k, n in ((4, 3, 50), (130, 257, 65), (1, 1, 1), (64, 1000, 200)):
        w_np, x_np = rng.standard_normal((m, k)).astype(np.float32), rng.standard_normal((k, n)).astype(np.float32)
        b_np = rng.standard_normal((1, n)).astype(np.float32)                                # tests/linear.py:84: a (1, N) bias
        w, x = tf.Tensor.from_np(w_np).eval(), tf.Tensor.from_np(x_np).eval()
        res = linear_cublas(w, x, None).to("cpu").data.reshape(n, m).T                    # tests/linear.py:70
        np.testing.assert_allclose(res, w_np.astype(np.float64) @ x_np.astype(np.float64), rtol=2e-5, atol=2e-5 * np.sqrt(k))
        w, x = tf.Tensor.from_np(w_np).eval(), tf.Tensor.from_np(x_np).eval()
        res = linear_cublas(w, x, tf.Tensor.from_np(b_np).eval()).to("cpu").data.reshape(n, m).T
        np.testing.assert_allclose(res, w_np.astype(np.float64) @ x_np + b_np, rtol=2e-5, atol=2e-5 * np.sqrt(k))
        from tinyfusers_amd.storage.device import Device
        r2 = tf.Tensor.from_np(np.ascontiguousarray((w_np @ x_np).T)).eval()                # column-major (m x n)
        Device("hip").add_bias(r2.dt_ptr, tf.Tensor.from_np(b_np).eval().dt_ptr, m, n)
        np.testing.assert_allclose(r2.to("cpu").data.T, w_np @ x_np + b_np, rtol=1e-6, atol=1e-6)
    B, M, K, N = 7, 4, 3, 50                                                                  # tests/linear.py:97-110 (integers: exact)
    A_np = rng.integers(0, K, size=(B, M, K)).astype(np.float32); B_np = rng.integers(0, K, size=(B, K, N)).astype(np.float32)
    c_gpu = gemm_batch(tf.Tensor.from_np(A_np), tf.Tensor.from_np(B_np))
    C_np = np.zeros((B, M, N), dtype=np.float32)
    for i in range(B):
        hip.tf_memcpy(C_np[i].ctypes.data, c_gpu[i], M * N * 4, 2)
        hip.tf_free(c_gpu[i])
    np.testing.assert_array_equal(np.transpose(C_np.reshape(B, N, M), axes=(0, 2, 1)), A_np @ B_np)
    # the raw entry with all four transpose combinations, alpha / beta, padded leading dimensions
    m, n, k = 37, 29, 41
    A, Bm, C0 = rng.standard_normal((m, k)), rng.standard_normal((k, n)), rng.standard_normal((m, n))
    for ta in (0, 1):
        for tb in (0, 1):
            lda, ldb, ldc = (k if ta else m) + 3, (n if tb else k) + 2, m + 5
            a_st = np.zeros((m if ta else k, lda), np.float32); a_st[:, : (k if ta else m)] = A if ta else A.T      # column-major storage = rows of the transpose
            b_st = np.zeros((k if tb else n, ldb), np.float32); b_st[:, : (n if tb else k)] = Bm if tb else Bm.T
            c_st = np.zeros((n, ldc), np.float32); c_st[:, :m] = C0.T
            da, db, dc = tf.Tensor.from_np(a_st).eval(), tf.Tensor.from_np(b_st).eval(), tf.Tensor.from_np(c_st).eval()
            hip.tf_sgemm_f32(ta, tb, m, n, k, 0.5, da.dt_ptr, lda, db.dt_ptr, ldb, -2.0, dc.dt_ptr, ldc, None)
            got = dc.to("cpu").data[:, :m].T
            np.testing.assert_allclose(got, 0.5 * (A.astype(np.float32) @ Bm.astype(np.float32)) - 2.0 * C0.astype(np.float32), rtol=1e-4, atol=1e-4)
            da.to("cpu"); db.to("cpu")
    with pytest.raises(RuntimeError):
        hip.tf_sgemm_f32(0, 0, 4, 4, 4, 1.0, da.dt_ptr, 2, db.dt_ptr, 4, 0.0, dc.dt_ptr, 4, None)          # lda < m


@pytest.mark.parametrize("c1,c2,hw,force", [(128, 128, 16, (64, 64, 1)), (320, 320, 16, None), (1280, 1280, 8, None), (128, 128, 8, (64, 160, 4)),
                                            (256, 128, 16, None), (1280, 640, 16, None), (640, 320, 32, (128, 160, 2)), (1280, 640, 8, (64, 160, 4))])
def test_group_norm_of_concat_from_producer_statistics(tf, c1, c2, hw, force):
    """GroupNorm(32) over concat(x, skip) of two conv outputs (vision/unet.py:72 into resnet.py:8) with the statistics merged from the
    two producers' partials (tf_group_norm_apply_cat_f16) vs the oracle's group_norm of the concat: equal splits (32 + 32 sub-groups,
    two per group) and the 2:1 splits of the output path (64 + 32 sub-groups, three per group, one group straddling the sources)."""
    from oracle import ops as O
    from tinyfusers_amd.native import lib
    from tinyfusers_amd.ff.group_norm import GroupNorm
    from tinyfusers_amd.vision.conv2d import Conv2d
    sub = c2 // 32
    ys = []
    for tag, c in (("a", c1), ("b", c2)):
        x = rnd("cs.x" + tag, (2, 64, hw, hw)); w = rnd("cs.w" + tag, (c, 64, 3, 3), (64 * 9) ** -0.5); b = rnd("cs.b" + tag, (c,), 0.3)
        m = Conv2d(64, c, [3, 3], padding=[1, 1], init=False); m.weight = dev(tf, w); m.bias = dev(tf, b)
        if force:
            lib.tf_gemm_force_config(*force)
        try:
            ys.append(m(dev(tf, x), gn=c // sub))
        finally:
            lib.tf_gemm_force_config(0, 0, 0)
        assert ys[-1].gn is not None and ys[-1].gn[2] == c // sub
    gam = 1.0 + rnd("cs.g", (c1 + c2,), 0.1); bet = rnd("cs.bt", (c1 + c2,), 0.1)
    g = GroupNorm(32, c1 + c2, init=False); g.weight = dev(tf, gam, "row"); g.bias = dev(tf, bet, "row")
    got = g((ys[0], ys[1]), silu=True).numpy()
    cat = torch.from_numpy(np.concatenate((ys[0].numpy(), ys[1].numpy()), 1))
    close(got, O.silu(O.group_norm_affine(cat, 32, gam, bet, 1e-5)).numpy())
    ys[0].gn = None
    plain = g((ys[0], ys[1]), silu=True).numpy()          # statistics pass over the concat
    np.testing.assert_allclose(got, plain, atol=2e-3, rtol=2e-3)
    with pytest.raises(RuntimeError):                      # sub-groups of different widths cannot be merged
        tf.hip.tf_group_norm_apply_cat_f16(ys[0].ptr, ys[0].ptr, ys[1].ptr, None, None, ys[1].gn[0].ptr, 1, 32, ys[1].gn[0].ptr, 1, 16, 2, hw * hw,
                                           c1, c2, 32, 1e-5, 0, None)


@pytest.mark.parametrize("vocab,dim,b,n,pos", [(10, 8, 1, 10, False), (10, 8, 1, 10, True), (49408, 768, 2, 77, True), (77, 1280, 3, 5, False)])
def test_embedding(tf, vocab, dim, b, n, pos):
    """tests/embedding.py:8-25 of the reference (vocab 10, 10 tokens, one sequence) against torch.nn.functional.embedding, plus
    the CLIP sizes and the fused position add; ids from the host or already on the device; out-of-range ids are refused."""
    from tinyfusers_amd.ff.embedding import Embedding, embedding
    w = rnd("emb.w", (vocab, dim)); pw = rnd("emb.p", (n + 2, dim))
    idx = np.random.default_rng(vocab + n).integers(0, vocab, size=(b, n))
    want = torch.nn.functional.embedding(torch.from_numpy(idx), torch.from_numpy(w.astype(np.float16).astype(np.float32))).numpy()
    if pos:
        want = (want + pw.astype(np.float16).astype(np.float32)[:n][None]).astype(np.float16).astype(np.float32)
    m = Embedding(vocab, dim, init=False); m.weight = dev(tf, w)
    got = embedding(m.weight, idx, dev(tf, pw)) if pos else m(idx)
    assert got.shape == (b, n, dim)
    close(got.numpy(), want, atol=0 if not pos else 1e-3, rtol=0 if not pos else 1e-3)
    ids_dev = tf.DeviceArray.from_numpy(idx.astype(np.int32), np.int32, "row")
    np.testing.assert_array_equal(embedding(m.weight, ids_dev, dev(tf, pw) if pos else None).numpy(), got.numpy())
    np.testing.assert_array_equal(m(idx.astype(np.float32)).numpy(), m(idx).numpy())       # the reference passes float position ids
    with pytest.raises(IndexError):
        m(np.full((1, 3), vocab))
    with pytest.raises(IndexError):
        m(np.full((1, 3), -1))


PATCH_CASES = [  # n, hw, c1, c2 (concat), cout, c3 (folded 1x1 skip source), forced (bm, bn, splitk), gn
    (2, 64, 64, 0, 160, 0, (64, 160, 1), 0),       # one image row per m-tile, a single channel group
    (2, 64, 320, 0, 320, 0, (64, 160, 1), 32),     # conv 3x3 320 @ 64^2 (the step's most frequent shape) + statistics
    (2, 64, 320, 0, 320, 0, (128, 160, 2), 32),    # two image rows per tile, split-K cuts inside a channel group
    (1, 32, 128, 64, 256, 0, (64, 128, 1), 0),     # concat input, two image rows per tile
    (2, 32, 640, 0, 640, 0, (128, 160, 4), 32),    # 6-row patches, splits of 22/23 K tiles
    (2, 16, 256, 128, 128, 0, (128, 128, 3), 0),   # 8 image rows per tile, odd split
    (2, 8, 128, 0, 128, 0, (64, 128, 1), 32),      # a whole 8x8 image per tile
    (2, 16, 128, 0, 128, 192, (64, 128, 1), 32),   # folded skip projection: extra 1x1 K tiles after the patches
    (1, 64, 128, 0, 160, 64, (64, 160, 2), 0),
]


@pytest.mark.parametrize("n,hw,c1,c2,cout,c3,force,gn", PATCH_CASES)
def test_conv2d_patch_variant(tf, n, hw, c1, c2, cout, c3, force, gn):
    """k_igemm_patch (3x3 / stride 1 / pad 1: the activation patch of a channel group staged once for its nine taps) forced through
    tf_gemm_debug(128), against the oracle and against the tap-by-tap kernel on the same inputs."""
    from oracle import ops as O
    from tinyfusers_amd.native import lib
    from tinyfusers_amd.vision.conv2d import Conv2d
    xa = rnd("pv.xa", (n, c1, hw, hw)); xb = rnd("pv.xb", (n, c2, hw, hw)) if c2 else None
    cin = c1 + c2
    wt = rnd("pv.w", (cout, cin, 3, 3), (cin * 9) ** -0.5); b = rnd("pv.b", (cout,), 0.1)
    e = rnd("pv.e", (n, cout), 0.5)
    m = Conv2d(cin, cout, [3, 3], padding=[1, 1], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    x = (dev(tf, xa), dev(tf, xb)) if c2 else dev(tf, xa)
    kw = {}
    want = O.conv2d_bias(np.concatenate((xa, xb), 1) if c2 else xa, wt, b, (1, 1))
    if c3:
        x3 = rnd("pv.x3", (n, c3, hw, hw)); ws = rnd("pv.ws", (cout, c3, 1, 1), c3 ** -0.5); bs = rnd("pv.bs", (cout,), 0.1)
        proj = Conv2d(c3, cout, [1, 1], init=False); proj.weight = dev(tf, ws); proj.bias = dev(tf, bs)
        kw["extra"] = (proj, dev(tf, x3))
        want = want + O.conv2d_bias(x3, ws, bs, (0, 0))
    else:
        kw["bias_nc"] = dev(tf, e)
        want = want + torch.from_numpy(e)[:, :, None, None]
    got = {}
    for flags in (128, 8):                                  # patch variant, then the deep-ring tap-by-tap kernel
        lib.tf_gemm_force_config(*force); lib.tf_gemm_debug(flags)
        try:
            y = m(x, gn=gn, **kw)
            got[flags] = (y.numpy(), None if y.gn is None else (y.gn[0].numpy().copy(), y.gn[1]))
        finally:
            lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    close(got[128][0], want.numpy())
    np.testing.assert_allclose(got[128][0], got[8][0], atol=4e-3, rtol=4e-3)      # same products, another summation order
    if gn:
        assert got[128][1] is not None and got[128][1][1] == got[8][1][1]


ALL8_CASES = [  # kind, dims, forced (bm, bn, splitk)
    ("conv", (2, 320, 64, 320, 3), (64, 160, 1)), ("conv", (2, 320, 64, 320, 3), (128, 160, 2)), ("conv", (2, 1280, 16, 1280, 3), (128, 160, 8)),
    ("conv", (2, 1280, 8, 1280, 3), (64, 160, 16)), ("conv", (2, 640, 32, 640, 1), (64, 128, 1)), ("conv", (1, 128, 16, 64, 3), (128, 64, 1)),
    ("lin", (8192, 320, 1280), (128, 128, 1)), ("lin", (512, 1280, 1280), (64, 64, 4)), ("lin", (154, 320, 768), (64, 64, 1)),
]


@pytest.mark.parametrize("kind,dims,force", ALL8_CASES)
def test_gemm_all8_variant_is_bit_identical(tf, kind, dims, force):
    """tf_gemm_debug(256): the consumer waves issue part of the weight loads of every K tile.  Same tiles, same K order, same
    accumulation as the deep-ring kernel, so the result must be bit-identical to it (and correct against the oracle)."""
    from oracle import ops as O
    from tinyfusers_amd.native import lib
    from tinyfusers_amd.ff.linear import Linear
    from tinyfusers_amd.vision.conv2d import Conv2d
    if kind == "conv":
        n, cin, hw, cout, k = dims
        x = rnd("a8.x", (n, cin, hw, hw)); wt = rnd("a8.w", (cout, cin, k, k), (cin * k * k) ** -0.5); b = rnd("a8.b", (cout,), 0.1)
        m = Conv2d(cin, cout, [k, k], padding=[k // 2, k // 2], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
        xd = dev(tf, x)
        call = lambda: m(xd)
        want = O.conv2d_bias(x, wt, b, (k // 2, k // 2)).numpy()
    else:
        mm, nn, kk = dims
        x = rnd("a8.x", (mm, kk)); wt = rnd("a8.w", (nn, kk), kk ** -0.5); b = rnd("a8.b", (nn,), 0.1)
        m = Linear(kk, nn, init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
        xd = dev(tf, x)
        call = lambda: m(xd)
        want = O.linear(x, wt, b).numpy()
    got = {}
    for flags in (256, 8):
        lib.tf_gemm_force_config(*force); lib.tf_gemm_debug(flags)
        try:
            got[flags] = call().numpy()
        finally:
            lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    close(got[256], want)
    np.testing.assert_array_equal(got[256], got[8])


def test_comm_single_rank_broadcast(tf):
    """tf_comm_unique_id / tf_comm_init_rank / tf_bcast / tf_comm_destroy (RCCL) with a world of one rank: the
    weight-arena broadcast a non-Python host would issue; in place, stream-ordered, the payload comes back unchanged."""
    import ctypes
    hip = tf.hip
    uid = (ctypes.c_char * 128)()
    hip.tf_comm_unique_id(uid)
    assert any(bytes(uid)), "empty unique id"
    comm = ctypes.c_void_p()
    hip.tf_comm_init_rank(ctypes.byref(comm), uid, 1, 0)
    try:
        x = np.arange(1 << 20, dtype=np.float32)
        d = tf.DeviceArray.from_numpy(x, np.float32, "row")
        hip.tf_bcast(comm, d.ptr, d.nbytes, 0, None)
        hip.tf_stream_sync(None)
        np.testing.assert_array_equal(d.numpy(), x)
        with pytest.raises(RuntimeError):
            hip.tf_bcast(comm, d.ptr, d.nbytes, 1, None)        # root outside the world
    finally:
        hip.tf_comm_destroy(comm)


# ---- BASELINE config 1: the reference's own single-op tests, at their own shapes ----------------------------------------------
# tests/conv2d.py:13-33 (test_conv_fp16): X (1, 2, 10000, 10000), W (1, 2, 2, 2), pad 0, stride 1 vs F.conv2d, atol = rtol = 1e-2;
# tests/conv2d.py:35-58 (test_conv_vs_tinygrad): the Conv2d module with an all-ones (1, 2, 2, 2) kernel on U[0, 1) input.
# Cin = 2 runs through the small-C path (im2col bands + the MFMA GEMM); the full 10000 x 10000 case is the reference's exact size.
@pytest.mark.parametrize("hw", [(33, 47), (1000, 1000), (10000, 10000)])
def test_conv2d_reference_test_family(tf, hw):
    from oracle import ops as O
    from tinyfusers_amd.storage.synth import synth_normal
    from tinyfusers_amd.vision.conv2d import Conv2d, conv_2d
    h, w = hw
    x = synth_normal(3, "cfg1.x", (1, 2, h, w)).astype(np.float16)
    wt = rnd("cfg1.w", (1, 2, 2, 2))
    got = conv_2d(dev(tf, x), dev(tf, wt), [0, 0], [1, 1], [1, 1]).numpy()
    want = O.conv_2d(x.astype(np.float32), wt, (0, 0), (1, 1), (1, 1)).numpy()
    close(got, want)
    del got, want
    # module form, ones kernel, non-negative input (tinygrad's Tensor.rand), bias present (the reference's Conv2d always has one)
    xu = np.abs(x) % 1.0
    m = Conv2d(2, 1, [2, 2], init=False)
    m.weight = dev(tf, np.ones((1, 2, 2, 2), np.float32)); m.bias = dev(tf, np.zeros((1,), np.float32))
    close(m(dev(tf, xu)).numpy(), O.conv_2d(xu.astype(np.float32), np.ones((1, 2, 2, 2), np.float32), (0, 0), (1, 1), (1, 1)).numpy())


# tests/layer_norm.py:22-41 (test_layernorm): x (2048, C, 10, 10), scale / bias (1, C, 10, 10), normalised over [C, H, W], eps 1e-3
@pytest.mark.parametrize("n,c", [(4, 16), (64, 768), (2048, 768), (2048, 1600)])
def test_layer_norm_reference_slab(tf, n, c):
    from oracle import ops as O
    from tinyfusers_amd.ff.layer_norm import layer_norm
    x = rnd("lnr.x", (n, c, 10, 10), 1.5) + 0.25
    sc, bs = rnd("lnr.s", (1, c, 10, 10)), rnd("lnr.b", (1, c, 10, 10))
    got = layer_norm(dev(tf, x), dev(tf, sc), dev(tf, bs), np.full((1, 1, 1, 1), 1e-3, np.float32)).numpy()
    close(got, O.layer_norm(x, sc, bs, 1e-3).numpy())


# tests/layer_norm.py:44-71 (test_layernorm_tinygrad): x (10, 32, 10, 10), LayerNorm over the last dim W = 10 (not a multiple of 8), eps 1e-3
@pytest.mark.parametrize("shape", [(10, 32, 10, 10), (3, 5, 7, 13), (2, 77, 4100)])
def test_layer_norm_last_dim_any_width(tf, shape):
    from oracle import ops as O
    from tinyfusers_amd.ff.layer_norm import LayerNorm
    wd = shape[-1]
    x = rnd("lnt.x", shape, 1.5) + 0.25
    m = LayerNorm(wd, eps=1e-3)
    m.weight = dev(tf, rnd("lnt.s", (wd,)), "row"); m.bias = dev(tf, rnd("lnt.b", (wd,)), "row")
    got = m(dev(tf, x, "row")).numpy()
    close(got, O.layer_norm(x, m.weight.numpy(), m.bias.numpy(), 1e-3).numpy())


# ---- the mask forms of attention/sdpa.py:67-68 (boolean: attend where True; anything else: additive) and head sizes beyond the
# flash kernel's, through the unfused matmul / softmax / matmul path
@pytest.mark.parametrize("b,nh,tq,tk,hs,kind", [(2, 3, 40, 77, 40, "bool"), (2, 3, 40, 77, 40, "additive"), (1, 2, 64, 64, 64, "bool_bh"),
                                                (1, 12, 77, 77, 64, "padding"), (1, 1, 256, 256, 512, "none"), (2, 1, 100, 60, 200, "additive")])
def test_sdpa_masks_and_large_heads(tf, b, nh, tq, tk, hs, kind):
    from oracle import ops as O
    from tinyfusers_amd.attention.sdpa import scaled_dot_product_attention
    q, k, v = rnd("m.q", (b, nh, tq, hs)), rnd("m.k", (b, nh, tk, hs)), rnd("m.v", (b, nh, tk, hs))
    rng = np.random.default_rng(5)
    if kind == "bool":
        mask = rng.random((tq, tk)) < 0.6; mask[:, 0] = True
    elif kind == "bool_bh":
        mask = rng.random((b, nh, tq, tk)) < 0.5; mask[..., 3] = True
    elif kind == "padding":
        mask = np.ones((1, 1, tq, tk), dtype=bool); mask[..., 50:] = False           # key-padding mask
    elif kind == "additive":
        mask = rng.standard_normal((tq, tk)).astype(np.float32) * 2.0
    else:
        mask = None
    got = scaled_dot_product_attention(dev(tf, q, "row"), dev(tf, k, "row"), dev(tf, v, "row"), mask).numpy()
    close(got, O.scaled_dot_product_attention(q, k, v, mask).numpy())


@pytest.mark.parametrize("b,nh,tq,tk,hs", [(1, 1, 128, 256, 512), (1, 2, 96, 77, 256)])
def test_sdpa_unfused_large_logits_keep_fp32_scores(tf, b, nh, tq, tk, hs):
    """Scaled logits in the hundreds (what a d = 512 head sees on real weights): the scores must reach the softmax in fp32, as the
    reference's do (attention/sdpa.py:63-66).  Stored as fp16 they would carry errors of 0.1 ... 0.25 at this size, i.e. 10 - 25 %
    on every probability; rows are built so that several keys stay within a few units of the maximum (a one-hot row hides the error)."""
    from oracle import ops as O
    from tinyfusers_amd.attention.sdpa import scaled_dot_product_attention
    q, k, v = rnd("big.q", (b, nh, tq, hs), 4.0), rnd("big.k", (b, nh, tk, hs), 4.0), rnd("big.v", (b, nh, tk, hs))
    base = rnd("big.dir", (hs,), 4.0)
    k = (k * 0.08 + base[None, None, None, :]).astype(np.float16).astype(np.float32)     # keys share a large common direction
    q = (q * 0.25 + base[None, None, None, :]).astype(np.float16).astype(np.float32)
    s = np.einsum("bhqd,bhkd->bhqk", q, k) / np.sqrt(hs)
    assert np.abs(s).max() > 250 and np.median(s.max(-1) - np.sort(s, -1)[..., -4]) < 8.0, (np.abs(s).max(), np.median(s.max(-1) - np.sort(s, -1)[..., -4]))
    got = scaled_dot_product_attention(dev(tf, q, "row"), dev(tf, k, "row"), dev(tf, v, "row")).numpy()
    close(got, O.scaled_dot_product_attention(q, k, v).numpy())


def test_clip_attention_general_mask(tf):
    """attention/attention.py:88-104 with a mask that is not the causal one (a key-padding mask on top of it, additive form):
    the reference hands any mask to sdpa.py:67-68; here it runs the unfused path on gathered heads."""
    import oracle
    from oracle import clip as OC
    from tinyfusers_amd.attention.attention import CLIPAttention
    from tinyfusers_amd.storage.state import update_state
    from tinyfusers_amd.storage.synth import synth_tensor
    t = 77
    names = {f"{n}.{w}": ((768, 768) if w == "weight" else (768,)) for n in ("q_proj", "k_proj", "v_proj", "out_proj") for w in ("weight", "bias")}
    W = {k_: synth_tensor(3, "clipattn." + k_, s_) for k_, s_ in names.items()}
    m = CLIPAttention(init=False)
    update_state(m, W, "")
    x = rnd("clipattn.x", (2, t, 768))
    mask = OC.causal_mask(t).numpy().copy()
    mask[..., 40:] = -np.inf                                       # keys 40 .. 76 are padding
    mask[..., np.arange(t), np.arange(t)] = np.where(np.arange(t) >= 40, 0.0, mask[..., np.arange(t), np.arange(t)])   # a padded query still sees itself
    got = m(dev(tf, x, "row"), mask).numpy()
    Wt = {"p." + k_: torch.from_numpy(v_.astype(np.float32)) for k_, v_ in W.items()}
    want = OC.clip_attention(torch.from_numpy(x), Wt, "p.", torch.from_numpy(mask), 12).numpy()
    close(got, want, atol=2e-2)
    # ... and the causal mask itself still takes the fused kernel and agrees with the same oracle
    cm = OC.causal_mask(t).numpy()
    close(m(dev(tf, x, "row"), cm).numpy(), OC.clip_attention(torch.from_numpy(x), Wt, "p.", torch.from_numpy(cm), 12).numpy(), atol=2e-2)


@pytest.mark.parametrize("c,hw", [(64, 8), (512, 16), (512, 64)])
def test_attn_block_intended_single_head(tf, c, hw):
    """attention/attention.py:10-24 in the LDM form (one head of size c over the h*w pixels): what real SD weights need."""
    import oracle
    from oracle import vae as OV
    from tinyfusers_amd import config
    from tinyfusers_amd.attention.attention import AttnBlock
    from tinyfusers_amd.storage.state import update_state
    from tinyfusers_amd.storage.synth import synth_tensor
    names = {f"{n}.{t}": ((c, c, 1, 1) if t == "weight" else (c,)) for n in ("q", "k", "v", "proj_out") for t in ("weight", "bias")}
    names.update({"norm.weight": (c,), "norm.bias": (c,)})
    W = {k_: synth_tensor(9, "vae.mid.attn_1." + k_, s_) for k_, s_ in names.items()}
    m = AttnBlock(c, init=False)
    update_state(m, W, "")
    x = rnd("ab.x", (1, c, hw, hw))
    old = config.head_merge
    config.head_merge = "intended"
    try:
        got = m(dev(tf, x)).numpy()
    finally:
        config.head_merge = old
    Wt = {"p." + k_: torch.from_numpy(v_.astype(np.float32)) for k_, v_ in W.items()}
    want = OV.attn_block(torch.from_numpy(x), Wt, "p", head_merge="intended").numpy()
    close(got, want, atol=2e-2)


# ---- GroupNorm (+ SiLU) of the input applied INSIDE the conv launch (tf_conv2d_gn_f16): GroupNorm -> SiLU -> conv3x3 of
# vision/resnet.py:13-17, :22-27 (patch kernel: the loader waves normalise each patch piece once, padding stays zero) and
# GroupNorm -> 1x1 conv of attention/attention.py:66-68 (tap-by-tap kernel), statistics from the producing convs' partials.
GI_CASES = [  # n, c1, c2 (concat partner or 0), hw, cout, k, c3 (extra 1x1 source: folded skip projection), silu, forced (bm, bn, splitk) or None
    (2, 128, 0, 8, 128, 3, 0, True, None), (2, 128, 0, 16, 128, 3, 0, True, (64, 128, 1)), (2, 320, 0, 64, 320, 3, 0, True, (64, 160, 1)),
    (2, 320, 0, 64, 320, 3, 0, True, (128, 160, 2)), (2, 640, 0, 32, 640, 3, 0, True, (128, 128, 2)), (2, 1280, 0, 16, 1280, 3, 0, True, (128, 160, 8)),
    (2, 1280, 0, 8, 1280, 3, 0, True, (64, 160, 16)), (2, 1280, 1280, 8, 1280, 3, 0, True, (64, 160, 16)), (2, 1280, 640, 16, 1280, 3, 0, True, None),
    (2, 640, 320, 32, 640, 3, 0, True, (128, 160, 4)), (2, 640, 0, 32, 640, 3, 320, True, (64, 160, 2)), (1, 128, 0, 16, 128, 3, 64, True, (64, 128, 1)),
    (2, 320, 0, 64, 320, 1, 0, False, (64, 64, 1)), (2, 320, 0, 64, 320, 1, 0, False, (64, 160, 1)), (2, 640, 0, 32, 640, 1, 0, False, (64, 128, 1)),
    (2, 1280, 0, 16, 1280, 1, 0, False, (128, 64, 1)), (2, 1280, 0, 8, 1280, 1, 0, False, None), (2, 128, 0, 16, 64, 1, 0, True, (128, 64, 1)),
    (3, 128, 0, 8, 128, 3, 0, False, (64, 128, 1)),
]


@pytest.mark.parametrize("n,c1,c2,hw,cout,k,c3,silu,force", GI_CASES)
def test_conv2d_with_input_group_norm_inside(tf, n, c1, c2, hw, cout, k, c3, silu, force):
    from oracle import ops as O
    from tinyfusers_amd import config
    from tinyfusers_amd.native import lib
    from tinyfusers_amd.ff.group_norm import GroupNorm
    from tinyfusers_amd.vision.conv2d import Conv2d
    # producers: two convs whose epilogues emit the statistics of x (and x2), as inside the UNet
    srcs = []
    sub = (c2 // 32) if c2 else 0
    for tag, c in (("a", c1), ("b", c2)):
        if not c:
            continue
        x0 = rnd("gi.x" + tag, (n, 64, hw, hw)); w0 = rnd("gi.w" + tag, (c, 64, 3, 3), (64 * 9) ** -0.5); b0 = rnd("gi.b" + tag, (c,), 0.5)
        m0 = Conv2d(64, c, [3, 3], padding=[1, 1], init=False); m0.weight = dev(tf, w0); m0.bias = dev(tf, b0)
        lib.tf_gemm_force_config(64, 64, 1)                # (a tile the statistics can ride on whatever the tuner would pick: m-tiles inside one image)
        try:
            srcs.append(m0(dev(tf, x0), gn=(c // sub) if c2 else 32))
        finally:
            lib.tf_gemm_force_config(0, 0, 0)
        assert srcs[-1].gn is not None
    C = c1 + c2
    gam = 1.0 + rnd("gi.g", (C,), 0.2); bet = rnd("gi.bt", (C,), 0.2)
    g = GroupNorm(32, C, init=False); g.weight = dev(tf, gam, "row"); g.bias = dev(tf, bet, "row")
    wt = rnd("gi.w", (cout, C, k, k), (C * k * k) ** -0.5); b = rnd("gi.b", (cout,), 0.1)
    m = Conv2d(C, cout, [k, k], padding=[k // 2, k // 2], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    e = rnd("gi.e", (n, cout), 0.5)
    kw = dict(bias_nc=dev(tf, e), gn=32)
    xe = None
    if c3:
        xe = rnd("gi.x3", (n, c3, hw, hw)); ws = rnd("gi.ws", (cout, c3, 1, 1), c3 ** -0.5); bs = rnd("gi.bs", (cout,), 0.1)
        proj = Conv2d(c3, cout, [1, 1], init=False); proj.weight = dev(tf, ws); proj.bias = dev(tf, bs)
        kw["extra"] = (proj, dev(tf, xe))
    xin = (srcs[0], srcs[1]) if c2 else srcs[0]
    assert lib.tf_conv2d_gn_supported(n, hw, hw, c1, c2, cout, k, k, 1, k // 2, 0, c3, 0, 32) == 1
    if force:
        lib.tf_gemm_force_config(*force)
    config.fuse_group_norm_3x3 = True
    try:
        y = m(xin, gn_in=(g, silu), **kw)                  # ONE launch: normalise + conv (+ statistics of y)
    finally:
        lib.tf_gemm_force_config(0, 0, 0)
        config.fuse_group_norm_3x3 = False
    config.fuse_group_norm = False
    try:
        y_ref = m(xin, gn_in=(g, silu), **kw)              # GroupNorm launch + conv launch on the same inputs
    finally:
        config.fuse_group_norm = True
    cat = torch.from_numpy(np.concatenate([s_.numpy() for s_ in srcs], 1))
    hn = O.group_norm_affine(cat, 32, gam, bet, 1e-5)
    hn = O.silu(hn) if silu else hn
    hn = hn.to(torch.float16).to(torch.float32)            # what the conv reads is the fp16-rounded normalised tensor
    want = O.conv2d_bias(hn, wt, b, (k // 2, k // 2)) + torch.from_numpy(e)[:, :, None, None]
    if c3:
        want = want + O.conv2d_bias(xe, ws, bs, (0, 0))
    close(y.numpy(), want.numpy())
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), atol=4e-3, rtol=4e-3)
    assert y.gn is not None or force is None or hw * hw % force[0] != 0 or cout // 32 < 4
    if y.gn is not None:                                   # the statistics emitted on the way still describe y
        g2 = GroupNorm(32, cout, init=False); g2.weight = dev(tf, 1.0 + rnd("gi.g2", (cout,), 0.1), "row"); g2.bias = dev(tf, rnd("gi.b2", (cout,), 0.1), "row")
        close(g2(y, silu=True).numpy(), O.silu(O.group_norm_affine(torch.from_numpy(y.numpy()), 32, g2.weight.numpy(), g2.bias.numpy(), 1e-5)).numpy())


# ---- conv -> GroupNorm (-> SiLU) behind a split-K shape: the reduce kernel owns whole (image, group) slabs, finishes the statistics and
# writes the normalised tensor next to y (tf_conv2d_fused_norm_f16 / k_splitk_reduce_gn_apply)
RGA_CASES = [  # n, cin, hw, cout, k, forced (bm, bn, splitk), silu, expect z
    (2, 1280, 8, 1280, 3, (64, 160, 16), True, True), (2, 1280, 16, 1280, 3, (128, 160, 8), True, True), (2, 640, 32, 640, 3, (128, 160, 4), True, True),
    (2, 320, 32, 640, 3, (64, 128, 2), False, True), (2, 640, 32, 320, 3, (64, 160, 4), True, True), (1, 128, 16, 128, 3, (64, 64, 2), True, True),
    (3, 640, 16, 1280, 1, (64, 160, 2), False, True),
    (2, 640, 64, 320, 3, (128, 160, 2), True, False),      # 4096 rows of 20-channel slabs do not fit a block's registers: reduce + apply launches
    (5, 128, 24, 320, 3, (64, 160, 2), True, True),        # 5 images x 576 rows
    (2, 320, 16, 320, 3, (64, 160, 1), True, False),       # unsplit: the statistics come from the GEMM epilogue, the apply stays a launch
]


@pytest.mark.parametrize("n,cin,hw,cout,k,force,silu,expect", RGA_CASES)
def test_conv2d_split_k_reduce_applies_group_norm(tf, n, cin, hw, cout, k, force, silu, expect):
    from oracle import ops as O
    from tinyfusers_amd.native import lib
    from tinyfusers_amd.ff.group_norm import GroupNorm
    from tinyfusers_amd.vision.conv2d import Conv2d
    x = rnd("rg.x", (n, cin, hw, hw)); wt = rnd("rg.w", (cout, cin, k, k), (cin * k * k) ** -0.5); b = rnd("rg.b", (cout,), 0.1)
    e = rnd("rg.e", (n, cout), 0.5); r = rnd("rg.r", (n, cout, hw, hw))
    gam = 1.0 + rnd("rg.g", (cout,), 0.1); bet = rnd("rg.bt", (cout,), 0.1)
    m = Conv2d(cin, cout, [k, k], padding=[k // 2, k // 2], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    g = GroupNorm(32, cout, init=False); g.weight = dev(tf, gam, "row"); g.bias = dev(tf, bet, "row")
    from tinyfusers_amd import config
    lib.tf_gemm_force_config(*force)
    try:
        y = m(dev(tf, x), bias_nc=dev(tf, e), residual=dev(tf, r), gn=32, out_norm=(g, silu))
    finally:
        lib.tf_gemm_force_config(0, 0, 0)
    assert (y.normed is not None) == expect
    want_y = O.conv2d_bias(x, wt, b, (k // 2, k // 2)) + torch.from_numpy(e)[:, :, None, None] + torch.from_numpy(r)
    close(y.numpy(), want_y.numpy())
    z = g(y, silu=silu)                                    # returns y.normed's tensor without a launch when the reduce wrote it
    if expect:
        assert z is y.normed[2] and y.gn is not None and y.gn[1] == 1
    yq = torch.from_numpy(y.numpy())
    want = O.group_norm_affine(yq, 32, gam, bet, 1e-5)
    want = O.silu(want) if silu else want
    close(z.numpy(), want.numpy())
    # the one-chunk statistics left behind serve any other reader of y (e.g. a later concat), and agree with a stand-alone pass
    y.normed = None
    again = g(y, silu=silu).numpy()
    np.testing.assert_allclose(again, z.numpy(), atol=2e-3, rtol=2e-3)
    y.gn = None
    plain = g(y, silu=silu).numpy()
    np.testing.assert_allclose(plain, z.numpy(), atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("sk", [1, 2, 3])
def test_256_row_tile_linear_and_conv(tf, sk):
    """k_igemm<256, 128> (round 2: the tile for problems large enough to give every CU a 256 x 128 tile; fragments pipelined per 32-deep
    k-step): exact small-integer linear with ragged M / N edges, and a 3x3 conv with bias, time embedding, residual and the GroupNorm
    statistics of the output, vs the oracle."""
    from oracle import ops as O
    from tinyfusers_amd.native import hip, lib
    from tinyfusers_amd.ff.group_norm import GroupNorm
    from tinyfusers_amd.vision.conv2d import Conv2d
    m, n, k = 777, 276, 448
    rs = np.random.RandomState(6)
    x = rs.randint(-3, 4, (m, k)).astype(np.float32); w = rs.randint(-2, 3, (n, k)).astype(np.float32); b = rs.randint(-4, 5, (n,)).astype(np.float32)
    y = tf.DeviceArray.empty((m, n))
    ws = tf.DeviceArray.empty((sk * m * n * 4 + 16,), np.uint8, "row")
    lib.tf_gemm_force_config(256, 128, sk)
    try:
        hip.tf_linear_f16(y.ptr, dev(tf, x).ptr, dev(tf, w).ptr, dev(tf, b).ptr, None, m, n, k, 0, ws.ptr, ws.nbytes, None)
        close(y.numpy(), x @ w.T + b, atol=0.5, rtol=1e-3)
        nimg, cin, hw, cout = 2, 128, 32, 256
        xc = rnd("t256.x", (nimg, cin, hw, hw)); wt = rnd("t256.w", (cout, cin, 3, 3), (cin * 9) ** -0.5); bc = rnd("t256.b", (cout,), 0.1)
        e = rnd("t256.e", (nimg, cout), 0.5); r = rnd("t256.r", (nimg, cout, hw, hw))
        conv = Conv2d(cin, cout, [3, 3], padding=[1, 1], init=False); conv.weight = dev(tf, wt); conv.bias = dev(tf, bc)
        yc = conv(dev(tf, xc), bias_nc=dev(tf, e), residual=dev(tf, r), gn=32)
    finally:
        lib.tf_gemm_force_config(0, 0, 0)
    want = O.conv2d_bias(xc, wt, bc, (1, 1)) + torch.from_numpy(e)[:, :, None, None] + torch.from_numpy(r)
    close(yc.numpy(), want.numpy())
    assert yc.gn is not None
    g = GroupNorm(32, cout, init=False); g.weight = dev(tf, np.ones(cout, np.float32), "row"); g.bias = dev(tf, np.zeros(cout, np.float32), "row")
    close(g(yc, silu=True).numpy(), O.silu(O.group_norm(torch.from_numpy(yc.numpy()), 32, 1e-5)).numpy())
