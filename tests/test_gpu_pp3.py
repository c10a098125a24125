"""GPU parity tests (pytest -m gpu) of k_igemm_pp3 -- the PATCH form of the ping-pong kernel: 3x3 / stride 1 / pad 1 convolutions whose 192-row
tile is a whole number of image rows (the 96 / 48 / 24-pixel levels of BASELINE config 5), the activation patch of a 64-channel slab staged once
for its nine taps.  Forced through tf_gemm_force_config(192, BN, 1) + tf_gemm_debug(2048), which fails loudly where the kernel cannot take a
launch.  Exact small-integer convolutions first (a wrong patch row, tap offset, swizzle or counted wait is an O(1) error), then the reference's
conv (vision/conv2d.py:9-58) with its ResBlock surroundings (time embedding, residual, concat input, GroupNorm statistics of the output:
vision/resnet.py:6-31) against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_pp import close, dev, forced, rnd, tf   # noqa: E402,F401  (the same fixtures and helpers)


def conv_nchw(x, w, pad=1):
    return torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=pad).numpy()


@pytest.mark.parametrize("n,c1,c2,hw,cout", [(1, 64, 0, 24, 64), (2, 64, 0, 24, 200), (3, 128, 64, 24, 320), (2, 64, 64, 48, 136), (1, 192, 0, 96, 160), (2, 320, 0, 48, 320),
                                             (1, 64, 64, 96, 200), (1, 128, 0, 48, 64), (2, 256, 64, 24, 136), (1, 320, 0, 96, 4), (2, 64, 0, 48, 4)])     # (4 channels: conv_out, unet.py:49)
def test_pp3_conv_exact_integers(tf, n, c1, c2, hw, cout):
    """1 ... 5 channel slabs (prologue-only patch, the double buffer's both parities), image borders on every side of a tile, ragged channel tiles,
    the concat pair, several images; bias + time embedding + residual in the shared epilogue.  Integers small enough that every partial sum is
    exact in fp16 / fp32: bit-exact against a float32 convolution."""
    from tinyfusers_amd.vision.conv2d import Conv2d
    rs = np.random.RandomState(n * 1000 + c1 + c2 + hw + cout)
    cin = c1 + c2
    xa = rs.randint(-1, 2, (n, c1, hw, hw)).astype(np.float32)
    xb = rs.randint(-1, 2, (n, c2, hw, hw)).astype(np.float32) if c2 else None
    wt = rs.randint(-1, 2, (cout, cin, 3, 3)).astype(np.float32)
    b = rs.randint(-4, 5, (cout,)).astype(np.float32); e = rs.randint(-3, 4, (n, cout)).astype(np.float32)
    r = rs.randint(-8, 9, (n, cout, hw, hw)).astype(np.float32)
    m = Conv2d(cin, cout, [3, 3], padding=[1, 1], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    x = (dev(tf, xa), dev(tf, xb)) if c2 else dev(tf, xa)
    ed, rd = dev(tf, e), dev(tf, r)
    with forced(160, 1, 2048, 192):        # (the tile width is the instance's own: 160 on 96-pixel rows, 128 on 48 / 24)
        got = m(x, bias_nc=ed, residual=rd).numpy()
    want = conv_nchw(np.concatenate((xa, xb), 1) if c2 else xa, wt) + b[None, :, None, None] + e[:, :, None, None] + r
    assert np.abs(want).max() < 2048
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("n,c,hs,cout", [(2, 64, 12, 64), (1, 192, 24, 136), (1, 128, 48, 320), (3, 128, 12, 200)])
def test_pp3_conv_with_folded_upsampling_exact_integers(tf, n, c, hs, cout):
    """Upsample (vision/unet.py:79-86 of the reference: nearest 2x, then conv 3x3): the patch gather reads source pixel (y >> 1, x >> 1); output rows of
    24 / 48 / 96 pixels from 12 / 24 / 48-pixel sources, image borders of the UP-SAMPLED image."""
    from tinyfusers_amd.vision.conv2d import Conv2d
    rs = np.random.RandomState(n * 100 + c + hs + cout)
    x = rs.randint(-1, 2, (n, c, hs, hs)).astype(np.float32); wt = rs.randint(-1, 2, (cout, c, 3, 3)).astype(np.float32)
    b = rs.randint(-4, 5, (cout,)).astype(np.float32)
    m = Conv2d(c, cout, [3, 3], padding=[1, 1], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    with forced(160, 1, 2048 | 64, 192):      # (64: the m-fastest tile order, as most of the shipped table's rows)
        got = m(dev(tf, x), upsample=True).numpy()
    want = conv_nchw(x.repeat(2, axis=2).repeat(2, axis=3), wt) + b[None, :, None, None]
    assert np.abs(want).max() < 2048
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("n,c1,c2,hw,cout,c3,c4", [(2, 64, 0, 24, 128, 64, 0), (1, 192, 0, 48, 200, 192, 0), (1, 64, 0, 96, 320, 128, 64), (2, 128, 0, 24, 136, 64, 320),
                                                   (1, 192, 0, 48, 64, 448, 0)])
def test_pp3_conv_with_folded_skip_projection_exact_integers(tf, n, c1, c2, hw, cout, c3, c4):
    """ResBlock's conv2 with the 1x1 skip projection folded in (vision/resnet.py:23-31 of the reference: h + skip(x)): 1 ... 7 extra K tiles behind the nine
    taps (the three activation slots over the patch buffers, both parities of the slab count, the concat pair as skip source; the conv's own input
    is one tensor there: vision/conv2d.py's fold takes no concat pair)."""
    from tinyfusers_amd.vision.conv2d import Conv2d
    rs = np.random.RandomState(n + c1 + c2 + hw + cout + c3 + c4)
    cin = c1 + c2
    ri = lambda *sh: rs.randint(-1, 2, sh).astype(np.float32)
    xa, xb = ri(n, c1, hw, hw), (ri(n, c2, hw, hw) if c2 else None)
    x3, x4 = ri(n, c3, hw, hw), (ri(n, c4, hw, hw) if c4 else None)
    wt, ws = ri(cout, cin, 3, 3), ri(cout, c3 + c4, 1, 1)
    b, bs = rs.randint(-4, 5, (cout,)).astype(np.float32), rs.randint(-4, 5, (cout,)).astype(np.float32)
    m = Conv2d(cin, cout, [3, 3], padding=[1, 1], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    proj = Conv2d(c3 + c4, cout, [1, 1], init=False); proj.weight = dev(tf, ws); proj.bias = dev(tf, bs)
    x = (dev(tf, xa), dev(tf, xb)) if c2 else dev(tf, xa)
    xs = (dev(tf, x3), dev(tf, x4)) if c4 else dev(tf, x3)
    with forced(160, 1, 2048, 192):
        got = m(x, extra=(proj, xs)).numpy()
    want = conv_nchw(np.concatenate((xa, xb), 1) if c2 else xa, wt) + conv_nchw(np.concatenate((x3, x4), 1) if c4 else x3, ws, pad=0) + (b + bs)[None, :, None, None]
    assert np.abs(want).max() < 2048
    np.testing.assert_array_equal(got, want)


def test_pp3_refuses_what_it_cannot_run(tf):
    """32-pixel rows do not divide the 192-row tile, stride 2 and 1x1 are other kernels' work: an explicit request must fail, not run something else."""
    from tinyfusers_amd.vision.conv2d import Conv2d
    for hw, k, stride in ((32, 3, 1), (48, 3, 2), (48, 1, 1)):
        m = Conv2d(64, 128, [k, k], stride=[stride, stride], padding=[k // 2, k // 2], init=False)
        m.weight = dev(tf, rnd("p3r.w", (128, 64, k, k))); m.bias = None
        with forced(160, 1, 2048, 192):
            with pytest.raises(RuntimeError):
                m(dev(tf, rnd("p3r.x", (2, 64, hw, hw))))


@pytest.mark.parametrize("n,c1,c2,hw,cout,bn,gn", [(2, 128, 64, 24, 320, 160, 32), (4, 320, 0, 48, 320, 160, 32), (2, 128, 0, 48, 128, 128, 32), (3, 64, 0, 24, 128, 128, 0),
                                                    (1, 64, 64, 96, 320, 160, 32), (2, 640, 320, 24, 640, 128, 32)])
def test_pp3_conv2d_against_the_oracle(tf, n, c1, c2, hw, cout, bn, gn):
    from oracle import ops as O
    from tinyfusers_amd.ff.group_norm import GroupNorm
    from tinyfusers_amd.vision.conv2d import Conv2d
    xa = rnd("p3c.xa", (n, c1, hw, hw)); xb = rnd("p3c.xb", (n, c2, hw, hw)) if c2 else None
    cin = c1 + c2
    wt = rnd("p3c.w", (cout, cin, 3, 3), (cin * 9) ** -0.5); b = rnd("p3c.b", (cout,), 0.1)
    m = Conv2d(cin, cout, [3, 3], padding=[1, 1], init=False); m.weight = dev(tf, wt); m.bias = dev(tf, b)
    x = (dev(tf, xa), dev(tf, xb)) if c2 else dev(tf, xa)
    e = rnd("p3c.e", (n, cout), 0.5); r = rnd("p3c.r", (n, cout, hw, hw))
    want = O.conv2d_bias(torch.from_numpy(np.concatenate((xa, xb), 1) if c2 else xa), wt, b, (1, 1)) + torch.from_numpy(e)[:, :, None, None] + torch.from_numpy(r)
    with forced(bn, 1, 2048, 192):
        y = m(x, gn=gn, bias_nc=dev(tf, e), residual=dev(tf, r))
        got = y.numpy()
    close(got, want.numpy())
    if gn:
        assert y.gn is not None, "the statistics of the output did not ride on the conv"
        g = GroupNorm(gn, cout, init=False); g.weight = dev(tf, rnd("p3c.g", (cout,), 0.2) + 1.0, "row"); g.bias = dev(tf, rnd("p3c.gb", (cout,), 0.1), "row")
        close(g(y, silu=True).numpy(), O.silu(O.group_norm_affine(torch.from_numpy(got), gn, g.weight.numpy(), g.bias.numpy(), 1e-5)).numpy())


@pytest.mark.parametrize("c,hw,cout", [(320, 96, 320), (640, 48, 640), (1280, 24, 1280)])
def test_pp3_agrees_with_the_deep_ring_kernel_at_config5_size(tf, c, hw, cout):
    """BASELINE config 5's three 3x3 geometries at full size (UNet batch 8; too large for the CPU oracle in a test): against the round-1 deep-ring
    kernel on the same inputs (same products, another summation order), plus linearity conv(2 x) = 2 conv(x), exact in floating point."""
    from tinyfusers_amd.native import lib
    from tinyfusers_amd.vision.conv2d import Conv2d
    n = 8
    x = rnd("p35.x", (n, c, hw, hw)); wt = rnd("p35.w", (cout, c, 3, 3), (c * 9) ** -0.5)
    m = Conv2d(c, cout, [3, 3], padding=[1, 1], init=False); m.weight = dev(tf, wt); m.bias = None
    xd = dev(tf, x)
    with forced(160, 1, 2048, 192):
        y_pp = m(xd).numpy()
        y_pp2 = m(dev(tf, 2.0 * x)).numpy()
    lib.tf_gemm_force_config(128, 128, 1); lib.tf_gemm_debug(8)
    try:
        y_ref = m(xd).numpy()
    finally:
        lib.tf_gemm_force_config(0, 0, 0); lib.tf_gemm_debug(0)
    assert np.isfinite(y_pp).all()
    np.testing.assert_allclose(y_pp, y_ref, atol=4e-3, rtol=4e-3)
    np.testing.assert_allclose(y_pp2, 2.0 * y_pp, rtol=0, atol=1.2e-7)
