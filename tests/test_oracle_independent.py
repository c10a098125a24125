"""Second opinion on the two oracle operators whose golden fixtures were produced through the same torch kernels the oracle calls
(under the stub import the reference's conv_2d and layer_norm became F.conv2d / F.layer_norm, which are also the oracle's
bodies, so those goldens pin only the glue).  Here they are restated in plain numpy float64 straight from the definitions the reference's own
tests assert (tests/conv2d.py:27-33: NCHW cross-correlation; tests/layer_norm.py:38-41: normalise over the trailing dims, biased variance)
and the oracle has to agree -- including batches > 1, which the B = 1 `ln_y` fixture does not cover."""
import numpy as np
import pytest

import oracle


def conv2d_numpy(x, w, pad, stride):
    """out[n,k,i,j] = sum_{c,r,s} x[n,c,i*stride+r-pad,j*stride+s-pad] * w[k,c,r,s]   (float64, zero padding)"""
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    n, c, h, wd = x.shape
    k, _, r, s = w.shape
    xp = np.zeros((n, c, h + 2 * pad, wd + 2 * pad))
    xp[:, :, pad:pad + h, pad:pad + wd] = x
    ho, wo = (h + 2 * pad - r) // stride + 1, (wd + 2 * pad - s) // stride + 1
    out = np.zeros((n, k, ho, wo))
    for dr in range(r):
        for ds in range(s):
            patch = xp[:, :, dr:dr + (ho - 1) * stride + 1:stride, ds:ds + (wo - 1) * stride + 1:stride]      # (n, c, ho, wo)
            out += np.einsum("nchw,kc->nkhw", patch, w[:, :, dr, ds])
    return out


@pytest.mark.parametrize("n,c,h,w,k,r,stride,pad", [
    (1, 2, 100, 100, 1, 2, 1, 0),          # the tests/conv2d.py:13-33 family (Cin = 2, 2 x 2 kernel, pad 0), cut to 100 x 100
    (2, 4, 16, 16, 320, 3, 1, 1),          # conv_in of the UNet
    (2, 64, 9, 7, 48, 3, 2, 1),            # Downsample: stride 2, odd sizes
    (3, 40, 6, 5, 24, 1, 1, 0),            # 1 x 1
    (1, 8, 5, 5, 8, 3, 1, 0),              # no padding, 3 x 3
])
def test_oracle_conv_matches_the_definition(n, c, h, w, k, r, stride, pad):
    rng = np.random.default_rng(n * 100 + c)
    x, wt = rng.standard_normal((n, c, h, w)).astype(np.float32), rng.standard_normal((k, c, r, r)).astype(np.float32)
    got = oracle.conv_2d(x, wt, (pad, pad), (stride, stride), (1, 1)).numpy()
    np.testing.assert_allclose(got, conv2d_numpy(x, wt, pad, stride), rtol=1e-5, atol=1e-4)
    b = rng.standard_normal(k).astype(np.float32)
    got_b = oracle.conv2d_bias(x, wt, b, (pad, pad), (stride, stride)).numpy()
    np.testing.assert_allclose(got_b, conv2d_numpy(x, wt, pad, stride) + b[None, :, None, None], rtol=1e-5, atol=1e-4)


def layer_norm_numpy(x, weight, bias, eps, ndims):
    x = np.asarray(x, np.float64)
    ax = tuple(range(x.ndim - ndims, x.ndim))
    mean = x.mean(axis=ax, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=ax, keepdims=True)
    y = (x - mean) / np.sqrt(var + eps)
    if weight is not None:
        y = y * np.asarray(weight, np.float64).reshape(x.shape[x.ndim - ndims:]) + np.asarray(bias, np.float64).reshape(x.shape[x.ndim - ndims:])
    return y


@pytest.mark.parametrize("shape,eps", [((2, 4096, 320), 1e-5), ((5, 77, 768), 1e-5), ((3, 1, 1280), 1e-3), ((16, 10), 1e-3)])
def test_oracle_layer_norm_last_dim_matches_the_definition_for_batches(shape, eps):
    rng = np.random.default_rng(len(shape) + shape[-1])
    x = (rng.standard_normal(shape) * 3 + 1).astype(np.float32)
    g, b = rng.standard_normal(shape[-1]).astype(np.float32), rng.standard_normal(shape[-1]).astype(np.float32)
    np.testing.assert_allclose(oracle.layer_norm(x, g, b, eps).numpy(), layer_norm_numpy(x, g, b, eps, 1), rtol=1e-5, atol=2e-5)


def test_oracle_layer_norm_slab_and_last_dim_forms_of_the_reference_tests():
    """tests/layer_norm.py:22-41 ((1, C, H, W) scale: normalise over [C, H, W]) and :44-71 ((1, 1, 1, W) scale: over [W]), batch 4."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((4, 24, 10, 10)).astype(np.float32)
    g, b = rng.standard_normal((1, 24, 10, 10)).astype(np.float32), rng.standard_normal((1, 24, 10, 10)).astype(np.float32)
    np.testing.assert_allclose(oracle.layer_norm(x, g, b, 1e-3).numpy(), layer_norm_numpy(x, g[0], b[0], 1e-3, 3), rtol=1e-5, atol=2e-5)
    g2, b2 = rng.standard_normal((1, 1, 1, 10)).astype(np.float32), rng.standard_normal((1, 1, 1, 10)).astype(np.float32)
    np.testing.assert_allclose(oracle.layer_norm(x, g2, b2, 1e-3).numpy(), layer_norm_numpy(x, g2.reshape(10), b2.reshape(10), 1e-3, 1), rtol=1e-5, atol=2e-5)
