"""GPU parity tests (pytest -m gpu) of the VAE encoder side -- vae/encoder.py:12-34 and AutoencoderKL.__call__ (vae/vae.py:12-18) --
against the oracle (oracle/vae.py).  The decoder side is pinned by the reference's own image (tests/test_gpu_model.py,
tests/golden/vae_sd15.npz); the encoder shares every op with it except the asymmetric [0, 1, 0, 1] padding of its stride-2 convs."""
import contextlib
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    import tinyfusers_amd.storage.tensor as T
    T.ensure_init(0)
    return T


def rnd(name, shape, std=1.0, seed=41):
    from tinyfusers_amd.storage.synth import synth_normal
    return synth_normal(seed, name, shape, std).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("n,c,h,w,cout", [(1, 64, 16, 16, 64), (2, 128, 10, 14, 128), (1, 8, 7, 9, 16), (2, 512, 8, 8, 512)])
def test_stride2_conv_with_right_bottom_padding(tf, n, c, h, w, cout):
    """padding=[0, 1, 0, 1] (vae/encoder.py:19): one zero pixel right and bottom, stride 2 -- even and odd sizes."""
    from oracle import ops as O
    from tinyfusers_amd.vision.conv2d import Conv2d
    x = rnd("pad.x", (n, c, h, w)); wt = rnd("pad.w", (cout, c, 3, 3), (9 * c) ** -0.5); b = rnd("pad.b", (cout,), 0.1)
    m = Conv2d(c, cout, [3, 3], stride=[2, 2], padding=[0, 1, 0, 1], init=False)
    m.weight = tf.DeviceArray.from_numpy(wt); m.bias = tf.DeviceArray.from_numpy(b, np.float16, "row")
    got = m(tf.DeviceArray.from_numpy(x)).numpy()
    want = O.conv2d_bias(torch.nn.functional.pad(torch.from_numpy(x), (0, 1, 0, 1)), wt, b, (0, 0), (2, 2)).numpy()
    assert got.shape == want.shape == (n, cout, (h - 2) // 2 + 1, (w - 2) // 2 + 1)
    np.testing.assert_allclose(got, want, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("size,head_merge", [(64, "reference_exact"), (128, "reference_exact"), (96, "intended")])
def test_encoder_and_autoencoder_round_trip_against_the_oracle(tf, size, head_merge):
    """Encoder + quant_conv -> means, then post_quant_conv + Decoder: the device against the CPU oracle on the same synthetic weights
    (reference_exact: the AttnBlock as the reference runs it -- one 'head' per channel over the h x w matrix, so w must be a multiple of 8
    for the fused kernel; intended: the LDM single-head form, any size)."""
    import oracle
    from tinyfusers_amd import config
    old_merge = config.head_merge
    config.head_merge = head_merge
    try:
        _round_trip(tf, size, head_merge)
    finally:
        config.head_merge = old_merge


def _round_trip(tf, size, head_merge):
    import oracle
    from tinyfusers_amd.storage.state import param_shapes, update_state
    from tinyfusers_amd.storage.synth import synth_state_dict
    from tinyfusers_amd.vae.vae import AutoencoderKL
    vae = AutoencoderKL(init=False)
    shapes = param_shapes(vae, "first_stage_model")
    want_names = set(oracle.vae_decoder_param_shapes()) | set(oracle.vae_encoder_param_shapes())
    assert set(shapes) == want_names
    W = synth_state_dict(shapes, 0)
    with contextlib.redirect_stdout(io.StringIO()) as out:
        update_state(vae, W, "first_stage_model")
    assert "skipped" not in out.getvalue()
    img = rnd("vae.img", (1, 3, size, size), 0.5)
    x = tf.DeviceArray.from_numpy(img)
    means = vae.encode(x).numpy()
    rec = vae(x).numpy()
    Wf = {k: v.astype(np.float16).astype(np.float32) for k, v in W.items()}
    lat, want = oracle.autoencoder_kl(img, Wf, head_merge=head_merge)
    lat, want = lat.numpy(), want.numpy()
    assert means.shape == lat.shape == (1, 4, size // 8, size // 8) and rec.shape == want.shape == (1, 3, size, size)
    rl = np.linalg.norm(means - lat) / np.linalg.norm(lat)
    rr = np.linalg.norm(rec - want) / np.linalg.norm(want)
    print(f"VAE encoder means rel-L2 {rl:.2e}, round trip rel-L2 {rr:.2e}")
    assert np.isfinite(rec).all() and rl < 5e-3 and rr < 1e-2
