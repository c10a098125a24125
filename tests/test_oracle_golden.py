"""Pin the CPU oracle: every restated function must reproduce the reference's own outputs
(tests/golden/*.npz, produced by tests/golden/make_golden.py from the reference's Python)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import ops as O
from oracle.unet import (SD15, basic_transformer_block, cross_attention, feed_forward, resblock,
                         spatial_transformer, unet_param_shapes)
from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict

TOL = dict(rtol=1e-5, atol=2e-5)


def close(a, b, **kw):
    t = dict(TOL); t.update(kw)
    np.testing.assert_allclose(np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32), **t)


def test_conv(golden):
    g = golden["ops"]
    close(O.conv_2d(g["conv_x"], g["conv_w3"], (1, 1), (1, 1), (1, 1)), g["conv_y_3x3_s1"])
    close(O.conv_2d(g["conv_x"], g["conv_w3"], (1, 1), (2, 2), (1, 1)), g["conv_y_3x3_s2"])
    close(O.conv_2d(g["conv_x"], g["conv_w1"], (0, 0), (1, 1), (1, 1)), g["conv_y_1x1"])
    close(O.conv2d_bias(g["conv_x"], g["conv_w3"], g["conv_b"], (1, 1)), g["conv_y_module"])
    close(O.conv_2d(g["convt_x"], g["convt_w"], (0, 0), (1, 1), (1, 1)), g["convt_y"])


def test_norms(golden):
    g = golden["ops"]
    close(O.group_norm(g["gn_x"], 32, 1e-5), g["gn_y_plain"])
    close(O.group_norm_affine(g["gn_x"], 32, g["gn_w"], g["gn_b"]), g["gn_y_affine"])
    close(O.group_norm(g["gn2_x"], 2, 1e-5), g["gn2_y"])
    close(O.layer_norm(g["ln_x"], g["ln_w"], g["ln_b"]), g["ln_y"])
    # the reference's GroupNorm == torch's (tests/group_norm.py:38-40)
    ref = torch.nn.functional.group_norm(torch.from_numpy(g["gn_x"]), 32, torch.from_numpy(g["gn_w"]), torch.from_numpy(g["gn_b"]), 1e-5)
    close(ref, g["gn_y_affine"])


def test_linear_geglu(golden):
    g = golden["ops"]
    close(O.linear(g["lin_x"], g["lin_w"], g["lin_b"]), g["lin_y"])
    close(O.linear(g["lin_x"], g["lin_w"]), g["lin_y_nobias"])
    close(O.geglu(g["lin_x"], g["geglu_w"], g["geglu_b"]), g["geglu_y"])


def test_sdpa(golden):
    g = golden["ops"]
    close(O.scaled_dot_product_attention(g["sdpa_q"], g["sdpa_k"], g["sdpa_v"]), g["sdpa_y_self"])
    close(O.scaled_dot_product_attention(g["sdpa_q"], g["sdpa_kc"], g["sdpa_vc"]), g["sdpa_y_cross"])
    mask = np.tril(np.ones((16, 16), dtype=bool))
    close(O.scaled_dot_product_attention(g["sdpa_q"], g["sdpa_k"], g["sdpa_v"], mask), g["sdpa_y_causal"])


def test_activations_embedding_schedule(golden):
    g = golden["ops"]
    for n in ("sigmoid", "silu", "gelu", "quick_gelu"):
        close(getattr(O, n)(g["act_x"]), g["act_" + n])
    close(O.silu(g["act_x"]), g["act_swish"])
    close(O.timestep_embedding(np.array([981]), 320), g["temb_981"], atol=1e-4)
    close(O.timestep_embedding(np.array([1]), 320), g["temb_1"], atol=1e-4)
    close(O.get_alphas_cumprod(), g["alphas_cumprod"], rtol=1e-6, atol=0)
    xp, p0 = O.get_x_prev_and_pred_x0(g["ddim_x"], g["ddim_e"], g["ddim_a_t"], g["ddim_a_prev"])
    close(xp, g["ddim_x_prev"]); close(p0, g["ddim_pred_x0"])


def test_up_down(golden):
    g = golden["ops"]
    close(O.conv2d_bias(O.upsample_nearest2x(g["ud_x"]), g["up_w"], g["up_b"], (1, 1)), g["up_y"])
    close(O.conv2d_bias(g["ud_x"], g["dn_w"], g["dn_b"], (1, 1), (2, 2)), g["dn_y"])


def _block_weights():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from block_shapes import BLOCK_SHAPES
    return {k: v.astype(np.float32) for k, v in synth_state_dict(BLOCK_SHAPES, 3).items()}


def test_blocks(golden):
    g = golden["blocks"]
    W = _block_weights()
    cfg = oracle.UNetConfig(num_groups=32)
    close(resblock(g["res_x"], torch.from_numpy(g["res_emb"]), {k: O.as_t(v) for k, v in W.items()}, "res", cfg), g["res_y"], atol=5e-5)
    Wt = {k: O.as_t(v) for k, v in W.items()}
    ctx = O.as_t(g["st_ctx"])
    t = "st.transformer_blocks.0"
    close(cross_attention(O.as_t(g["blk_x"]), None, Wt, t + ".attn1", 2), g["attn1_y"], atol=5e-5)
    close(cross_attention(O.as_t(g["blk_x"]), ctx, Wt, t + ".attn2", 2), g["attn2_y"], atol=5e-5)
    close(feed_forward(O.as_t(g["blk_x"]), Wt, t + ".ff"), g["ff_y"], atol=5e-5)
    close(basic_transformer_block(O.as_t(g["blk_x"]), ctx, Wt, t, 2), g["blk_y"], atol=1e-4)
    close(spatial_transformer(O.as_t(g["st_x"]), ctx, Wt, "st", 2, cfg), g["st_y"], atol=1e-4)
    # D11: the reference's head merge differs from the LDM-intended one
    y_int = cross_attention(O.as_t(g["blk_x"]), None, Wt, t + ".attn1", 2, head_merge="intended")
    assert float((y_int - torch.from_numpy(g["attn1_y"])).abs().max()) > 1e-2


def test_param_census():
    P = unet_param_shapes(SD15)
    assert len(P) == 686
    assert sum(int(np.prod(s)) for s in P.values()) == 859520964


@pytest.mark.slow
def test_unet_full_sd15(golden):
    """Whole UNet + CFG + DDIM against the reference's trajectory (2 steps, ~1 min on 8 cores)."""
    if "unet_sd15" not in golden:
        pytest.skip("unet_sd15.npz not generated")
    g = golden["unet_sd15"]
    W = {k: v.astype(np.float32) for k, v in synth_state_dict(unet_param_shapes(SD15), 0).items()}
    latent = synth_normal(1234, "sd.latent", (1, 4, 64, 64))
    ctx = synth_normal(1234, "sd.context", (1, 77, 768)); unc = synth_normal(1234, "sd.uncond", (1, 77, 768))
    ts, al, ap = g["timesteps"], g["alphas"], g["alphas_prev"]
    Wt = {k: O.as_t(v) for k, v in W.items()}
    x = oracle.sd_step(unc, ctx, latent, np.array([ts[49]]), al[49:50], ap[49:50], np.array([7.5]), Wt)
    err = float((x - torch.from_numpy(g["x_after_step0"])).abs().max())
    assert err < 2e-4, err


def test_unet50_fixture_agrees_with_the_two_step_fixture(golden):
    """unet50_sd15.npz (the reference's StableDiffusion.__call__ over all 50 steps, make_golden.py unet50 -- with exp / tanh / matmul
    of the cupy stand-in on torch instead of numpy for speed) starts with the very two steps unet_sd15.npz holds (plain numpy stand-in):
    same reference code, same inputs, results equal to float round-off; and it holds the checkpoints the GPU test gates."""
    if "unet50_sd15" not in golden or "unet_sd15" not in golden:
        pytest.skip("fixtures not generated")
    g50, g2 = golden["unet50_sd15"], golden["unet_sd15"]
    for n in (0, 1):
        a, b = g50[f"x_after_step{n}"], g2[f"x_after_step{n}"]
        assert float(np.abs(a - b).max()) < 2e-5 * (1 + float(np.abs(b).max())), n
    for n in (9, 19, 29, 39, 49):
        x = g50[f"x_after_step{n}"]
        assert x.shape == (1, 4, 64, 64) and np.isfinite(x).all()
    np.testing.assert_array_equal(g50["timesteps"], g2["timesteps"])


@pytest.mark.slow
def test_vae_decode_sd15(golden):
    """decode() of the reference (VAE decoder incl. its per-channel AttnBlock) on the golden latent (~40 s on 8 cores)."""
    import os
    p = os.path.join(os.path.dirname(__file__), "golden", "vae_sd15.npz")
    if not os.path.exists(p):
        pytest.skip("vae_sd15.npz not generated")
    g = np.load(p)
    W = {k: v.astype(np.float32) for k, v in synth_state_dict(oracle.vae_decoder_param_shapes(), 0).items()}
    latent = synth_normal(1234, "vae.latent", (1, 4, 64, 64), 0.18215 * 0.8)
    pre, u8 = oracle.sd_decode(latent, W)
    np.testing.assert_allclose(pre.numpy()[:, :, ::4, ::4], g["pre_sub"], rtol=1e-4, atol=2e-4)
    assert int(np.abs(u8[::4, ::4].astype(np.int32) - g["img_sub"].astype(np.int32)).max()) <= 1


def _clip_weights(with_embeddings=True, seed=0):
    shapes = oracle.clip_param_shapes()
    if not with_embeddings:
        shapes = {k: v for k, v in shapes.items() if ".embeddings." not in k}
    return {k: v.astype(np.float32) for k, v in synth_state_dict(shapes, seed).items()}


def test_clip_encoder_stack_vs_reference():
    """12 x CLIPEncoderLayer + final LayerNorm of the reference (vae/encoder.py:39-81) on the golden hidden states."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_text.npz"))
    pre = "cond_stage_model.transformer.text_model."
    W = _clip_weights(False)
    h = synth_normal(1234, "clip.hidden", (1, 77, 768), 0.05)
    mask = oracle.clip.causal_mask(77)
    l0 = oracle.clip_encoder_layer(O.as_t(h), W, pre + "encoder.layers.0.", mask, 12)
    close(l0.numpy()[:, ::4], g["layer0"], rtol=1e-4, atol=1e-4)
    y = O.layer_norm(oracle.clip_encoder(h, W, pre, mask), W[pre + "final_layer_norm.weight"], W[pre + "final_layer_norm.bias"])
    close(y.numpy(), g["out"], rtol=2e-4, atol=5e-4)


def test_clip_text_transformer_vs_huggingface():
    """The whole text encoder (embedding gathers included, which the reference gets wrong: SURVEY D7) against
    transformers.CLIPTextModel carrying the same weights: the class tree of vae/encoder.py:36-81 is a transcription of it."""
    transformers = pytest.importorskip("transformers")
    cfg = transformers.CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                      num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu")
    model = transformers.CLIPTextModel(cfg).eval()
    pre = "cond_stage_model.transformer.text_model."
    W = _clip_weights(True)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in W.items():
            name = k[len(pre):]
            sd["text_model." + name if "text_model." + name in sd else name].copy_(torch.from_numpy(v))
    rng = np.random.default_rng(5)
    ids = rng.integers(0, 49408, size=(2, 77))
    ids[:, 0] = 49406; ids[0, 9:] = 49407; ids[1, 30:] = 49407          # <start> ... <end> padding, as the tokenizer emits
    with torch.no_grad():
        want = model(input_ids=torch.from_numpy(ids)).last_hidden_state.numpy()
    got = oracle.clip_text_transformer(ids, W).numpy()
    close(got, want, rtol=2e-4, atol=5e-4)


@pytest.mark.slow
def test_fp8_layer_policy_meets_the_config5_gate_on_the_oracle():
    """BASELINE config 5's precision gate (UNet rel-L2 <= 0.1 vs fp32) for the layer policy the HIP path implements: e4m3 operands for
    the ResBlocks' 3x3 convolutions, the FeedForward pair and the attention projections (K >= 640), per-output-channel weight scales,
    block-scaled activations -- evaluated on the CPU oracle with e4m3 emulation (oracle/fp8.py) at every shape of a batch-2 forward
    (assume_supported: on the device the kernel takes these layers at config 5's row counts); with the shape rule of the device a batch-2
    forward at 64 x 64 quantises only the first level's convolutions; the contexts restore the fp32 functions."""
    import oracle
    from oracle import fp8 as O8
    from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict
    W = {k: torch.from_numpy(v.astype(np.float32)) for k, v in synth_state_dict(oracle.unet_param_shapes(oracle.SD15), 0).items()}
    x = torch.from_numpy(synth_normal(1234, "sd.latent", (1, 4, 64, 64))).repeat(2, 1, 1, 1)
    ctx = torch.from_numpy(np.concatenate([synth_normal(1234, "sd.uncond", (1, 77, 768)), synth_normal(1234, "sd.context", (1, 77, 768))]))
    ref = oracle.unet_forward(x, np.array([981.0], np.float32), ctx, W)
    with O8.policy(assume_supported=True):
        got = oracle.unet_forward(x, np.array([981.0], np.float32), ctx, W)
    with O8.policy():
        part = oracle.unet_forward(x, np.array([981.0], np.float32), ctx, W)
    again = oracle.unet_forward(x, np.array([981.0], np.float32), ctx, W)
    assert torch.equal(again, ref)                         # the policy contexts restored the fp32 functions
    # with the device's shape rule fewer layers qualify at 8192 rows (the 320-channel convs do, on the 192 x 128 tile: 43 x 3 = 129 blocks)
    assert 0.0 < float((part - ref).norm() / ref.norm()) < float((got - ref).norm() / ref.norm())
    assert O8.mx_gemm_supported(8 * 9216, 320, 2880, howo=9216, c_parts=(320,)) and not O8.mx_gemm_supported(8 * 144, 1280, 11520, howo=144)
    assert O8.mx_gemm_supported(8192, 320, 2880, howo=4096, c_parts=(320,)) and not O8.mx_gemm_supported(2048, 640, 5760, howo=1024)
    rl2 = float((got - ref).norm() / ref.norm())
    assert 0.01 < rl2 <= 0.1, rl2
