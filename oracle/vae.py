"""CPU restatement of the reference's VAE: post_quant_conv + Decoder + image tail (the sampler's side), and the
Encoder + quant_conv of AutoencoderKL.__call__.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Weight names are the LDM keys the reference's update_state walk
produces from the StableDiffusion root (``first_stage_model.decoder...``)."""
import numpy as np
import torch

from . import ops

__all__ = ["vae_decoder_param_shapes", "vae_encoder_param_shapes", "resnet_block", "attn_block", "vae_decoder", "vae_encoder", "autoencoder_kl", "sd_decode"]

_SZ = [(128, 256), (256, 512), (512, 512), (512, 512)]


def vae_decoder_param_shapes(prefix="first_stage_model."):
    """vae/vae.py:10, vae/decoder.py:9-20, vae/mid.py:6-9, attention/attention.py:11-16, vision/resnet.py:34-39."""
    P = {}

    def conv(p, i, o, k):
        P[p + ".weight"] = (o, i, k, k); P[p + ".bias"] = (o,)

    def norm(p, c):
        P[p + ".weight"] = (c,); P[p + ".bias"] = (c,)

    def res(p, i, o):
        norm(p + ".norm1", i); conv(p + ".conv1", i, o, 3); norm(p + ".norm2", o); conv(p + ".conv2", o, o, 3)
        if i != o:
            conv(p + ".nin_shortcut", i, o, 1)
    d = prefix + "decoder"
    conv(prefix + "post_quant_conv", 4, 4, 1)
    conv(d + ".conv_in", 4, 512, 3)
    res(d + ".mid.block_1", 512, 512)
    norm(d + ".mid.attn_1.norm", 512)
    for n in ("q", "k", "v", "proj_out"):
        conv(d + ".mid.attn_1." + n, 512, 512, 1)
    res(d + ".mid.block_2", 512, 512)
    for i, s in enumerate(_SZ):
        res(f"{d}.up.{i}.block.0", s[1], s[0]); res(f"{d}.up.{i}.block.1", s[0], s[0]); res(f"{d}.up.{i}.block.2", s[0], s[0])
        if i != 0:
            conv(f"{d}.up.{i}.upsample.conv", s[0], s[0], 3)
    norm(d + ".norm_out", 128); conv(d + ".conv_out", 128, 3, 3)
    return P


def resnet_block(x, W, p):
    """vision/resnet.py:41-45 -- ResnetBlock.__call__."""
    h = ops.conv2d_bias(ops.silu(ops.group_norm_affine(x, 32, W[p + ".norm1.weight"], W[p + ".norm1.bias"])), W[p + ".conv1.weight"], W[p + ".conv1.bias"], (1, 1))
    h = ops.conv2d_bias(ops.silu(ops.group_norm_affine(h, 32, W[p + ".norm2.weight"], W[p + ".norm2.bias"])), W[p + ".conv2.weight"], W[p + ".conv2.bias"], (1, 1))
    if p + ".nin_shortcut.weight" in W:
        x = ops.conv2d_bias(x, W[p + ".nin_shortcut.weight"], W[p + ".nin_shortcut.bias"])
    return x + h


def attn_block(x, W, p, head_merge="reference_exact"):
    """attention/attention.py:19-24 -- AttnBlock.__call__.  q/k/v stay NCHW and go straight into
    scaled_dot_product_attention, which therefore reads them as (B, NH, T, HS) = (b, c, h, w) (reference-exact).
    head_merge='intended' is the LDM form the comment at :18 refers to ("copied from AttnBlock in ldm repo"): ONE head of size c
    attending over the h*w pixels (what real SD weights need)."""
    h_ = ops.group_norm_affine(x, 32, W[p + ".norm.weight"], W[p + ".norm.bias"])
    q, k, v = [ops.conv2d_bias(h_, W[f"{p}.{n}.weight"], W[f"{p}.{n}.bias"]) for n in ("q", "k", "v")]
    if head_merge == "intended":
        b, c, h, w = q.shape
        tok = lambda t: t.reshape(b, c, h * w).permute(0, 2, 1).reshape(b, 1, h * w, c)
        h_tf = ops.scaled_dot_product_attention(tok(q), tok(k), tok(v)).reshape(b, h * w, c).permute(0, 2, 1).reshape(b, c, h, w)
    else:
        h_tf = ops.scaled_dot_product_attention(q, k, v)
    return x + ops.conv2d_bias(h_tf, W[p + ".proj_out.weight"], W[p + ".proj_out.bias"])


def vae_decoder(x, W, d="first_stage_model.decoder", head_merge="reference_exact"):
    """vae/decoder.py:22-34 -- Decoder.__call__."""
    x = ops.conv2d_bias(x, W[d + ".conv_in.weight"], W[d + ".conv_in.bias"], (1, 1))
    x = resnet_block(x, W, d + ".mid.block_1"); x = attn_block(x, W, d + ".mid.attn_1", head_merge); x = resnet_block(x, W, d + ".mid.block_2")
    for i in reversed(range(4)):
        for j in range(3):
            x = resnet_block(x, W, f"{d}.up.{i}.block.{j}")
        if i != 0:
            x = ops.conv2d_bias(ops.upsample_nearest2x(x), W[f"{d}.up.{i}.upsample.conv.weight"], W[f"{d}.up.{i}.upsample.conv.bias"], (1, 1))
    x = ops.silu(ops.group_norm_affine(x, 32, W[d + ".norm_out.weight"], W[d + ".norm_out.bias"]))
    return ops.conv2d_bias(x, W[d + ".conv_out.weight"], W[d + ".conv_out.bias"], (1, 1))


def sd_decode(latent, W, prefix="first_stage_model.", head_merge="reference_exact"):
    """variants/sd.py:48-54 -- decode: returns (float image in [-..], uint8 HWC image)."""
    W = {k: ops.as_t(v) for k, v in W.items()}
    x = ops.conv2d_bias(1 / 0.18215 * ops.as_t(latent), W[prefix + "post_quant_conv.weight"], W[prefix + "post_quant_conv.bias"])
    x = vae_decoder(x, W, prefix + "decoder", head_merge)
    img = (x + 1.0) / 2.0
    h, w = img.shape[2], img.shape[3]
    u8 = (torch.clip(img.reshape(3, h, w).permute(1, 2, 0), 0, 1) * 255).numpy().astype(np.uint8)
    return x, u8


_ESZ = [(128, 128), (128, 256), (256, 512), (512, 512)]


def vae_encoder_param_shapes(prefix="first_stage_model."):
    """vae/vae.py:8-9, vae/encoder.py:13-26 (+ mid / attention / resnet as for the decoder)."""
    P = {}

    def conv(p, i, o, k):
        P[p + ".weight"] = (o, i, k, k); P[p + ".bias"] = (o,)

    def norm(p, c):
        P[p + ".weight"] = (c,); P[p + ".bias"] = (c,)

    def res(p, i, o):
        norm(p + ".norm1", i); conv(p + ".conv1", i, o, 3); norm(p + ".norm2", o); conv(p + ".conv2", o, o, 3)
        if i != o:
            conv(p + ".nin_shortcut", i, o, 1)
    e = prefix + "encoder"
    conv(prefix + "quant_conv", 8, 8, 1)
    conv(e + ".conv_in", 3, 128, 3)
    for i, s in enumerate(_ESZ):
        res(f"{e}.down.{i}.block.0", s[0], s[1]); res(f"{e}.down.{i}.block.1", s[1], s[1])
        if i != 3:
            conv(f"{e}.down.{i}.downsample.conv", s[1], s[1], 3)
    res(e + ".mid.block_1", 512, 512)
    norm(e + ".mid.attn_1.norm", 512)
    for n in ("q", "k", "v", "proj_out"):
        conv(e + ".mid.attn_1." + n, 512, 512, 1)
    res(e + ".mid.block_2", 512, 512)
    norm(e + ".norm_out", 512); conv(e + ".conv_out", 512, 8, 3)
    return P


def vae_encoder(x, W, e="first_stage_model.encoder", head_merge="reference_exact"):
    """vae/encoder.py:28-34 -- Encoder.__call__.  The stride-2 convs carry ``padding=[0,1,0,1]`` (:19): one zero pixel on the right and at
    the bottom (the LDM / tinygrad form this file was taken from; a 4-list is not a padding cuDNN's conv_fprop or torch's conv2d accept,
    so the reference's own call cannot run as written -- the restated semantics are the intended ones: PARITY UNPINNED for this one call,
    every other op of the encoder is the decoder's, pinned by tests/golden/vae_sd15.npz)."""
    x = ops.conv2d_bias(ops.as_t(x), W[e + ".conv_in.weight"], W[e + ".conv_in.bias"], (1, 1))
    for i in range(4):
        for j in range(2):
            x = resnet_block(x, W, f"{e}.down.{i}.block.{j}")
        if i != 3:
            x = torch.nn.functional.pad(x, (0, 1, 0, 1))
            x = ops.conv2d_bias(x, W[f"{e}.down.{i}.downsample.conv.weight"], W[f"{e}.down.{i}.downsample.conv.bias"], (0, 0), (2, 2))
    x = resnet_block(x, W, e + ".mid.block_1"); x = attn_block(x, W, e + ".mid.attn_1", head_merge); x = resnet_block(x, W, e + ".mid.block_2")
    x = ops.silu(ops.group_norm_affine(x, 32, W[e + ".norm_out.weight"], W[e + ".norm_out.bias"]))
    return ops.conv2d_bias(x, W[e + ".conv_out.weight"], W[e + ".conv_out.bias"], (1, 1))


def autoencoder_kl(x, W, prefix="first_stage_model.", head_merge="reference_exact"):
    """vae/vae.py:12-18 -- AutoencoderKL.__call__: returns (latent means, reconstruction)."""
    W = {k: ops.as_t(v) for k, v in W.items()}
    lat = ops.conv2d_bias(vae_encoder(x, W, prefix + "encoder", head_merge), W[prefix + "quant_conv.weight"], W[prefix + "quant_conv.bias"])[:, 0:4]
    z = ops.conv2d_bias(lat, W[prefix + "post_quant_conv.weight"], W[prefix + "post_quant_conv.bias"])
    return lat, vae_decoder(z, W, prefix + "decoder", head_merge)
