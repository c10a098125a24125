"""CPU restatement of the reference's UNet graph, blocks and sampler step (fp32, torch-CPU).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional style: weights come in a flat dict keyed
by the LDM checkpoint names that the reference's ``update_state`` walk produces
(storage/state.py:4-23), relative to the UNet root (``input_blocks.1.0.in_layers.0.weight`` ...).
"""
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch

from . import ops

__all__ = ["UNetConfig", "SD15", "TINY", "unet_param_shapes", "resblock", "cross_attention",
           "basic_transformer_block", "spatial_transformer", "feed_forward", "unet_forward",
           "sd_step", "sampler_schedule"]


@dataclass(frozen=True)
class UNetConfig:
    """Shape parameters of vision/unet.py:9-49 (hard-coded there to the SD-1.x values)."""
    in_channels: int = 4
    out_channels: int = 4
    model_channels: int = 320
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    attention_levels: Tuple[int, ...] = (0, 1, 2)
    n_heads: int = 8
    context_dim: int = 768
    num_groups: int = 32

    @property
    def emb_channels(self):
        return self.model_channels * 4


SD15 = UNetConfig()
# down-scaled graph with the same topology, for second-scale tests
TINY = UNetConfig(model_channels=64, channel_mult=(1, 2, 2), attention_levels=(0, 1), n_heads=2,
                  context_dim=64)


def _graph(cfg: UNetConfig):
    """Block list of vision/unet.py:12-44 generated from the channel plan.
    Returns (input_blocks, middle, output_blocks); each block is a list of
    ('conv', cin, cout) | ('res', cin, cout) | ('st', ch) | ('down', ch) | ('up', ch)."""
    mc = cfg.model_channels
    inp = [[("conv", cfg.in_channels, mc)]]
    chans = [mc]
    ch = mc
    nlev = len(cfg.channel_mult)
    for lev, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            blk = [("res", ch, mc * mult)]
            ch = mc * mult
            if lev in cfg.attention_levels:
                blk.append(("st", ch))
            inp.append(blk)
            chans.append(ch)
        if lev != nlev - 1:
            inp.append([("down", ch)])
            chans.append(ch)
    mid = [("res", ch, ch), ("st", ch), ("res", ch, ch)]
    out = []
    for lev in reversed(range(nlev)):
        mult = cfg.channel_mult[lev]
        for i in range(cfg.num_res_blocks + 1):
            skip = chans.pop()
            blk = [("res", ch + skip, mc * mult)]
            ch = mc * mult
            if lev in cfg.attention_levels:
                blk.append(("st", ch))
            if lev > 0 and i == cfg.num_res_blocks:
                blk.append(("up", ch))
            out.append(blk)
    return inp, mid, out


def unet_param_shapes(cfg: UNetConfig = SD15) -> Dict[str, tuple]:
    """Every weight/bias leaf of the UNet with its shape, in update_state naming (storage/state.py)."""
    P: Dict[str, tuple] = {}
    emb = cfg.emb_channels

    def lin(p, i, o, bias=True):
        P[p + ".weight"] = (o, i)
        if bias:
            P[p + ".bias"] = (o,)

    def conv(p, i, o, k):
        P[p + ".weight"] = (o, i, k, k)
        P[p + ".bias"] = (o,)

    def norm(p, c):
        P[p + ".weight"] = (c,)
        P[p + ".bias"] = (c,)

    def res(p, i, o):
        norm(p + ".in_layers.0", i); conv(p + ".in_layers.2", i, o, 3)
        lin(p + ".emb_layers.1", emb, o)
        norm(p + ".out_layers.0", o); conv(p + ".out_layers.3", o, o, 3)
        if i != o:
            conv(p + ".skip_connection", i, o, 1)

    def st(p, c):
        norm(p + ".norm", c); conv(p + ".proj_in", c, c, 1)
        t = p + ".transformer_blocks.0"
        for a, cd in (("attn1", c), ("attn2", cfg.context_dim)):
            lin(f"{t}.{a}.to_q", c, c, False); lin(f"{t}.{a}.to_k", cd, c, False); lin(f"{t}.{a}.to_v", cd, c, False)
            lin(f"{t}.{a}.to_out.0", c, c)
        lin(t + ".ff.net.0.proj", c, c * 8); lin(t + ".ff.net.2", c * 4, c)
        norm(t + ".norm1", c); norm(t + ".norm2", c); norm(t + ".norm3", c)
        conv(p + ".proj_out", c, c, 1)

    def block(p, b):
        for j, l in enumerate(b):
            q = f"{p}.{j}"
            if l[0] == "conv": conv(q, l[1], l[2], 3)
            elif l[0] == "res": res(q, l[1], l[2])
            elif l[0] == "st": st(q, l[1])
            elif l[0] == "down": conv(q + ".op", l[1], l[1], 3)
            elif l[0] == "up": conv(q + ".conv", l[1], l[1], 3)

    lin("time_embed.0", cfg.model_channels, emb); lin("time_embed.2", emb, emb)
    inp, mid, out = _graph(cfg)
    for i, b in enumerate(inp): block(f"input_blocks.{i}", b)
    block("middle_block", mid)
    for i, b in enumerate(out): block(f"output_blocks.{i}", b)
    norm("out.0", cfg.model_channels); conv("out.2", cfg.model_channels, cfg.out_channels, 3)
    return P


# ----------------------------------------------------------------------------- blocks

def resblock(x, emb, W, p, cfg=SD15):
    """vision/resnet.py:25-31 -- ResBlock.__call__."""
    g = cfg.num_groups
    h = ops.silu(ops.group_norm_affine(x, g, W[p + ".in_layers.0.weight"], W[p + ".in_layers.0.bias"]))
    h = ops.conv2d_bias(h, W[p + ".in_layers.2.weight"], W[p + ".in_layers.2.bias"], (1, 1))
    emb_out = ops.linear(ops.silu(emb), W[p + ".emb_layers.1.weight"], W[p + ".emb_layers.1.bias"])
    h = h + emb_out.reshape(*emb_out.shape, 1, 1)
    h = ops.silu(ops.group_norm_affine(h, g, W[p + ".out_layers.0.weight"], W[p + ".out_layers.0.bias"]))
    h = ops.conv2d_bias(h, W[p + ".out_layers.3.weight"], W[p + ".out_layers.3.bias"], (1, 1))
    if p + ".skip_connection.weight" in W:
        x = ops.conv2d_bias(x, W[p + ".skip_connection.weight"], W[p + ".skip_connection.bias"])
    return x + h


def cross_attention(x, context, W, p, n_heads, head_merge="reference_exact"):
    """attention/attention.py:35-41 -- CrossAttention.__call__.
    head_merge='reference_exact' reshapes the (b,h,t,d) SDPA output straight to (b,-1,h*d) with
    no transpose back (attention.py:38-39); 'intended' is the LDM head merge."""
    context = x if context is None else context
    q = ops.linear(x, W[p + ".to_q.weight"]); k = ops.linear(context, W[p + ".to_k.weight"]); v = ops.linear(context, W[p + ".to_v.weight"])
    b = x.shape[0]
    d = q.shape[-1] // n_heads
    q, k, v = [y.reshape(b, -1, n_heads, d).permute(0, 2, 1, 3) for y in (q, k, v)]
    o = ops.scaled_dot_product_attention(q, k, v)
    if head_merge == "reference_exact":
        o = o.reshape(b, -1, n_heads * d)
    else:
        o = o.permute(0, 2, 1, 3).reshape(b, -1, n_heads * d)
    return ops.linear(o, W[p + ".to_out.0.weight"], W[p + ".to_out.0.bias"])


def feed_forward(x, W, p):
    """ff/nn.py:22-23 -- FeedForward: GEGLU -> (dropout slot) -> Linear."""
    h = ops.geglu(x, W[p + ".net.0.proj.weight"], W[p + ".net.0.proj.bias"])
    return ops.linear(h, W[p + ".net.2.weight"], W[p + ".net.2.bias"])


def basic_transformer_block(x, context, W, p, n_heads, head_merge="reference_exact"):
    """attention/attention.py:52-56 -- BasicTransformerBlock.__call__."""
    x = cross_attention(ops.layer_norm(x, W[p + ".norm1.weight"], W[p + ".norm1.bias"]), None, W, p + ".attn1", n_heads, head_merge) + x
    x = cross_attention(ops.layer_norm(x, W[p + ".norm2.weight"], W[p + ".norm2.bias"]), context, W, p + ".attn2", n_heads, head_merge) + x
    x = feed_forward(ops.layer_norm(x, W[p + ".norm3.weight"], W[p + ".norm3.bias"]), W, p + ".ff") + x
    return x


def spatial_transformer(x, context, W, p, n_heads, cfg=SD15, head_merge="reference_exact"):
    """attention/attention.py:66-76 -- SpatialTransformer.__call__."""
    b, c, h, w = x.shape
    x_in = x
    x = ops.group_norm_affine(x, cfg.num_groups, W[p + ".norm.weight"], W[p + ".norm.bias"])
    x = ops.conv2d_bias(x, W[p + ".proj_in.weight"], W[p + ".proj_in.bias"])
    x = x.reshape(b, c, h * w).permute(0, 2, 1)
    x = basic_transformer_block(x, context, W, p + ".transformer_blocks.0", n_heads, head_merge)
    x = x.permute(0, 2, 1).reshape(b, c, h, w)
    return ops.conv2d_bias(x, W[p + ".proj_out.weight"], W[p + ".proj_out.bias"]) + x_in


def unet_forward(x, timesteps, context, W, cfg=SD15, head_merge="reference_exact", taps=None):
    """vision/unet.py:51-76 -- UNetModel.__call__.  W: dict name -> array (UNet-root-relative)."""
    W = {k: ops.as_t(v) for k, v in W.items()}
    x, context = ops.as_t(x), ops.as_t(context)
    t_emb = ops.timestep_embedding(timesteps, cfg.model_channels)
    emb = ops.linear(t_emb, W["time_embed.0.weight"], W["time_embed.0.bias"])
    emb = ops.linear(ops.silu(emb), W["time_embed.2.weight"], W["time_embed.2.bias"])
    inp, mid, out = _graph(cfg)

    def run(x, l, p):
        if l[0] == "conv": return ops.conv2d_bias(x, W[p + ".weight"], W[p + ".bias"], (1, 1))
        if l[0] == "res": return resblock(x, emb, W, p, cfg)
        if l[0] == "st": return spatial_transformer(x, context, W, p, cfg.n_heads, cfg, head_merge)
        if l[0] == "down": return ops.conv2d_bias(x, W[p + ".op.weight"], W[p + ".op.bias"], (1, 1), (2, 2))
        if l[0] == "up": return ops.conv2d_bias(ops.upsample_nearest2x(x), W[p + ".conv.weight"], W[p + ".conv.bias"], (1, 1))
        raise ValueError(l)

    saved = []
    for i, b in enumerate(inp):
        for j, l in enumerate(b):
            x = run(x, l, f"input_blocks.{i}.{j}")
        saved.append(x)
        if taps is not None: taps[f"input_blocks.{i}"] = x
    for j, l in enumerate(mid):
        x = run(x, l, f"middle_block.{j}")
    if taps is not None: taps["middle_block"] = x
    for i, b in enumerate(out):
        x = torch.cat((x, saved.pop()), dim=1)
        for j, l in enumerate(b):
            x = run(x, l, f"output_blocks.{i}.{j}")
        if taps is not None: taps[f"output_blocks.{i}"] = x
    x = ops.silu(ops.group_norm_affine(x, cfg.num_groups, W["out.0.weight"], W["out.0.bias"]))
    return ops.conv2d_bias(x, W["out.2.weight"], W["out.2.bias"], (1, 1))


def sd_step(unconditional_context, context, latent, timestep, a_t, a_prev, guidance, W, cfg=SD15,
            head_merge="reference_exact"):
    """variants/sd.py:56-59 -- StableDiffusion.__call__ = get_model_output (:27-46) + DDIM (:14-25).
    Generalised from the reference's batch-1 broadcast (:31) to [uncond x B ; cond x B]."""
    latent = ops.as_t(latent)
    x = torch.cat((latent, latent), dim=0)
    ctx = torch.cat((ops.as_t(unconditional_context), ops.as_t(context)), dim=0)
    latents = unet_forward(x, timestep, ctx, W, cfg, head_merge)
    e_t = ops.cfg_combine(latents, guidance)
    x_prev, _ = ops.get_x_prev_and_pred_x0(latent, e_t, a_t, a_prev)
    return x_prev


def sampler_schedule(steps):
    """example/sd1.py:54-57 -- timesteps = range(1,1000,1000//steps); alphas, alphas_prev."""
    timesteps = list(range(1, 1000, 1000 // steps))
    ac = ops.get_alphas_cumprod()
    alphas = ac[timesteps]
    alphas_prev = np.concatenate((np.array([1.0]), alphas[:-1])).astype(np.float32)
    return timesteps, alphas, alphas_prev
