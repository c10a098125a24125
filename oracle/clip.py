"""CPU restatement of the reference's CLIP text encoder: CLIPTextTransformer and its parts.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity status: the encoder stack (12 x CLIPEncoderLayer + final
LayerNorm) is PINNED against the reference's own classes run under the stub import (tests/golden/make_golden.py
``clip`` -> tests/golden/clip_text.npz).  The reference's Embedding (ff/embedding.py:10-24) is defective (SURVEY D7:
one-hot matrix with swapped axes pushed through a cuBLAS wrapper); its intended semantics -- a row gather, the
Hugging Face CLIPTextEmbeddings the class tree is modelled on -- is restated here and checked against
``transformers.CLIPTextModel`` on identical weights in tests/test_oracle_golden.py.
Weight names are the LDM checkpoint keys the reference's update_state walk produces from the StableDiffusion root
(``cond_stage_model.transformer.text_model...``, variants/sd.py:12)."""
import numpy as np
import torch

from . import ops

__all__ = ["clip_param_shapes", "clip_attention", "clip_mlp", "clip_encoder_layer", "clip_encoder", "clip_text_transformer", "CLIP_TEXT"]

# vae/encoder.py:49-81, attention/attention.py:78-86, ff/nn.py:25-28: every size is hard-coded in the reference
CLIP_TEXT = dict(vocab=49408, positions=77, dim=768, heads=12, layers=12, mlp=3072)


def clip_param_shapes(prefix="cond_stage_model.transformer.text_model.", cfg=CLIP_TEXT):
    d, P = cfg["dim"], {}
    P[prefix + "embeddings.token_embedding.weight"] = (cfg["vocab"], d)
    P[prefix + "embeddings.position_embedding.weight"] = (cfg["positions"], d)
    for i in range(cfg["layers"]):
        l = f"{prefix}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            P[l + f"self_attn.{n}.weight"] = (d, d); P[l + f"self_attn.{n}.bias"] = (d,)
        for n in ("layer_norm1", "layer_norm2"):
            P[l + n + ".weight"] = (d,); P[l + n + ".bias"] = (d,)
        P[l + "mlp.fc1.weight"] = (cfg["mlp"], d); P[l + "mlp.fc1.bias"] = (cfg["mlp"],)
        P[l + "mlp.fc2.weight"] = (d, cfg["mlp"]); P[l + "mlp.fc2.bias"] = (d,)
    P[prefix + "final_layer_norm.weight"] = (d,); P[prefix + "final_layer_norm.bias"] = (d,)
    return P


def causal_mask(t):
    """vae/encoder.py:79 -- additive (1,1,T,T) mask, -inf strictly above the diagonal."""
    return torch.triu(torch.full((1, 1, t, t), float("-inf")), diagonal=1)


def clip_attention(x, W, p, mask, heads):
    """attention/attention.py:88-104 -- q/k/v/out Linear with bias, heads split (b,t,h,d)->(b,h,t,d), masked SDPA,
    heads merged back WITH the transpose (:100; unlike CrossAttention's D11)."""
    b, t, d = x.shape
    q, k, v = (ops.linear(x, W[p + f"{n}.weight"], W[p + f"{n}.bias"]) for n in ("q_proj", "k_proj", "v_proj"))
    q, k, v = (y.reshape(b, t, heads, d // heads).permute(0, 2, 1, 3) for y in (q, k, v))
    o = ops.scaled_dot_product_attention(q, k, v, mask)
    o = o.permute(0, 2, 1, 3).reshape(b, t, d)
    return ops.linear(o, W[p + "out_proj.weight"], W[p + "out_proj.bias"])


def clip_mlp(x, W, p):
    """ff/nn.py:25-34 -- fc1 -> quick_gelu (storage/tensor.py:77-78) -> fc2."""
    h = ops.linear(x, W[p + "fc1.weight"], W[p + "fc1.bias"])
    return ops.linear(ops.quick_gelu(h), W[p + "fc2.weight"], W[p + "fc2.bias"])


def clip_encoder_layer(x, W, p, mask, heads):
    """vae/encoder.py:49-66 -- pre-LN attention and MLP, both residual."""
    x = x + clip_attention(ops.layer_norm(x, W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"]), W, p + "self_attn.", mask, heads)
    return x + clip_mlp(ops.layer_norm(x, W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"]), W, p + "mlp.")


def clip_encoder(x, W, prefix, mask, cfg=CLIP_TEXT):
    """vae/encoder.py:39-47."""
    x = ops.as_t(x)
    for i in range(cfg["layers"]):
        x = clip_encoder_layer(x, W, f"{prefix}encoder.layers.{i}.", mask, cfg["heads"])
    return x


def clip_text_transformer(input_ids, W, prefix="cond_stage_model.transformer.text_model.", cfg=CLIP_TEXT):
    """vae/encoder.py:68-81 -- token + position embedding (row gathers), 12 layers under the causal mask, final LN.
    input_ids: (B, T) integers, T <= 77."""
    ids = torch.as_tensor(np.asarray(input_ids), dtype=torch.long)
    b, t = ids.shape
    tok = ops.as_t(W[prefix + "embeddings.token_embedding.weight"])[ids]            # ff/embedding.py: intended gather
    pos = ops.as_t(W[prefix + "embeddings.position_embedding.weight"])[torch.arange(t)]
    x = clip_encoder(tok + pos[None], W, prefix, causal_mask(t), cfg)
    return ops.layer_norm(x, W[prefix + "final_layer_norm.weight"], W[prefix + "final_layer_norm.bias"])
