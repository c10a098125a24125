"""The reference's vae/encoder.py: the VAE Encoder (:12-34; nothing on the sampler's path uses it -- AutoencoderKL.__call__ does) and the CLIP
text encoder (:36-81: CLIPEncoder, CLIPEncoderLayer, CLIPTextEmbeddings, CLIPTextTransformer).

Produces the (B, 77, 768) contexts the UNet's cross-attention reads (example/sd1.py:46-49).  Per layer: 5 launches of
the UNet's own kernels -- LN1 folded into the fused q|k|v GEMM, causal SDPA (d = 64, LDS-DMA kernel), out_proj + residual,
LN2 folded into fc1, quick-GELU, fc2 + residual."""
import numpy as np

from ..attention.attention import CLIPAttention
from ..ff.embedding import Embedding, embedding
from ..ff.layer_norm import LayerNorm
from ..ff.nn import CLIPMLP


ENCODER_WIDTHS = ((128, 128), (128, 256), (256, 512), (512, 512))     # (in, out) channels of encoder level 0 (image resolution) .. 3


class Encoder:
    """vae/encoder.py:12-34.  Attribute tree = checkpoint keys: ``conv_in``, ``down.<level>.block.<0..1>``, ``down.<level>.downsample.conv``
    (levels 0..2), ``mid``, ``norm_out``, ``conv_out``.  A level runs two ResnetBlocks and, except for the last, halves the resolution with a
    stride-2 3x3 conv whose padding is [0, 1, 0, 1] -- one pixel on the right and at the bottom only (the LDM form; the reference hands that
    list to its conv as ``padding``), here an explicit zero pad followed by an unpadded conv (vision/conv2d.py::pad_image).  The output has
    8 channels: means and log-variances of the latent."""

    def __init__(self, init=True):
        from ..ff.group_norm import GroupNorm
        from ..vision.conv2d import Conv2d
        from ..vision.resnet import ResnetBlock
        from .mid import Mid
        self.conv_in = Conv2d(3, 128, kernel_size=[3, 3], padding=[1, 1], init=init)
        self.down = []
        for i, (cin, cout) in enumerate(ENCODER_WIDTHS):
            level = {"block": [ResnetBlock(cin, cout, init=init), ResnetBlock(cout, cout, init=init)]}
            if i != len(ENCODER_WIDTHS) - 1:
                level["downsample"] = {"conv": Conv2d(cout, cout, kernel_size=[3, 3], stride=[2, 2], padding=[0, 1, 0, 1], init=init)}
            self.down.append(level)
        top = ENCODER_WIDTHS[-1][1]
        self.mid = Mid(top, init=init)
        self.norm_out = GroupNorm(32, top, init=init)
        self.conv_out = Conv2d(top, 8, kernel_size=[3, 3], padding=[1, 1], init=init)

    def __call__(self, x):
        x = self.conv_in(x)
        for level in self.down:
            for block in level["block"]:
                x = block(x)
            if "downsample" in level:
                x = level["downsample"]["conv"](x)
        x = self.mid(x)
        return self.conv_out(self.norm_out(x, silu=True))


class CLIPEncoderLayer:
    def __init__(self, init=True):
        self.self_attn = CLIPAttention(init=init)
        self.layer_norm1 = LayerNorm(768, init=init)
        self.mlp = CLIPMLP(init=init)
        self.layer_norm2 = LayerNorm(768, init=init)

    def __call__(self, hidden_states, causal_attention_mask):
        hidden_states = self.self_attn(hidden_states, causal_attention_mask, residual=hidden_states, ln=self.layer_norm1)
        return self.mlp(hidden_states, residual=hidden_states, ln=self.layer_norm2)


class CLIPEncoder:
    def __init__(self, init=True):
        self.layers = [CLIPEncoderLayer(init=init) for i in range(12)]

    def __call__(self, hidden_states, causal_attention_mask):
        for l in self.layers:
            hidden_states = l(hidden_states, causal_attention_mask)
        return hidden_states


class CLIPTextEmbeddings:
    def __init__(self, init=True):
        self.token_embedding = Embedding(49408, 768, init=init)
        self.position_embedding = Embedding(77, 768, init=init)

    def __call__(self, input_ids, position_ids=None):
        """position_ids: None or arange(T) (all the reference passes, vae/encoder.py:78): row i is added to token i."""
        if position_ids is not None:
            p = np.asarray(position_ids).reshape(-1)
            assert (p == np.arange(p.size)).all(), "position_ids must be arange(T)"
        return embedding(self.token_embedding.weight, input_ids, self.position_embedding.weight)


class CLIPTextTransformer:
    def __init__(self, init=True):
        self.embeddings = CLIPTextEmbeddings(init=init)
        self.encoder = CLIPEncoder(init=init)
        self.final_layer_norm = LayerNorm(768, init=init)

    def __call__(self, input_ids):
        """input_ids: (B, T <= 77) integer token ids (host array or int32 DeviceArray) -> (B, T, 768) f16 context."""
        t = input_ids.shape[1]
        x = self.embeddings(input_ids)
        mask = np.triu(np.full((1, 1, t, t), float("-inf"), dtype=np.float32), k=1)       # vae/encoder.py:79
        x = self.encoder(x, mask)
        return self.final_layer_norm(x)
