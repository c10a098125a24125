"""CLIP text encoder -- mirrors tinyfusers/vae/encoder.py:36-81 (CLIPEncoder, CLIPEncoderLayer, CLIPTextEmbeddings,
CLIPTextTransformer; the file also holds the VAE Encoder, which nothing on the sampler's path uses and which is not built).

Produces the (B, 77, 768) contexts the UNet's cross-attention reads (example/sd1.py:46-49).  Per layer: 5 launches of
the UNet's own kernels -- LN1 folded into the fused q|k|v GEMM, causal SDPA (d = 64, LDS-DMA kernel), out_proj + residual,
LN2 folded into fc1, quick-GELU, fc2 + residual."""
import numpy as np

from ..attention.attention import CLIPAttention
from ..ff.embedding import Embedding, embedding
from ..ff.layer_norm import LayerNorm
from ..ff.nn import CLIPMLP


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
