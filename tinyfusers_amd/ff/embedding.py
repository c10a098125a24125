"""embedding / Embedding -- mirrors tinyfusers/ff/embedding.py:6-24.  The reference builds a one-hot matrix on the
host and pushes it through its cuBLAS wrapper (and gets the axes wrong: SURVEY D7); the intended result -- row
``idx[b, i]`` of ``weight`` for every token -- is one gather launch here (tf_embedding_f16)."""
import numpy as np

from ..native import hip
from ..storage.tensor import DeviceArray, _sh, asarray


def _ids_to_device(idx, vocab):
    if isinstance(idx, DeviceArray):
        assert idx.dtype == np.int32, "token ids on the device must be int32"
        return idx
    ids = np.asarray(idx)
    if ids.dtype.kind == "f":                       # the reference passes float32 position ids (vae/encoder.py:78)
        assert (ids == np.round(ids)).all(), "token ids must be integers"
    ids = ids.astype(np.int64)
    if ids.size and (ids.min() < 0 or ids.max() >= vocab):
        raise IndexError(f"embedding index out of range [0, {vocab}): min {ids.min()} max {ids.max()}")
    return DeviceArray.from_numpy(ids.astype(np.int32), np.int32, "row")


def embedding(weight, idx, pos_weight=None):
    """weight (V, D) f16, idx (B, N) integers (host array or int32 DeviceArray) -> (B, N, D);
    pos_weight (T, D): also add row ``i`` to token ``i`` of every sequence (CLIPTextEmbeddings, vae/encoder.py:72-73)."""
    vocab, dim = weight.shape
    ids = _ids_to_device(idx, vocab)
    b, n = ids.shape
    if pos_weight is not None:
        assert pos_weight.shape[0] >= n and pos_weight.shape[1] == dim
    out = DeviceArray.empty((b, n, dim), np.float16, "row")
    hip.tf_embedding_f16(out.ptr, weight.ptr, ids.ptr, pos_weight.ptr if pos_weight is not None else None, b * n, dim, vocab, n, _sh())
    out._base = ids
    return out


class Embedding:
    def __init__(self, vocab_size: int, embed_size: int, init=True):
        self.vocab_sz = vocab_size
        self.embed_sz = embed_size
        self.weight = asarray(np.ones((vocab_size, embed_size), dtype=np.float16)) if init else None   # ff/embedding.py:14

    def __call__(self, idx):
        return embedding(self.weight, idx)
