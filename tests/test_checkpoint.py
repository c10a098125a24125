"""the checkpoint reader (torch-zip ``.ckpt`` as storage/unpicker.py:75-86 reads it, and safetensors) must hand
update_state exactly the tensors torch / safetensors wrote -- bit for bit."""
import collections
import os

import numpy as np
import pytest
import torch

from tinyfusers_amd.storage.unpicker import load_checkpoint, load_safetensors, load_weights, save_safetensors


def _state():
    g = torch.Generator().manual_seed(3)
    base = torch.randn(6, 10, generator=g)
    sd = collections.OrderedDict()
    sd["model.diffusion_model.input_blocks.0.0.weight"] = torch.randn(8, 4, 3, 3, generator=g)
    sd["model.diffusion_model.input_blocks.0.0.bias"] = torch.randn(8, generator=g).half()
    sd["first_stage_model.decoder.conv_in.weight"] = torch.randn(4, 4, 1, 1, generator=g).to(torch.bfloat16)
    sd["view.transposed"] = base.t()                         # non-contiguous strides
    sd["view.offset"] = base[2:5, 3:7]                       # storage offset + strides, shares storage with the above
    sd["scalar"] = torch.tensor(3.5)
    sd["cond_stage_model.transformer.text_model.embeddings.position_ids"] = torch.arange(77).reshape(1, 77)
    sd["ints32"] = torch.arange(-3, 5, dtype=torch.int32)
    sd["empty"] = torch.zeros(0, 3)
    return sd


def _same(got, want):
    want_np = want.float().numpy() if want.dtype == torch.bfloat16 else want.numpy()
    assert tuple(got.shape) == tuple(want_np.shape), (got.shape, want_np.shape)
    assert got.dtype == want_np.dtype, (got.dtype, want_np.dtype)
    assert np.array_equal(np.asarray(got), want_np)


def test_torch_zip_checkpoint_roundtrip(tmp_path):
    sd = _state()
    p = os.path.join(tmp_path, "toy.ckpt")
    # the extra entries mimic a lightning checkpoint: plain python values and an object of a class the reader must not import
    torch.save({"state_dict": sd, "epoch": 7, "global_step": 1234, "callbacks": {"k": torch.optim.SGD}}, p)
    obj = load_weights(p)
    assert obj["epoch"] == 7 and obj["global_step"] == 1234
    assert list(obj["state_dict"].keys()) == list(sd.keys())
    for k, v in sd.items():
        _same(obj["state_dict"][k], v)
    flat = load_checkpoint(p)
    assert set(flat) == set(sd)


def test_torch_zip_nn_module_state_dict(tmp_path):
    m = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3), torch.nn.GroupNorm(2, 8), torch.nn.Linear(8, 5, bias=False)).half()
    p = os.path.join(tmp_path, "m.pt")
    torch.save(m.state_dict(), p)
    got = load_checkpoint(p)
    for k, v in m.state_dict().items():
        _same(got[k], v)


def test_unsupported_format_raises_like_the_reference(tmp_path):
    p = os.path.join(tmp_path, "legacy.bin")
    with open(p, "wb") as f:
        f.write(b"\x80\x02not a zip")
    with pytest.raises(NameError):
        load_weights(p)


def test_safetensors_roundtrip(tmp_path):
    st = pytest.importorskip("safetensors.torch")
    sd = {k: v.contiguous() for k, v in _state().items() if v.numel() > 0}
    p = os.path.join(tmp_path, "toy.safetensors")
    st.save_file(sd, p, metadata={"format": "pt"})
    got = load_safetensors(p)
    assert set(got) == set(sd)
    for k, v in sd.items():
        _same(got[k], v)
    # and our writer is readable by the safetensors library
    q = os.path.join(tmp_path, "ours.safetensors")
    mine = {k: np.asarray(v) for k, v in got.items()}
    save_safetensors(q, mine, {"source": "tinyfusers_amd"})
    back = st.load_file(q)
    for k, v in mine.items():
        assert np.array_equal(back[k].numpy(), v), k
    assert set(load_checkpoint(q)) == set(mine)


def test_update_state_key_walk_matches_checkpoint_names():
    """The attribute walk of update_state (storage/state.py:4-23) over StableDiffusion asks the checkpoint for exactly the
    LDM names the oracle's shape tables list: UNet (686), VAE decoder, CLIP text encoder -- plus the bias-free
    to_q/to_k/to_v probes, which the reference's walk makes too (state.py:17-19).  No GPU: nothing is uploaded."""
    import contextlib
    import io
    import oracle
    from tinyfusers_amd.storage.state import update_state
    from tinyfusers_amd.variants.sd import StableDiffusion

    class Recorder(dict):
        def __init__(self):
            super().__init__()
            self.seen = []

        def __contains__(self, k):
            self.seen.append(k)
            return False
    rec = Recorder()
    with contextlib.redirect_stdout(io.StringIO()):
        update_state(StableDiffusion(init=False), rec)
    seen = set(rec.seen)
    assert len(seen) == len(rec.seen), "a leaf was visited twice"
    want = {"model.diffusion_model." + k for k in oracle.unet_param_shapes(oracle.SD15)}
    want |= set(oracle.vae_decoder_param_shapes()) | set(oracle.vae_encoder_param_shapes()) | set(oracle.clip_param_shapes())
    assert len(oracle.clip_param_shapes()) == 2 + 12 * 16 + 2
    extra = {k for k in seen - want if not k.endswith((".to_q.bias", ".to_k.bias", ".to_v.bias"))}
    assert not extra, sorted(extra)[:5]
    assert not (want - seen), sorted(want - seen)[:5]
    # the package's own shape enumerator (what example/sd1.py synthesises weights from) agrees name for name, shape for shape
    from tinyfusers_amd.storage.state import param_shapes
    shapes = {"model.diffusion_model." + k: tuple(v) for k, v in oracle.unet_param_shapes(oracle.SD15).items()}
    shapes.update({k: tuple(v) for k, v in oracle.vae_decoder_param_shapes().items()})
    shapes.update({k: tuple(v) for k, v in oracle.vae_encoder_param_shapes().items()})
    shapes.update({k: tuple(v) for k, v in oracle.clip_param_shapes().items()})
    assert param_shapes(StableDiffusion(init=False)) == shapes


def _malformed_zip(path, size, stride, offset=0, numel=4, payload=16):
    """A torch-zip whose pickle describes a tensor view reaching outside its 4-element storage (what torch.save never writes)."""
    import io
    import pickle
    import zipfile

    class Ref:
        pass

    class Evil:
        def __reduce__(self):
            return (torch._utils._rebuild_tensor_v2, (Ref(), offset, tuple(size), tuple(stride), False, collections.OrderedDict()))

    class P(pickle.Pickler):
        def persistent_id(self, obj):
            if isinstance(obj, Ref):
                return ("storage", torch.FloatStorage, "0", "cpu", numel)
            return None
    buf = io.BytesIO()
    P(buf, protocol=2).dump({"state_dict": {"w": Evil()}})
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as zf:
        zf.writestr("archive/data.pkl", buf.getvalue())
        zf.writestr("archive/data/0", b"\x00" * payload)
        zf.writestr("archive/version", "3\n")


@pytest.mark.parametrize("size,stride,offset,numel,payload", [
    ((65536,), (4096,), 0, 4, 16),        # the first touch of this view used to segfault
    ((4,), (2,), 0, 4, 16),               # small stride: reads past the storage, inside the file
    ((2, 2), (2, 1), 1, 4, 16),           # offset pushes the last element out
    ((4,), (1,), -1, 4, 16),              # negative offset
    ((4,), (-1,), 3, 4, 16),              # negative stride
    ((4,), (1,), 0, 8, 16),               # the pickle claims more elements than the zip entry holds
])
def test_malformed_tensor_geometry_is_refused(tmp_path, size, stride, offset, numel, payload):
    import pickle
    p = str(tmp_path / "evil.ckpt")
    _malformed_zip(p, size, stride, offset, numel, payload)
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint(p)


def test_inbounds_strided_view_still_loads(tmp_path):
    p = str(tmp_path / "ok.ckpt")
    _malformed_zip(p, (2, 2), (1, 2), 0, 4, 16)       # a transposed 2x2 view of the 4-element storage: legal
    w = load_checkpoint(p)["w"]
    assert w.shape == (2, 2) and float(np.asarray(w).sum()) == 0.0
