"""Checkpoint reader -- mirrors tinyfusers/storage/unpicker.py:15-86 (``load_weights`` of a torch-zip ``.ckpt``), plus
safetensors as the alternative on-disk format.  Pure host code, no torch import, nothing executed from
the file: the unpickler resolves a closed list of globals (the reference's list, :55-73, extended by the storage types
and rebuild helpers newer torch versions write) and maps everything else to an inert placeholder.

Differences from the reference, all in the direction of the intended behaviour:
  * storages are memory-mapped in place (torch writes them uncompressed, 64-byte aligned) instead of being read into a
    dict of memoryviews and expanded through ``.tolist()`` -- a 4 GB checkpoint costs no host RAM until a tensor is
    touched, and ``update_state`` streams each tensor to the device as it walks;
  * dtypes are honoured (the reference casts HalfStorage bytes as 'f', :36, which mis-reads fp16 checkpoints);
    bfloat16 comes back as float32 (numpy has no bf16);
  * storage offset and strides are honoured (the reference reshapes the whole storage, :26).
``load_weights`` returns the unpickled object (for an LDM checkpoint a dict with a 'state_dict' entry), tensors as
numpy arrays, like the reference."""
import collections
import io
import json
import pickle
import struct
import zipfile

import numpy as np

__all__ = ["load_weights", "load_safetensors", "load_checkpoint", "save_safetensors"]

_STORAGE_DTYPES = {
    "FloatStorage": np.dtype("<f4"), "HalfStorage": np.dtype("<f2"), "DoubleStorage": np.dtype("<f8"),
    "BFloat16Storage": "bf16", "LongStorage": np.dtype("<i8"), "IntStorage": np.dtype("<i4"), "ShortStorage": np.dtype("<i2"),
    "CharStorage": np.dtype("i1"), "ByteStorage": np.dtype("u1"), "BoolStorage": np.dtype("?"), "UntypedStorage": np.dtype("u1"),
}


def _bf16_to_f32(u16):
    return (np.asarray(u16, dtype=np.uint32) << np.uint32(16)).view(np.float32)


class _Opaque:
    """Stands in for any global the reader does not know (lightning callbacks, optimizer classes ...): accepts any
    construction / state and does nothing.  The reference returns the string "model_checkpoint" for the one it knows (:70-71)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __setstate__(self, state):
        pass

    def __reduce__(self):
        return (_Opaque, ())


class _Storage:
    __slots__ = ("dtype", "key", "numel")

    def __init__(self, dtype, key, numel):
        self.dtype, self.key, self.numel = dtype, key, numel


class _ZipData:
    """Maps ``<base>/data/<key>`` entries of a torch zip to numpy arrays over one shared read-only memory map."""

    def __init__(self, path):
        self.path = path
        self.zf = zipfile.ZipFile(path, "r")
        names = self.zf.namelist()
        self.base = names[0].split("/", 1)[0]
        self.mm = None

    def _entry(self, key):
        zi = self.zf.getinfo(f"{self.base}/data/{key}")
        if zi.compress_type != zipfile.ZIP_STORED:
            return np.frombuffer(self.zf.read(zi), dtype=np.uint8)
        if self.mm is None:
            self.mm = np.memmap(self.path, dtype=np.uint8, mode="r")
        # local file header: 30 fixed bytes, then name and extra field whose lengths are stored at offsets 26 and 28
        nlen, elen = struct.unpack("<HH", bytes(self.mm[zi.header_offset + 26: zi.header_offset + 30]))
        start = zi.header_offset + 30 + nlen + elen
        return self.mm[start: start + zi.file_size]

    def array(self, st: _Storage):
        raw = self._entry(st.key)
        if st.dtype == "bf16":
            return raw[: st.numel * 2].view(np.uint16), True
        dt = st.dtype
        return raw[: st.numel * dt.itemsize].view(dt), False


def _make_unpickler(data: _ZipData):
    def rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
        flat, is_bf16 = data.array(storage)
        size, stride = tuple(int(v) for v in size), tuple(int(v) for v in stride)
        storage_offset = int(storage_offset)
        # the view must stay inside the storage's bytes as they exist in the FILE: offset, sizes and strides come from the pickle
        # (untrusted) and go to as_strided, which checks nothing.  The reference's np.reshape would have raised (storage/unpicker.py:38).
        if flat.size < storage.numel:
            raise pickle.UnpicklingError(f"storage {storage.key}: the zip entry holds {flat.size} elements, the pickle claims {storage.numel}")
        if len(size) != len(stride) or storage_offset < 0 or any(v < 0 for v in size) or any(v < 0 for v in stride):
            raise pickle.UnpicklingError(f"storage {storage.key}: bad tensor geometry offset={storage_offset} size={size} stride={stride}")
        if all(v > 0 for v in size):
            last = storage_offset + sum((n - 1) * st for n, st in zip(size, stride))
            if last >= flat.size:
                raise pickle.UnpicklingError(f"storage {storage.key}: tensor view (offset {storage_offset}, size {size}, stride {stride}) "
                                             f"reaches element {last} of a {flat.size}-element storage")
        if len(size) == 0:
            t = flat[storage_offset: storage_offset + 1].reshape(())
        else:
            t = np.lib.stride_tricks.as_strided(flat[storage_offset:], shape=size, strides=tuple(s * flat.itemsize for s in stride),
                                                writeable=False)
        return _bf16_to_f32(t) if is_bf16 else t

    def rebuild_parameter(data_, requires_grad=False, backward_hooks=None, *a):
        return data_

    class TorchUnpickler(pickle.Unpickler):
        def persistent_load(self, saved_id):
            assert saved_id[0] == "storage", saved_id[0]
            _, type_class, key, _location, numel = saved_id[:5]
            return _Storage(type_class, str(key), int(numel))

        def find_class(self, module, name):
            if module == "collections" and name == "OrderedDict":
                return collections.OrderedDict
            if module == "torch._utils" and name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
                return rebuild_tensor_v2
            if module == "torch._utils" and name in ("_rebuild_parameter", "_rebuild_parameter_with_state"):
                return rebuild_parameter
            if module in ("torch", "torch.storage") and name in _STORAGE_DTYPES:
                return _STORAGE_DTYPES[name]
            if module == "torch" and name in ("Size",):
                return tuple
            if module in ("numpy.core.multiarray", "numpy._core.multiarray") and name == "scalar":
                return np.core.multiarray.scalar if hasattr(np, "core") else np._core.multiarray.scalar
            if module == "numpy" and name == "dtype":
                return np.dtype
            if module == "_codecs" and name == "encode":
                import _codecs
                return _codecs.encode
            if module == "builtins" and name in ("set", "frozenset", "dict", "list", "tuple", "int", "float", "str", "bool", "complex", "slice"):
                return getattr(__import__("builtins"), name)
            return _Opaque           # never import or run anything named by the file
    return TorchUnpickler


def load_weights(weight_path):
    """storage/unpicker.py:75-86: the object pickled in a torch-zip checkpoint, tensors as (memory-mapped) numpy arrays."""
    if not zipfile.is_zipfile(weight_path):
        raise NameError(f"File format not supported: {weight_path}")       # same exception as the reference (:86)
    data = _ZipData(weight_path)
    with data.zf.open(f"{data.base}/data.pkl") as f:
        return _make_unpickler(data)(io.BytesIO(f.read())).load()


_ST_DTYPES = {"F32": np.dtype("<f4"), "F16": np.dtype("<f2"), "F64": np.dtype("<f8"), "BF16": "bf16", "I64": np.dtype("<i8"),
              "I32": np.dtype("<i4"), "I16": np.dtype("<i2"), "I8": np.dtype("i1"), "U8": np.dtype("u1"), "BOOL": np.dtype("?")}


def load_safetensors(path):
    """name -> numpy array (memory-mapped) of a .safetensors file: u64 header length, JSON header, raw little-endian data."""
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    if mm.size < 8:
        raise NameError(f"File format not supported: {path}")
    (n,) = struct.unpack("<Q", bytes(mm[:8]))
    if n > mm.size - 8:
        raise NameError(f"File format not supported: {path}")
    header = json.loads(bytes(mm[8: 8 + n]).decode("utf-8"))
    out = {}
    for name, info in header.items():
        if name == "__metadata__":
            continue
        dt = _ST_DTYPES[info["dtype"]]
        b, e = info["data_offsets"]
        raw = mm[8 + n + b: 8 + n + e]
        shape = tuple(info["shape"])
        if dt == "bf16":
            out[name] = _bf16_to_f32(raw.view(np.uint16).reshape(shape))
        else:
            out[name] = raw.view(dt).reshape(shape)
    return out


def save_safetensors(path, tensors, metadata=None):
    """Write name -> numpy array as .safetensors (fp16 / fp32 / ints); the packed-arena export of a loaded model."""
    inv = {v: k for k, v in _ST_DTYPES.items() if v != "bf16"}
    header, off, blobs = {}, 0, []
    for name in sorted(tensors):
        a = np.asarray(tensors[name])
        if not a.flags.c_contiguous:
            a = a.copy()                                # (np.ascontiguousarray would turn a 0-d array into shape (1,))
        dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
        code = inv[np.dtype(dt.str.replace("=", "<")) if dt.itemsize > 1 else dt]
        header[name] = {"dtype": code, "shape": list(a.shape), "data_offsets": [off, off + a.nbytes]}
        off += a.nbytes
        blobs.append(a)
    if metadata:
        header["__metadata__"] = {str(k): str(v) for k, v in metadata.items()}
    hj = json.dumps(header, separators=(",", ":")).encode("utf-8")
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj))); f.write(hj)
        for a in blobs:
            f.write(a.tobytes())


def load_checkpoint(path):
    """Flat LDM state dict (name -> numpy array) from a torch-zip ``.ckpt`` / ``.pt`` or a ``.safetensors`` file."""
    if zipfile.is_zipfile(path):
        obj = load_weights(path)
        return obj["state_dict"] if isinstance(obj, dict) and "state_dict" in obj else obj
    return load_safetensors(path)
