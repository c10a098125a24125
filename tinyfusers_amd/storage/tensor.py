"""Device arrays and activation statics.

Mirrors tinyfusers/storage/tensor.py: the reference's ``Tensor`` is (a) a cudaMalloc/cudaMemcpy handle
holder (:10-62) and (b) a bag of activation statics used as free functions over CuPy arrays (:64-86).
Here the CuPy ndarray's role is played by ``DeviceArray`` -- a thin (pointer, logical shape, dtype,
layout) handle over HIP memory obtained through the C-ABI (tf_malloc / tf_memcpy / tf_free).

Layout: activations are fp16.  A 4-D array keeps the reference's *logical* NCHW shape but is stored
channels-last (NHWC); (B, T, C) token arrays are plain row-major, so the NCHW<->(b,hw,c) transposes of
attention/attention.py:71,74 are free re-views.  Weights of Conv2d keep logical (K,C,R,S) and are stored
"KRSC" by the very same rule.
"""
import ctypes
import functools
import math
import weakref

import numpy as np

from ..native import hip

H2D, D2H, D2D = 1, 2, 3
_ALIGN = 256


# ------------------------------------------------------------------------------------------------
# streams and allocation
class Stream:
    """An explicit HIP stream (the reference only ever uses the NULL stream, storage/device.py:98)."""

    def __init__(self, handle=None):
        if handle is None:
            h = ctypes.c_void_p()
            hip.tf_stream_create(ctypes.byref(h))
            self.handle, self._own = h, True
        else:
            self.handle, self._own = handle, False

    def synchronize(self):
        hip.tf_stream_sync(self.handle)

    def __del__(self):
        if getattr(self, "_own", False) and self.handle:
            try:
                hip.tf_stream_destroy(self.handle)
            except Exception:
                pass


_NULL_STREAM = Stream(ctypes.c_void_p(None))
_stream_stack = [_NULL_STREAM]
_initialised = [False]


def ensure_init(device=None):
    if not _initialised[0] or device is not None:
        hip.tf_init(int(device or 0))
        _initialised[0] = True


def current_stream():
    return _stream_stack[-1]


def _order(after, before):
    """Everything launched on ``after`` from now on runs after everything already launched on ``before``."""
    if after is before:
        return
    e = Branch._event()
    hip.tf_event_record(e, before.handle)
    hip.tf_stream_wait_event(after.handle, e)
    Branch._events.append(e)


class use_stream:
    """``with use_stream(s):`` routes every op launched inside onto stream ``s``.

    The switch keeps program order: ``s`` first waits for what was queued on the stream it takes over from, and that
    stream waits for ``s`` on exit.  The streams are non-blocking HIP streams (no implicit ordering with the NULL stream),
    and the pool hands a block freed on one stream to the next allocation on any stream, so without the two edges an op
    on ``s`` could read an input, or be given a block, that work queued before the ``with`` has not finished with.
    ``ordered=False`` skips the edges (a caller that re-enters the same stream back to back, e.g. the sampler loop)."""

    def __init__(self, s, ordered=True): self.s, self.ordered = s, ordered

    def __enter__(self):
        if self.ordered:
            _order(self.s, _stream_stack[-1])
        _stream_stack.append(self.s)
        return self.s

    def __exit__(self, *a):
        _stream_stack.pop()
        if self.ordered:
            _order(_stream_stack[-1], self.s)


def _sh():
    return _stream_stack[-1].handle


import os as _os
_POISON = _os.environ.get("TF_POOL_POISON", "0")   # debugging: fill freed ("release") / fresh ("fresh") / both ("1") blocks with 0xFF (NaN in fp16 / fp32)
_POISON_FRESH, _POISON_RELEASE = _POISON in ("1", "fresh"), _POISON in ("1", "release")


class Pool:
    """Stream-ordered caching allocator: freed blocks are re-used (most recently freed first, so the
    re-used block is still hot in L2 / Infinity Cache) instead of returned to the driver.  Allocation
    order is deterministic, which is what makes whole-step HIP-graph capture of the eager op stream
    possible: no hipMalloc/hipFree ever happens inside a captured region once the pool is warm."""

    def __init__(self):
        self.free_blocks = {}
        self.allocated = 0
        self.frozen = False      # True while capturing a graph: growing the pool would call hipMalloc
        self.held = None         # list of (ptr, n) released while a Branch is recording (see Branch)
        self.capture = None      # size -> [ptr] private free lists while a graph is being captured (see begin_capture)
        self.owned = {}          # ptr -> size of every block that belongs to a captured graph

    def alloc(self, nbytes):
        n = max(_ALIGN, (int(nbytes) + _ALIGN - 1) // _ALIGN * _ALIGN)
        if self.capture is not None:
            lst = self.capture.get(n)
            if lst:
                return lst.pop(), n
        lst = self.free_blocks.get(n)
        if lst:
            ptr = lst.pop()
            if self.capture is not None:
                self.owned[ptr] = n
            return ptr, n
        if self.frozen:
            raise RuntimeError(f"Pool: allocation of {n} B while frozen (graph capture) -- warm up with one eager step first")
        ensure_init()
        p = ctypes.c_void_p()
        hip.tf_malloc(ctypes.byref(p), n)
        self.allocated += n
        if _POISON_FRESH:
            hip.tf_memset_async(p, 0xFF, n, None)
            hip.tf_device_sync()
        return p.value, n

    def release(self, ptr, n):
        if _POISON_RELEASE:
            hip.tf_memset_async(ptr, 0xFF, n, _sh())     # TF_POOL_POISON=1: a freed block reads back as NaN (stream-ordered)
        if self.held is not None:
            self.held.append((ptr, n))       # a side branch is open: its temporaries stay out of the pool until the join
            return
        if self.capture is not None:
            self.owned[ptr] = n              # freed inside the capture: re-usable by the capture only, the graph keeps it
            self.capture.setdefault(n, []).append(ptr)
            return
        if ptr in self.owned:
            return                           # a captured graph replays into this block: never hand it to anybody else
        self.free_blocks.setdefault(n, []).append(ptr)

    # -- graph capture: every block the captured program touches becomes the graph's own.  Without this a block that the
    # program freed (in Python's eyes) is handed to the next eager allocation -- e.g. a weight that a later eager call
    # packs and caches -- and the next replay of the graph writes its activations over it.
    def begin_capture(self):
        assert self.capture is None, "nested graph capture"
        self.capture = {}
        self._owned_before = set(self.owned)
        self.frozen = True

    def end_capture(self):
        """-> the blocks {ptr: size} owned by the graph just captured (hand them back with disown when it is destroyed)."""
        self.capture = None
        self.frozen = False
        return {p: n for p, n in self.owned.items() if p not in self._owned_before}

    def disown(self, blocks):
        for ptr, n in blocks.items():
            if self.owned.pop(ptr, None) is not None:
                self.free_blocks.setdefault(n, []).append(ptr)


_pool = Pool()


class Branch:
    """Fork / join of an independent chain of ops onto a side stream (a parallel branch of the step graph).

        br = Branch()
        with br:
            skip = conv1x1(x)        # launched on the side stream, after everything already queued on the current one
        ...                          # the caller keeps launching on the current stream: runs concurrently
        br.join()                    # the current stream waits for the branch; use ``skip`` afterwards

    Stream-ordered allocation stays valid: blocks the branch takes from the pool were freed in program order before the
    fork (the side stream waits for the fork event); blocks the branch frees are held back until ``join`` so that nothing
    queued on the main stream in between can be handed memory the branch is still using; the branch's INPUTS must be kept
    referenced by the caller until ``join``.  Works inside hipGraph capture (the edges become graph dependencies)."""

    _side = []          # idle side streams
    _events = []        # idle events

    def __init__(self):
        self.main = current_stream()
        self.side = Branch._side.pop() if Branch._side else Stream()
        self.ev_fork, self.ev_join = self._event(), self._event()
        self.held = []
        self._joined = False

    @staticmethod
    def _event():
        if Branch._events:
            return Branch._events.pop()
        e = ctypes.c_void_p()
        hip.tf_event_create(ctypes.byref(e))
        return e

    def __enter__(self):
        hip.tf_event_record(self.ev_fork, self.main.handle)
        hip.tf_stream_wait_event(self.side.handle, self.ev_fork)
        _stream_stack.append(self.side)
        assert _pool.held is None, "nested Branch recording is not supported"
        _pool.held = self.held
        return self

    def __exit__(self, *a):
        _pool.held = None
        _stream_stack.pop()
        hip.tf_event_record(self.ev_join, self.side.handle)

    def join(self):
        if self._joined:
            return
        self._joined = True
        hip.tf_stream_wait_event(current_stream().handle, self.ev_join)
        for ptr, n in self.held:
            _pool.release(ptr, n)
        self.held = []
        Branch._side.append(self.side)
        Branch._events.extend((self.ev_fork, self.ev_join))


def pool():
    return _pool


def _pool_free(ptr, n):
    _pool.release(ptr, n)


_NP = {"f16": np.float16, "f32": np.float32}


# bfloat16: numpy has no such type, so a DeviceArray of bfloat16 elements carries this tagged uint16 dtype (2-byte items; the tag lives
# in the dtype's metadata, which numpy ignores in comparisons -- use is_bfloat16).  The reference's op tests run bfloat16 next to
# float16 (tests/group_norm.py:12-19, tests/layer_norm.py:13-27, tests/linear.py:13); here it selects the tf_*_bf16 entries.
bfloat16 = np.dtype(np.uint16, metadata={"bfloat16": True})


def is_bfloat16(dtype):
    md = getattr(dtype, "metadata", None)
    return bool(md and md.get("bfloat16"))


def dtag(dtype):
    """TF_DTYPE_* tag of the C-ABI's dtype-tagged entries (tf_*_16): 0 = float16, 1 = bfloat16."""
    return 1 if is_bfloat16(dtype) else 0


def f32_to_bf16_bits(x):
    """float array -> uint16 bfloat16 bit patterns, round-to-nearest-even (what v_cvt_pk_bf16_f32 does); NaN stays NaN."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + (((u >> 16) & 1) + np.uint32(0x7FFF))) >> 16).astype(np.uint16)
    nan = (u & np.uint32(0x7FFFFFFF)) > np.uint32(0x7F800000)
    return np.where(nan, ((u >> 16) | np.uint32(0x40)).astype(np.uint16), r)


def bf16_bits_to_f32(b):
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


class DeviceArray:
    """(ptr, logical shape, dtype, layout) handle.  layout: 'nhwc' (4-D, logical NCHW) or 'row'."""
    __slots__ = ("ptr", "shape", "dtype", "layout", "_base", "_fin", "gn", "normed", "_uid", "_version", "__weakref__")
    _next_uid = [0]

    def __init__(self, ptr, shape, dtype=np.float16, layout=None, base=None):
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.layout = layout or ("nhwc" if len(self.shape) == 4 else "row")
        self._base = base
        self._fin = None
        self.gn = None       # (partials, chunks, groups): GroupNorm statistics emitted by the conv that produced this array
        self.normed = None   # (GroupNorm module, silu, z): z = that module applied to this array, written by the producing conv's split-K reduce
        DeviceArray._next_uid[0] += 1
        self._uid, self._version = DeviceArray._next_uid[0], 0

    @property
    def wkey(self):
        """Identity of this array's CONTENTS for derived-weight caches (folded / packed / concatenated weights): unique per handle
        (a recycled device pointer under a new handle is a different key) and bumped by every in-place upload."""
        return (self._uid, self._version)

    # -- construction
    @staticmethod
    def empty(shape, dtype=np.float16, layout=None):
        shape = tuple(int(s) for s in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        ptr, n = _pool.alloc(nbytes)
        a = DeviceArray(ptr, shape, dtype, layout)
        a._fin = weakref.finalize(a, _pool_free, ptr, n)
        return a

    @staticmethod
    def from_numpy(x, dtype=np.float16, layout=None):
        x = np.asarray(x)
        return DeviceArray.empty(x.shape, dtype, layout).copy_from_numpy(x)

    def copy_from_numpy(self, x):
        """Synchronous host -> device copy into this array (same logical shape; cast / laid out on the host first).
        Use this instead of ``tf_memcpy(ptr, some_expression.ctypes.data, ...)``: an address taken from a temporary
        array is dangling by the time the call runs, the staging array here stays referenced until the copy returned."""
        host = f32_to_bf16_bits(np.asarray(x)) if is_bfloat16(self.dtype) else np.asarray(x).astype(self.dtype, copy=False)
        if host.shape != self.shape:
            raise ValueError(f"copy_from_numpy: host shape {host.shape} != device shape {self.shape}")
        if self.layout == "nhwc":
            host = host.transpose(0, 2, 3, 1)
        host = np.ascontiguousarray(host)
        if _sh():
            # the copy runs on the NULL stream and the library's streams are non-blocking: kernels already queued on the current
            # stream may still be reading the block this array was just handed by the pool -- drain it first
            hip.tf_stream_sync(_sh())
        hip.tf_memcpy(self.ptr, host.ctypes.data, host.nbytes, H2D)
        self._version += 1
        return self

    @staticmethod
    def zeros(shape, dtype=np.float16, layout=None):
        a = DeviceArray.empty(shape, dtype, layout)
        hip.tf_memset_async(a.ptr, 0, a.nbytes, _sh())
        return a

    # -- properties
    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    def numpy(self):
        """Blocking download in the *logical* layout (NCHW for images), as float32."""
        hip.tf_stream_sync(_sh())
        if self.layout == "nhwc":
            n, c, h, w = self.shape
            host = np.empty((n, h, w, c), dtype=self.dtype)
        else:
            host = np.empty(self.shape, dtype=self.dtype)
        if host.nbytes:
            hip.tf_memcpy(host.ctypes.data, self.ptr, host.nbytes, D2H)
        if self.layout == "nhwc":
            host = host.transpose(0, 3, 1, 2)
        if is_bfloat16(self.dtype):
            return bf16_bits_to_f32(np.ascontiguousarray(host))
        return np.ascontiguousarray(host).astype(np.float32)

    def view(self, shape, layout="row", offset_elems=0):
        """Re-view of the same memory (no copy); keeps the parent alive."""
        return DeviceArray(self.ptr + offset_elems * self.dtype.itemsize, shape, self.dtype, layout, base=self)

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = list(shape)
        if -1 in shape:
            i = shape.index(-1)
            shape[i] = self.size // max(1, -int(np.prod(shape)))
        assert int(np.prod(shape)) == self.size, (self.shape, shape)
        assert self.layout == "row", "reshape of an NHWC image: use .tokens() / .image()"
        return self.view(tuple(shape), "row")

    def tokens(self):
        """(b,c,h,w) image -> (b, h*w, c) tokens: attention/attention.py:71 as a free re-view."""
        assert self.layout == "nhwc"
        n, c, h, w = self.shape
        return self.view((n, h * w, c), "row")

    def image(self, b, c, h, w):
        """(b, h*w, c) tokens -> (b,c,h,w) image: attention/attention.py:74 as a free re-view."""
        assert self.layout == "row" and self.size == b * c * h * w
        return self.view((b, c, h, w), "nhwc")

    def astype(self, dtype):
        dtype = np.dtype(dtype)
        if dtype == self.dtype:
            return self
        out = DeviceArray.empty(self.shape, dtype, self.layout)
        if dtype == np.float16 and self.dtype == np.float32:
            hip.tf_cast_f32_to_f16(out.ptr, self.ptr, self.size, _sh())
        elif dtype == np.float32 and self.dtype == np.float16:
            hip.tf_cast_f16_to_f32(out.ptr, self.ptr, self.size, _sh())
        else:
            raise TypeError(f"astype {self.dtype} -> {dtype}")
        return out

    def __add__(self, other):
        assert isinstance(other, DeviceArray) and other.size == self.size and self.dtype.itemsize == 2 and dtag(other.dtype) == dtag(self.dtype)
        out = DeviceArray.empty(self.shape, self.dtype, self.layout)
        hip.tf_add_16(dtag(self.dtype), out.ptr, self.ptr, other.ptr, self.size, _sh())
        return out

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, layout={self.layout}, ptr=0x{self.ptr:x})"


def asarray(x, dtype=np.float16, layout=None):
    """numpy / torch-CPU / DeviceArray -> DeviceArray (cp.asarray in the reference, storage/state.py:20)."""
    if isinstance(x, DeviceArray):
        return x
    if hasattr(x, "detach"):
        x = x.detach().cpu()
        if str(x.dtype) == "torch.bfloat16":           # (numpy cannot hold it: go through float32, exact)
            x, dtype = x.float(), bfloat16
        x = x.numpy()
    return DeviceArray.from_numpy(np.asarray(x), dtype, layout)


def _unary(fn_name, x):
    out = DeviceArray.empty(x.shape, x.dtype, x.layout)
    getattr(hip, fn_name)(out.ptr, x.ptr, x.size, _sh())
    return out


class Tensor:
    """Activation statics of storage/tensor.py:64-86 (used as free functions throughout the model),
    plus the from_np / zeros / eval / to handle-holder API of :10-62 mapped onto DeviceArray."""

    def __init__(self, shape, dtype=np.float32, device=None, data=None):
        self.shape = shape if data is None else data.shape
        self.dtype = np.dtype(dtype if data is None else data.dtype)
        self.data = data
        self.num_elem = math.prod(self.shape)
        self.nbytes = self.num_elem * self.dtype.itemsize
        self.dt_ptr = ctypes.c_void_p()
        self.device = device if device is not None else Tensor.default_device()     # a Device, as in storage/tensor.py:11
        self.strides = None

    def eval(self):
        """cudaMalloc + H2D copy (storage/tensor.py:20-34)."""
        ensure_init()
        hip.tf_malloc(ctypes.byref(self.dt_ptr), self.nbytes)
        if self.data is not None:
            host = np.ascontiguousarray(self.data)
            hip.tf_memcpy(self.dt_ptr, host.ctypes.data, host.nbytes, H2D)
        else:
            hip.tf_memset_async(self.dt_ptr, 0, self.nbytes, None)
        return self

    def to(self, device):
        """D2H copy + free (storage/tensor.py:36-50)."""
        if str(device) == "cpu" and self.dt_ptr:
            if self.data is None or not self.data.flags.writeable:
                self.data = np.empty(self.shape, dtype=self.dtype)
            self.data = np.ascontiguousarray(self.data)
            hip.tf_device_sync()
            hip.tf_memcpy(self.data.ctypes.data, self.dt_ptr, self.data.nbytes, D2H)
            hip.tf_free(self.dt_ptr)
            self.dt_ptr = ctypes.c_void_p()
            from .device import Device
            self.device = Device("cpu")
        return self

    @staticmethod
    def default_device():
        """storage/tensor.py:59-61 (Device("cuda") there)."""
        from .device import Device
        return Device("hip")

    def T(self, out_tensor, axes=(1, 0)):
        """storage/tensor.py:90-92: permute this tensor's axes into ``out_tensor`` through the owning Device (the reference's
        Device.transpose launch); ``out_tensor`` takes the permuted shape, and so does its host array for the ``to('cpu')`` that follows."""
        self.device.transpose(out_tensor, self, tuple(axes))
        if out_tensor.data is not None:
            out_tensor.data = np.ascontiguousarray(out_tensor.data).reshape(out_tensor.shape)
        return out_tensor

    @staticmethod
    def from_np(data):
        return Tensor(data.shape, data.dtype, data=data)

    @staticmethod
    def zeros(shape, dtype):
        return Tensor.from_np(np.zeros(shape, dtype=dtype))

    # -- activations
    @staticmethod
    def sigmoid(x): return _unary("tf_sigmoid_f16", x)
    @staticmethod
    def silu(x): return _unary("tf_silu_f16", x)
    @staticmethod
    def swish(x): return _unary("tf_silu_f16", x)
    @staticmethod
    def quick_gelu(x): return _unary("tf_quick_gelu_f16", x)
    @staticmethod
    def gelu(x): return _unary("tf_gelu_f16", x)

    @staticmethod
    def sequential(iterable, init):
        return functools.reduce(lambda x, f: f(x), iterable, init)
