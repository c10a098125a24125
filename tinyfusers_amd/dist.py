"""Batch-parallel plumbing for the sampler: one process per GPU, images sharded across ranks, ONE collective
(the weight-arena broadcast from rank 0) at start-up and none per step.

torch.distributed is used as plumbing only (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests); nothing here touches the reference, which has no distributed code at all (device_id = 0 hard-coded,
storage/device.py:23)."""
import os

import numpy as np

ALIGN = 256


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(global_batch, rank, world):
    """Images [lo, hi) of a global batch owned by `rank` (contiguous, balanced, CFG pairs never split)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def plan_arena(shapes, itemsize=2):
    """name -> byte offset in one packed arena (256-B aligned slots), and the arena size."""
    offs, off = {}, 0
    for k, s in shapes.items():
        offs[k] = off
        off += (int(np.prod(s)) * itemsize + ALIGN - 1) // ALIGN * ALIGN
    return offs, off


def pack_tensor(w):
    """Device layout of one tensor: 4-D conv weights go (K,C,R,S) -> KRSC, everything else row-major fp16."""
    w = np.asarray(w, dtype=np.float16)
    if w.ndim == 4:
        w = w.transpose(0, 2, 3, 1)
    return np.ascontiguousarray(w)


def broadcast_arena(arena, src=0):
    """The one collective of the path: broadcast the packed weight arena (a 1-D uint8 torch tensor)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("TF_BENCH_FORCE_DIST")):
        dist.broadcast(arena, src=src)
    return arena


def max_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- the C-ABI's own collective (csrc/comm.hip: tf_comm_unique_id / tf_comm_init_rank / tf_bcast / tf_comm_destroy over librccl) -----------------
# For hosts without torch.distributed -- and what `bench.py --comm tf` exercises: rank 0 draws the 128-byte RCCL unique id and hands it to
# the other ranks through a FILE (any host channel would do; a file needs nothing but a shared /tmp on the one node the path runs on).
UNIQUE_ID_BYTES = 128


def comm_id_path():
    """Where the ranks of one job meet: $TF_COMM_ID_FILE, else a name derived from the rendezvous the launcher already gave every rank."""
    p = os.environ.get("TF_COMM_ID_FILE")
    if p:
        return p
    import tempfile
    return os.path.join(tempfile.gettempdir(), "tf_comm_id_%s_%s" % (os.environ.get("MASTER_ADDR", "local"), os.environ.get("MASTER_PORT", str(os.getppid()))))


def exchange_unique_id(rank, make_id, path=None, timeout=120.0):
    """Rank 0 calls make_id() -> 128 bytes and publishes them atomically (write to a temporary name, then rename); every other rank waits
    for the file.  Returns the id.  The file is left in place until rank 0 calls release_unique_id (after every rank has joined)."""
    import time
    path = path or comm_id_path()
    if rank == 0:
        uid = bytes(make_id())
        assert len(uid) == UNIQUE_ID_BYTES, len(uid)
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == UNIQUE_ID_BYTES:
                return uid
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError("rank %d: no RCCL unique id at %s after %.0f s (did rank 0 start?)" % (rank, path, timeout))
        time.sleep(0.01)


def release_unique_id(path=None):
    try:
        os.remove(path or comm_id_path())
    except OSError:
        pass


class TfComm:
    """One RCCL communicator through the C-ABI (tf_init(device) must have run on this rank)."""

    def __init__(self, rank, world, path=None):
        import ctypes
        from .native import hip

        def make_id():
            buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
            hip.tf_comm_unique_id(buf)
            return buf.raw
        self.rank, self.world, self._path = rank, world, path or comm_id_path()
        uid = exchange_unique_id(rank, make_id, self._path)
        self._h = ctypes.c_void_p()
        hip.tf_comm_init_rank(ctypes.byref(self._h), ctypes.create_string_buffer(uid, UNIQUE_ID_BYTES), world, rank)   # (collective: returns once every rank has joined)
        if rank == 0:
            release_unique_id(self._path)

    def bcast(self, ptr, nbytes, root=0, stream=None):
        from .native import hip
        hip.tf_bcast(self._h, ptr, nbytes, root, stream)

    def destroy(self):
        from .native import hip
        if self._h:
            hip.tf_comm_destroy(self._h)
            self._h = None
