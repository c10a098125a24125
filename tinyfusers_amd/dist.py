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
_NONCE_BYTES = 16


def job_nonce():
    """16 bytes every rank of ONE job derives identically and another job does not: a digest of $TF_COMM_NONCE (``bench.py --gpus N`` draws one
    per launch and hands it to its ranks through the environment), else of torchrun's run id + the launcher's pid (every local rank is a child
    of the same agent process; the default run id of a stand-alone torchrun is the constant "none", so the id alone would not do).  Ranks
    started by hand with neither get the all-zero nonce, i.e. the round-4 behaviour: give them a $TF_COMM_NONCE or a fresh $TF_COMM_ID_FILE."""
    import hashlib
    tok = os.environ.get("TF_COMM_NONCE")
    if not tok and os.environ.get("TORCHELASTIC_RUN_ID") is not None:
        tok = "%s/%d" % (os.environ["TORCHELASTIC_RUN_ID"], os.getppid())
    return hashlib.sha256(tok.encode()).digest()[:_NONCE_BYTES] if tok else bytes(_NONCE_BYTES)


def comm_id_path():
    """Where the ranks of one job meet: $TF_COMM_ID_FILE, else a per-user, per-job name derived from the rendezvous the launcher already gave every
    rank and the job's nonce (a file an earlier job left behind then has another name, and another user's cannot be mistaken for ours)."""
    p = os.environ.get("TF_COMM_ID_FILE")
    if p:
        return p
    import tempfile
    return os.path.join(tempfile.gettempdir(), "tf_comm_id_%d_%s_%s_%s" % (os.getuid(), os.environ.get("MASTER_ADDR", "local"),
                                                                          os.environ.get("MASTER_PORT", str(os.getppid())), job_nonce().hex()[:16]))


def exchange_unique_id(rank, make_id, path=None, timeout=120.0, nonce=None):
    """Rank 0 calls make_id() -> 128 bytes and publishes [job nonce | id] atomically: any file already at the name is unlinked first, the bytes go
    to a temporary name created with O_EXCL and mode 0600, then a rename.  Every other rank waits for a file that carries THIS job's nonce -- a
    file left behind by a job that died between publishing and release_unique_id (or planted by someone else) is ignored, so nobody joins
    ncclCommInitRank with a dead id.  Returns the id.  The file stays until rank 0 calls release_unique_id (after every rank has joined)."""
    import time
    path = path or comm_id_path()
    nonce = job_nonce() if nonce is None else bytes(nonce)
    assert len(nonce) == _NONCE_BYTES
    if rank == 0:
        uid = bytes(make_id())
        assert len(uid) == UNIQUE_ID_BYTES, len(uid)
        try:
            os.unlink(path)                                   # a stale id of an earlier job must not be readable while we write ours
        except OSError:
            pass
        tmp = "%s.%d.tmp" % (path, os.getpid())
        try:
            os.unlink(tmp)
        except OSError:
            pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(nonce + uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                blob = f.read()
            if len(blob) == _NONCE_BYTES + UNIQUE_ID_BYTES and blob[:_NONCE_BYTES] == nonce:
                return blob[_NONCE_BYTES:]
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError("rank %d: no RCCL unique id of this job at %s after %.0f s (did rank 0 start? do all ranks share TF_COMM_NONCE / the launcher?)" % (rank, path, timeout))
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
