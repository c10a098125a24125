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
