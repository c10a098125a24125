"""world_size-2 gloo rehearsal of the N>1 path on CPU: rank 0 fills the packed weight arena, ONE broadcast, every
rank decodes identical tensors; images are sharded with distinct seeds; timing is the max over ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from tinyfusers_amd.dist import broadcast_arena, max_over_ranks, pack_tensor, plan_arena, shard_range
    from tinyfusers_amd.storage.synth import synth_normal, synth_tensor
    shapes = oracle.unet_param_shapes(oracle.TINY)
    offs, total = plan_arena(shapes)
    arena = torch.zeros(total, dtype=torch.uint8)
    if rank == 0:                       # only rank 0 ever generates / reads weights
        for k, s in shapes.items():
            w = pack_tensor(synth_tensor(5, k, s))
            arena[offs[k]:offs[k] + w.nbytes] = torch.from_numpy(w.view(np.uint8).reshape(-1))
    broadcast_arena(arena, src=0)
    # every rank decodes the same bytes
    k = "input_blocks.1.0.in_layers.2.weight"
    w = arena[offs[k]:offs[k] + int(np.prod(shapes[k])) * 2].numpy().view(np.float16).reshape(shapes[k][0], 3, 3, shapes[k][1])
    ok = np.array_equal(w, pack_tensor(synth_tensor(5, k, shapes[k])))
    lo, hi = shard_range(8, rank, world)
    lat = synth_normal(1234 + rank, "sd.latent", (1, 4, 8, 8))
    tmax = max_over_ranks(1.0 + rank)
    q.put((rank, ok, int(arena.to(torch.int64).sum()), (lo, hi), float(lat[0, 0, 0, 0]), tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_and_sharding_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ok0, sum0, sh0, l0, t0), (r1, ok1, sum1, sh1, l1, t1) = res
    assert ok0 and ok1 and sum0 == sum1 and sum0 > 0          # identical arenas after the one broadcast
    assert sh0 == (0, 4) and sh1 == (4, 8)                    # images sharded, no overlap
    assert l0 != l1                                           # distinct per-rank latent seeds
    assert t0 == t1 == 2.0                                    # timing = max over ranks
