#!/usr/bin/env python3
"""UNet denoising steps/sec, SD1.5 512x512 (64x64x4 latent), one image per GPU, CFG batch 2.

    python bench.py --gpus N --steps K --warmup W           # starts its N ranks itself (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

``python bench.py --gpus N`` with N > 1 and no RANK in the environment is a LAUNCHER: it makes no GPU call, starts
``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a child, relays rank 0's single JSON line and exits
with the child's status.  Under torch.distributed.run, WORLD_SIZE must equal --gpus (anything else is an error, not a
silent 1-GPU run).

One process per GPU.  Rank 0 generates the synthetic weights and broadcasts the packed fp16 weight arena once
over RCCL (torch.distributed "nccl"); there is NO per-step collective: every rank runs its own image's DDIM
trajectory (weak scaling, value = all images' steps / max-over-ranks time).

A "step" = one StableDiffusion.__call__ of the reference (variants/sd.py:56-59): CFG duplicate, one UNet forward
at batch 2, CFG combine + DDIM update, replayed as one HIP graph.  Inputs are resident in HBM before timing.
Extra objects on the JSON line: "roofline" (the implicit-GEMM conv/linear kernel family, timed per launch with
HIP events on its own stream in an instrumented eager pass), "cpu_baseline" (the CPU oracle = the torch-CPU op
path the reference's tests compare against, timed on this box's host cores; rank 0, N=1 only) and "e2e" (BASELINE config 3:
CLIP text encoder x2 -> 50 graph-replayed steps -> VAE decode, img/s with the three segment times; N=1 only).

``--dry-run`` rehearses the N>1 plumbing on CPU (gloo): arena broadcast + image shard + max-over-ranks, no kernels.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = 1.6088e12          # algorithmic FLOPs of one step (conv 887.89 G + linear 466.49 G + SDPA 252.10 G + 2.28 G)
PEAK_MFMA_TFLOPS = {"fp16": 2500.0, "bf16": 2500.0, "fp8": 5000.0}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
TRAFFIC_PROFILE = os.path.join("profiles", "r05_pmc_traffic.json")
FAMILY_PROFILE = os.path.join("profiles", "r05_kernel_family.json")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the config-3 end-to-end leg (CLIP x2 -> 50 steps -> VAE decode)")
    ap.add_argument("--e2e-images", type=int, default=3)
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 leg (4 images per GPU, 96x96 latents, fp16 and fp8 conv/linear, --config5-steps timed graph replays each)")
    ap.add_argument("--config5-steps", type=int, default=12)
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of the HIP-graph replay")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--tune-cache", default="", help="file to load/save the GEMM autotuner's per-shape choices (optional)")
    ap.add_argument("--images", type=int, default=1, help="images per GPU (UNet batch = 2x); default 1 = BASELINE config 2")
    ap.add_argument("--latent", type=int, default=64, help="latent height = width; default 64 (512x512 images); 96 = 768x768 (config 5)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16", "fp8"], help="fp16 (the metric's configuration); bf16 = every 16-bit tensor bfloat16 (a correct step on the plain per-op structure, not a tuned one); "
                    "fp8 = config 5: block-scaled e4m3 conv / linear operands, fp32 accumulate, fp16 residual stream")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the multi-rank plumbing (gloo): no GPU, no kernels")
    ap.add_argument("--comm", default="torch", choices=["torch", "tf"], help="who carries the one weight-arena broadcast: torch.distributed (nccl = RCCL), or the C-ABI's own "
                    "tf_comm_unique_id / tf_comm_init_rank / tf_bcast (csrc/comm.hip; the id travels through a file, tinyfusers_amd.dist.TfComm)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher: the parent of an N-rank run.  Never touches the GPU (no torch import, no HIP call).
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """How many GPUs the ranks will see, WITHOUT a HIP call (the launcher never touches the GPU): KFD topology nodes that have compute units,
    cut down by ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES.  None = unknown (no KFD sysfs, or its properties are unreadable): no check then."""
    import re
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(base)
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            m = re.search(r"^simd_count\s+(\d+)", open(os.path.join(base, d, "properties")).read(), re.M)
        except OSError:
            continue
        if m and int(m.group(1)) > 0:
            n += 1
    if n == 0:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def launch_ranks(args, argv):
    if not args.dry_run:
        have = visible_gpus()
        if have is not None and have < args.gpus:
            # one named error from the parent, before any rank starts (eight ranks failing one by one inside torch.distributed.run say much less)
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU{'s are' if have != 1 else ' is'} visible on this host "
                             "(KFD topology, *_VISIBLE_DEVICES): not starting any rank\n")
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one token per launch: the ranks' RCCL unique-id hand-over (--comm tf, tinyfusers_amd.dist.exchange_unique_id) accepts only a file that
    # carries it, so an id file an earlier job left behind on the same rendezvous port is never mistaken for this job's
    env.setdefault("TF_COMM_NONCE", "%d-%s" % (os.getpid(), os.urandom(8).hex()))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, text=True)
    line = None
    for ln in p.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()                       # rank 0's single result line
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if rc != 0:
        sys.stderr.write(f"bench.py: a rank failed (torch.distributed.run exit status {rc})\n")
        return rc
    if line is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n")
        return 1
    print(line, flush=True)
    return 0


def csrc_hash():
    """Hash of the kernel sources the loaded library was built from (profiles are stamped with it)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "tinyfusers_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------------------
def dry_run(args):
    """World-size-N rehearsal on CPU (gloo): rank 0 fills the packed weight arena of the TINY UNet, ONE broadcast, every rank
    checks the bytes, takes its image shard and seeds, and the timing is the max over ranks.  No kernels, no GPU."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist
    import oracle
    from tinyfusers_amd.dist import broadcast_arena, max_over_ranks, pack_tensor, plan_arena, shard_range
    from tinyfusers_amd.storage.synth import synth_normal, synth_tensor
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = oracle.unet_param_shapes(oracle.TINY)
    offs, total = plan_arena(shapes)
    arena = torch.zeros(total, dtype=torch.uint8)
    t0 = time.time()
    if rank == 0:
        for k, s in shapes.items():
            w = pack_tensor(synth_tensor(5, k, s))
            arena[offs[k]:offs[k] + w.nbytes] = torch.from_numpy(w.view(np.uint8).reshape(-1))
    t_gen = time.time() - t0
    t1 = time.time()
    broadcast_arena(arena, src=0)
    t_bcast = time.time() - t1
    k = next(iter(shapes))
    got = arena[offs[k]:offs[k] + int(np.prod(shapes[k])) * 2].numpy().view(np.float16)
    assert np.array_equal(got, pack_tensor(synth_tensor(5, k, shapes[k])).reshape(-1)), "arena differs after the broadcast"
    lo, hi = shard_range(world * args.images, rank, world)
    assert hi - lo == args.images
    lat = synth_normal(1234 + rank, "sd.latent", (args.images, 4, args.latent, args.latent))     # this rank's images: its own seed
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))                      # stands in for the K replayed steps; the slowest rank sets the time
    if world > 1:
        dist.barrier()
    wall = max_over_ranks(time.perf_counter() - t0)
    checks = torch.tensor([float(arena.to(torch.int64).sum()), float(lat.reshape(-1)[0])], dtype=torch.float64)
    allc = [torch.zeros_like(checks) for _ in range(world)]
    if world > 1:
        dist.all_gather(allc, checks)
    else:
        allc = [checks]
    if rank == 0:
        assert all(float(c[0]) == float(allc[0][0]) for c in allc), "ranks hold different arenas"
        assert len({float(c[1]) for c in allc}) == world, "ranks must draw distinct latents"
        print(json.dumps({"metric": "unet_denoise_steps_per_sec", "value": round(world * args.images * args.steps / wall, 2), "unit": "steps/s",
                          "n_gpus": world, "rccl_ranks": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "none", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "dry run: arena broadcast + image shard + max-over-ranks on CPU (gloo), no kernels",
                                     "global_batch": world * args.images, "images_per_rank": args.images, "latent": [4, args.latent, args.latent],
                                     "requested_dtype": args.dtype, "parallelism": f"dp{world}"},
                          "weights": {"bytes": total, "synth_s": round(t_gen, 3), "bcast_s": round(t_bcast, 4)}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------------------------------
def build_weight_arena(unet, rank, world, device_index, comm="torch"):
    """Pack every UNet tensor into ONE fp16 device arena (torch owns the memory: plumbing), filled on rank 0 and
    broadcast once over RCCL; every module leaf becomes a view into it."""
    import torch
    from tinyfusers_amd.storage.state import unet_param_shapes, update_state
    from tinyfusers_amd.storage.synth import synth_tensor
    from tinyfusers_amd.storage.tensor import DeviceArray
    from tinyfusers_amd.native import hip
    from tinyfusers_amd.dist import broadcast_arena, pack_tensor, plan_arena
    shapes = unet_param_shapes(unet)
    offs, off = plan_arena(shapes)
    from tinyfusers_amd import config as _cfg
    from tinyfusers_amd.storage.tensor import bfloat16 as _bf16, f32_to_bf16_bits
    wdtype = _bf16 if _cfg.is_bf16() else np.float16
    arena = torch.empty(off, dtype=torch.uint8, device=f"cuda:{device_index}")
    base = arena.data_ptr()
    t0 = time.time()
    if rank == 0:
        from concurrent.futures import ThreadPoolExecutor

        def gen(k):
            if _cfg.is_bf16():                      # bfloat16 bit patterns in the same packed layout (conv weights KRSC)
                w = np.asarray(synth_tensor(0, k, shapes[k]), dtype=np.float32)
                return k, np.ascontiguousarray(f32_to_bf16_bits(w.transpose(0, 2, 3, 1) if w.ndim == 4 else w))
            return k, pack_tensor(synth_tensor(0, k, shapes[k]))
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            for k, w in ex.map(gen, list(shapes)):
                hip.tf_memcpy(base + offs[k], w.ctypes.data, w.nbytes, 1)
    t_gen = time.time() - t0
    t_bcast = 0.0
    import torch.distributed as dist
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.time()
        if comm == "tf":                       # the C-ABI's own RCCL wrappers (what a host without torch.distributed would call)
            from tinyfusers_amd.dist import TfComm
            c = TfComm(rank, world)
            c.bcast(base, off, 0, None)
            hip.tf_stream_sync(None)
            c.destroy()
        else:
            broadcast_arena(arena, src=0)      # the ONLY collective of the path (RCCL over xGMI)
        torch.cuda.synchronize()
        t_bcast = time.time() - t1
    state = {k: DeviceArray(base + offs[k], shapes[k], wdtype, None, base=arena) for k in shapes}
    update_state(unet, state, "")
    return arena, off, t_gen, t_bcast, state


def host_cores():
    """Host cores this process may actually use: the affinity mask, cut down to the cgroup CPU quota where one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(steps):
    """The reference's CPU op path (F.conv2d / GroupNorm / F.layer_norm / F.linear / SDPA, fp32) = oracle.sd_step,
    on a bounded sample: `steps` full denoising steps of the same workload after one warm-up; plus the two single ops of
    BASELINE config 1 (tests/conv2d.py, tests/group_norm.py of the reference) at a bounded size."""
    import torch
    import torch.nn.functional as F
    import oracle
    from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict
    cores = host_cores()
    torch.set_num_threads(cores)
    W = {k: torch.from_numpy(v.astype(np.float32)) for k, v in synth_state_dict(oracle.unet_param_shapes(oracle.SD15), 0).items()}
    lat = synth_normal(1234, "sd.latent", (1, 4, 64, 64))
    ctx = synth_normal(1234, "sd.context", (1, 77, 768)); unc = synth_normal(1234, "sd.uncond", (1, 77, 768))
    ts, al, ap = oracle.sampler_schedule(50)
    x = torch.from_numpy(lat)
    times = []
    for n in range(steps + 1):
        i = 49 - n
        t0 = time.time()
        x = oracle.sd_step(unc, ctx, x, np.array([ts[i]], np.float32), al[i:i + 1], ap[i:i + 1], np.array([7.5]), W)
        times.append(time.time() - t0)
    sec = float(np.median(times[1:]))
    # config 1: the reference's own single-op tests on CPU.  tests/conv2d.py:13-33: X (1,2,10000,10000) fp32 (800 MB), W (1,2,2,2), pad 0,
    # stride 1 -- at the test's own size when the host has the memory for it (input + output + torch's scratch: < 4 GB), else bounded to
    # 4000 x 4000 (same arithmetic per output); tests/group_norm.py:22-41: (2048, C, 2, 2) with 2 groups, C = 1600 (the largest of its
    # parametrisation).
    side = 4000
    try:
        avail_kb = int(next(ln for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")).split()[1])
        if avail_kb > 16 * 1024 * 1024:
            side = 10000
    except Exception:
        pass
    xc = torch.from_numpy(np.random.default_rng(7).standard_normal((1, 2, side, side), dtype=np.float32)); wc = torch.from_numpy(synth_normal(7, "cfg1.conv.w", (1, 2, 2, 2)))
    F.conv2d(xc[:, :, :1000, :1000], wc)
    t0 = time.time(); yc = F.conv2d(xc, wc); t_conv = time.time() - t0
    conv_shape = f"x(1,2,{side},{side}) w(1,2,2,2) pad 0 stride 1 fp32 (" + ("the reference test's own size" if side == 10000 else "16 % of the reference test's 10000^2 pixels: host memory") + ")"
    n_out = yc.numel()
    del xc, yc
    xg = torch.from_numpy(synth_normal(7, "cfg1.gn.x", (2048, 1600, 2, 2)))
    oracle.group_norm(xg, 2, 1e-5)
    t0 = time.time(); oracle.group_norm(xg, 2, 1e-5); t_gn = time.time() - t0
    return {"value": 1.0 / sec, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} full SD1.5 denoising steps (CFG batch 2, 64x64 latent) after 1 warm-up, torch-CPU fp32 oracle, median {sec:.2f} s/step",
            "single_ops": {"conv2d_tests_conv2d_py": {"shape": conv_shape, "ms": round(t_conv * 1e3, 2), "gpixel_per_s": round(n_out / t_conv / 1e9, 3)},
                           "group_norm_tests_group_norm_py": {"shape": "x(2048,1600,2,2) 2 groups fp32", "ms": round(t_gn * 1e3, 2),
                                                              "gb_per_s": round(2 * xg.numel() * 4 / t_gn / 1e9, 2)}}}


def e2e_leg(sd, n_images, seed):
    """BASELINE config 3 (example/sd1.py:44-79 of the reference): both prompts through the CLIP text encoder, the full
    50-step schedule as graph replays, VAE decode to a uint8 image.  Returns img/s and the three segment times (medians)."""
    import tinyfusers_amd.storage.tensor as T
    from tinyfusers_amd.storage.state import param_shapes, update_state
    from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        for name, sub in (("first_stage_model", sd.first_stage_model), ("cond_stage_model", sd.cond_stage_model)):
            update_state(sub, synth_state_dict(param_shapes(sub, name), 0), name)
    rng = np.random.default_rng(seed)
    prompt = np.full((1, 77), 49407, dtype=np.int64); prompt[0, 0] = 49406; prompt[0, 1:10] = rng.integers(0, 49406, 9)
    empty = np.full((1, 77), 49407, dtype=np.int64); empty[0, 0] = 49406
    text_model = sd.cond_stage_model.transformer.text_model
    text_model(prompt)                                 # first call folds the LayerNorms / fuses q|k|v once
    T.hip.tf_stream_sync(None)
    timesteps = list(range(1, 1000, 20))
    alphas = sd.alphas_cumprod[timesteps]
    alphas_prev = np.concatenate((np.array([1.0]), alphas[:-1])).astype(np.float32)
    latent = sd.latent_from_numpy(synth_normal(seed, "sd.latent", (1, 4, 64, 64)))
    recs = []
    for n in range(n_images + 1):                      # image 0 = compile + warm-up
        t0 = time.perf_counter()
        context = text_model(prompt)
        unconditional_context = text_model(empty)
        T.hip.tf_stream_sync(None)
        t1 = time.perf_counter()
        if n == 0:
            sd.compile(unconditional_context, context, latent, timesteps=timesteps)
        t2 = time.perf_counter()
        if n > 0:
            sd.set_context(unconditional_context, context)     # same buffers as the captured graph: the stacked context and its K|V projection, in place
        sd.set_latent(synth_normal(seed + n, "sd.latent", (1, 4, 64, 64)))
        for index, timestep in list(enumerate(timesteps))[::-1]:
            sd.step(timestep, alphas[index], alphas_prev[index], 7.5)
        sd.synchronize()
        t3 = time.perf_counter()
        assert np.isfinite(latent.numpy()).all(), f"e2e image {n}: non-finite latent"
        t4 = time.perf_counter()
        with T.use_stream(sd._stream):
            img = sd.decode(latent)
        t5 = time.perf_counter()
        assert img.shape == (512, 512, 3) and img.dtype == np.uint8
        recs.append((t1 - t0, t3 - t2, t5 - t4))
    c, s, d = (float(np.median([r[i] for r in recs[1:]])) for i in range(3))
    return {"metric": "sd15_end_to_end_images_per_sec", "value": round(1.0 / (c + s + d), 3), "unit": "img/s", "images_timed": n_images,
            "clip_ms": round(c * 1e3, 2), "sampler_ms": round(s * 1e3, 2), "decode_ms": round(d * 1e3, 2), "steps": 50,
            "workload": "CLIP text encoder x2 -> 50 DDIM steps (CFG, one HIP-graph replay each) -> VAE decode to 512x512x3 uint8, batch 1, fp16"}


def roofline_leg(sd, run, n_inst, peak, default_workload):
    """The dominant kernel family = the implicit-GEMM conv2d + linear launches (84 % of the step's FLOPs): `n_inst` instrumented eager
    steps with HIP events on the launch stream around every launch of the family.  `achieved` counts a GEMM together with the split-K
    reduce launch that finishes it (a split conv is not done before its reduce has run); the bracket that ends behind the k_igemm* kernel
    itself -- the figure rocprofv3 lists under that name -- is the sub-key `gemm_kernel_only`."""
    import ctypes
    from tinyfusers_amd.native import hip, lib
    lib.tf_prof_enable(1)
    run(n_inst, True)
    sd.synchronize()
    gfull, gms, gfl, gl = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
    hip.tf_prof_read_full(ctypes.byref(gfull), ctypes.byref(gms), ctypes.byref(gfl), ctypes.byref(gl))
    fams = {}
    for fam, name in ((1, "group_norm"), (2, "splitk_reduce"), (3, "layer_norm"), (4, "sdpa")):
        fms, fwork, fn = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
        hip.tf_prof_read_family(fam, ctypes.byref(fms), ctypes.byref(fwork), ctypes.byref(fn))
        fams[name] = (fms.value, fwork.value, fn.value)
    lib.tf_prof_enable(0)
    ach = gfl.value / (gfull.value * 1e-3) / 1e12 if gfull.value > 0 else 0.0
    ach_k = gfl.value / (gms.value * 1e-3) / 1e12 if gms.value > 0 else 0.0
    # HBM bytes per launch of this kernel family from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950 correction +
    # WRITE_SIZE, separate runs of this same command; tools/pmc_summary.py): they cannot be sampled live, so the profile is stamped with
    # the hash of the kernel sources it was taken on and is reported only while that still matches
    traffic, tnote, pj = None, f"no {TRAFFIC_PROFILE}", {}
    try:
        pj = json.load(open(os.path.join(ROOT, TRAFFIC_PROFILE)))
        if pj.get("csrc_sha16") == csrc_hash() and default_workload:
            traffic = round(pj["k_igemm"]["hbm_bytes_per_launch"])
            tnote = f"HBM bytes per launch ({TRAFFIC_PROFILE}, kernel sources {pj['csrc_sha16']})"
        else:
            tnote = f"{TRAFFIC_PROFILE} was taken on other kernel sources or another workload ({pj.get('csrc_sha16')} vs {csrc_hash()}): not reported"
    except Exception:
        pass
    # The HBM-bound families against the 8 TB/s roof (north_star: "rocprof HBM GB/s ... against the chip's peak" for the norm kernels): live
    # event time per launch (same brackets, same overhead correction as the GEMM family) over the launch's ALGORITHMIC bytes (every operand
    # read once, every result written once: what the launcher computes from its shapes); `traffic` = the fabric bytes per launch the
    # committed PMC passes counted for that family, quoted while the kernel sources match
    hbm = {}
    kernels = {"group_norm": ("k_gn_stats / k_gn_apply (GroupNorm [+ SiLU] apply; the statistics ride on the producing conv)", "k_gn"),
               "splitk_reduce": ("k_splitk_reduce / _gn / _gn_apply (partial slabs -> y [+ statistics, + the next GroupNorm applied])", "k_splitk_reduce"),
               "layer_norm": ("k_layer_norm* (stand-alone LayerNorm: inside the UNet it is folded into the consuming GEMM)", "k_layer_norm")}
    for name, (desc, pkey) in kernels.items():
        fms, fwork, fn = fams[name]
        if fn == 0 or fms <= 0:
            hbm[name] = {"kernel": desc, "launches_per_step": 0.0}
            continue
        gbs = fwork / (fms * 1e-3) / 1e9
        tr = None
        try:
            if pj.get("csrc_sha16") == csrc_hash() and default_workload and pkey in pj:
                tr = round(pj[pkey]["hbm_bytes_per_launch"])
        except Exception:
            pass
        hbm[name] = {"kernel": desc, "bound": "hbm", "launches_per_step": fn / n_inst, "avg_launch_us": round(fms * 1e3 / fn, 2), "ms_per_step": round(fms / n_inst, 4),
                     "algorithmic_bytes_per_launch": round(fwork / fn), "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                     "traffic": tr, "traffic_gbs": round(tr / (fms * 1e-3 / fn) / 1e9, 1) if tr else None}
    sms, swork, sn = fams["sdpa"]
    sdpa = None
    if sn and sms > 0:
        stf = swork / (sms * 1e-3) / 1e12
        sdpa = {"kernel": "k_sdpa_dma / k_sdpa (flash attention; FLOPs = 4 B NH Tq Tk d)", "bound": "mfma", "launches_per_step": sn / n_inst, "avg_launch_us": round(sms * 1e3 / sn, 2),
                "ms_per_step": round(sms / n_inst, 4), "achieved": round(stf, 1), "peak": PEAK_MFMA_TFLOPS["fp16"], "unit": "TFLOP/s", "frac": round(stf / PEAK_MFMA_TFLOPS["fp16"], 4)}
    # the same family in the committed rocprofv3 --kernel-trace --stats run (tools/prof_summary.py), quoted while the sources match
    rocprof = None
    try:
        fj = json.load(open(os.path.join(ROOT, FAMILY_PROFILE)))
        if fj.get("csrc_sha16") == csrc_hash() and default_workload:
            k, kr = fj["k_igemm"], fj.get("k_igemm_plus_reduce", fj["k_igemm"])
            per_step = gfl.value / n_inst
            rocprof = {"gemm_plus_reduce_ms_per_step": round(kr["ms_per_step"], 4),
                       "frac": round(per_step / (kr["ms_per_step"] * 1e-3) / 1e12 / peak, 4) if kr["ms_per_step"] > 0 else None,
                       "gemm_kernel_only_ms_per_step": round(k["ms_per_step"], 4), "gemm_kernel_only_avg_launch_us": round(k["avg_launch_us"], 2),
                       "gemm_kernel_only_frac": round(per_step / (k["ms_per_step"] * 1e-3) / 1e12 / peak, 4) if k["ms_per_step"] > 0 else None,
                       "file": FAMILY_PROFILE}
    except Exception:
        pass
    return {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": traffic, "traffic_unit": tnote,
            "kernel": "k_igemm* / k_gemm_c4 / k_gemm_ar (implicit-GEMM conv2d + linear) together with the split-K reduce launches that finish them",
            "launches_per_step": gl.value / n_inst, "avg_launch_us": round(gfull.value * 1e3 / max(1, gl.value), 2),
            "gemm_ms_per_step": round(gfull.value / n_inst, 4), "gemm_flop_per_step": gfl.value / n_inst,
            "gemm_kernel_only": {"achieved": round(ach_k, 1), "frac": round(ach_k / peak, 4), "ms_per_step": round(gms.value / n_inst, 4),
                                 "avg_launch_us": round(gms.value * 1e3 / max(1, gl.value), 2)},
            "event_bracket_overhead_us": round(float(lib.tf_prof_overhead_us()), 2),
            "rocprof": rocprof, "hbm": hbm, "sdpa": sdpa}


def config5_leg(wstate, steps, seed):
    """BASELINE config 5's per-GPU workload (4 images per GPU = UNet batch 8, 96 x 96 latents), fp16 and fp8 conv / linear, on the
    weights already resident: `steps` graph-replayed steps each after 2 warm-up steps, plus one instrumented eager step for the GEMM
    family.  Rank 0 at N = 1 only; the 8-GPU form of the config is the same per-GPU workload on every rank."""
    import contextlib
    import ctypes
    import io
    import tinyfusers_amd.storage.tensor as T
    from tinyfusers_amd import config
    from tinyfusers_amd.native import hip
    from tinyfusers_amd.storage.state import update_state
    from tinyfusers_amd.storage.synth import synth_normal
    from tinyfusers_amd.variants.sd import StableDiffusion
    B, S = 4, 96
    out = {"workload": f"{B} images per GPU, {S}x{S}x4 latents (768x768 images), UNet batch {2 * B}, one DDIM step; {steps} timed graph replays per dtype",
           "flop_per_image_step": 4.30e12}
    timesteps = list(range(1, 1000, 20))
    for dtype in ("fp16", "fp8"):
        config.set_dtype(dtype)
        try:
            sd = StableDiffusion(init=False)
            with contextlib.redirect_stdout(io.StringIO()):
                update_state(sd.model.diffusion_model, wstate, "")
            alphas = sd.alphas_cumprod[timesteps]
            alphas_prev = np.concatenate((np.array([1.0]), alphas[:-1])).astype(np.float32)
            lat = sd.latent_from_numpy(synth_normal(seed, "sd.latent", (B, 4, S, S)))
            ctx = T.DeviceArray.from_numpy(synth_normal(seed, "sd.context", (B, 77, 768)))
            unc = T.DeviceArray.from_numpy(synth_normal(seed, "sd.uncond", (B, 77, 768)))
            sd.compile(unc, ctx, lat, timesteps=timesteps)

            def run(n, eager):
                for s_ in range(n):
                    i = 49 - (s_ % 50)
                    sd.step(timesteps[i], alphas[i], alphas_prev[i], 7.5, eager=eager)
            run(2, False)
            sd.synchronize()
            ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
            hip.tf_event_create(ctypes.byref(ev0)); hip.tf_event_create(ctypes.byref(ev1))
            hip.tf_event_record(ev0, sd._stream.handle)
            run(steps, False)
            hip.tf_event_record(ev1, sd._stream.handle)
            sd.synchronize()
            ms = ctypes.c_float()
            hip.tf_event_elapsed_ms(ctypes.byref(ms), ev0, ev1)
            assert np.isfinite(lat.numpy()).all(), f"config5 {dtype}: non-finite latent"
            peak = PEAK_MFMA_TFLOPS[dtype]
            rl = roofline_leg(sd, run, 1, peak, False)
            per = ms.value / steps
            out[dtype] = {"image_steps_per_s": round(B * 1e3 / per, 2), "ms_per_step": round(per, 3),
                          "step_tflops": round(4.30e12 * B / (per * 1e-3) / 1e12, 1),
                          "gemm_family_tflops": rl["achieved"], "gemm_ms_per_step": rl["gemm_ms_per_step"], "gemm_launches_per_step": rl["launches_per_step"],
                          "frac_of_2500_tflops_f16_peak": round(rl["achieved"] / 2500.0, 4), "frac_of_5000_tflops_f8_peak": round(rl["achieved"] / 5000.0, 4),
                          "hbm": {k: {kk: v.get(kk) for kk in ("launches_per_step", "avg_launch_us", "ms_per_step", "algorithmic_bytes_per_launch", "achieved", "frac")} for k, v in rl["hbm"].items()},
                          "sdpa": {kk: rl["sdpa"].get(kk) for kk in ("launches_per_step", "avg_launch_us", "ms_per_step", "achieved", "frac")} if rl["sdpa"] else None}
            del sd
        finally:
            config.set_dtype("fp16")
    if "fp16" in out and "fp8" in out:
        out["fp8_over_fp16"] = round(out["fp16"]["ms_per_step"] / out["fp8"]["ms_per_step"], 3)
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    t_proc = time.perf_counter()                   # start-up clock of this rank (everything before the warm-up steps: imports, weights, compile + capture)
    in_group = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # started by torch.distributed.run (or by our launcher through it)
    if args.gpus > 1 and not in_group:
        return launch_ranks(args, argv)           # parent: no GPU call before or after
    if args.dry_run:
        return dry_run(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if in_group else 1
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if in_group else 0
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or give torch.distributed.run the same N")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    use_dist = in_group and (world > 1 or bool(os.environ.get("TF_BENCH_FORCE_DIST")))    # (FORCE: rehearse the RCCL path with one rank on a one-GPU box)
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)

    import tinyfusers_amd.storage.tensor as T
    from tinyfusers_amd import config
    from tinyfusers_amd.native import hip, lib
    from tinyfusers_amd.storage.synth import synth_normal
    from tinyfusers_amd.variants.sd import StableDiffusion
    import ctypes
    T.ensure_init(local_rank)
    config.set_dtype(args.dtype)
    t_import = time.perf_counter() - t_proc

    sd = StableDiffusion()
    t_w0 = time.perf_counter()
    arena, arena_bytes, t_gen, t_bcast, wstate = build_weight_arena(sd.model.diffusion_model, rank, world, local_rank, args.comm)
    t_weights = time.perf_counter() - t_w0
    seed = 1234 + rank
    B, S = args.images, args.latent
    lat = sd.latent_from_numpy(synth_normal(seed, "sd.latent", (B, 4, S, S)))
    ctx = T.DeviceArray.from_numpy(synth_normal(seed, "sd.context", (B, 77, 768)))
    unc = T.DeviceArray.from_numpy(synth_normal(seed, "sd.uncond", (B, 77, 768)))
    timesteps = list(range(1, 1000, 20))
    ac = sd.alphas_cumprod
    alphas = ac[timesteps]
    alphas_prev = np.concatenate((np.array([1.0]), alphas[:-1])).astype(np.float32)
    if args.tune_cache:
        hip.tf_gemm_tune_load(args.tune_cache.encode())
    t_c0 = time.perf_counter()
    sd.compile(unc, ctx, lat, timesteps=timesteps)
    sd.synchronize()
    t_compile = time.perf_counter() - t_c0
    t_startup = time.perf_counter() - t_proc
    if args.tune_cache and rank == 0:
        hip.tf_gemm_tune_save(args.tune_cache.encode())

    lat0 = T.DeviceArray.empty(lat.shape, np.float32, "row")
    hip.tf_memcpy(lat0.ptr, lat.ptr, lat.nbytes, 3)

    def run(n, eager):
        for s in range(n):
            i = 49 - (s % 50)
            if i == 49 and s > 0:
                # a new 50-step trajectory starts from the initial noise again (64 KB device copy on the step stream): with
                # synthetic weights a latent pushed through several schedules back to back eventually overflows fp16
                hip.tf_memcpy_async(lat.ptr, lat0.ptr, lat.nbytes, 3, sd._stream.handle)
            sd.step(timesteps[i], alphas[i], alphas_prev[i], 7.5, eager=eager)

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()

    run(args.warmup, args.eager)
    sd.synchronize()
    ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
    hip.tf_event_create(ctypes.byref(ev0)); hip.tf_event_create(ctypes.byref(ev1))
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    hip.tf_event_record(ev0, sd._stream.handle)
    run(args.steps, args.eager)
    hip.tf_event_record(ev1, sd._stream.handle)
    sd.synchronize(); torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    ms = ctypes.c_float()
    hip.tf_event_elapsed_ms(ctypes.byref(ms), ev0, ev1)
    from tinyfusers_amd.dist import max_over_ranks
    wall = max_over_ranks(wall, device=f"cuda:{local_rank}")
    # start-up per rank, slowest rank: with N Python drivers on one host (each imports torch, packs nothing but compiles and captures its own
    # graph) host contention shows here first -- the timed region itself has no host work beyond one graph launch per step
    startup = {"startup_s": round(max_over_ranks(t_startup, device=f"cuda:{local_rank}"), 2),
               "import_init_s": round(max_over_ranks(t_import, device=f"cuda:{local_rank}"), 2),
               "weights_s": round(max_over_ranks(t_weights, device=f"cuda:{local_rank}"), 2),
               "compile_capture_s": round(max_over_ranks(t_compile, device=f"cuda:{local_rank}"), 2),
               "note": "max over ranks; wall clock of each rank from interpreter entry of main() to the captured step graph (imports + device init, weight arena synthesis / broadcast, eager warm-up + capture)"}
    final = lat.numpy()
    assert np.isfinite(final).all(), "non-finite latent after the timed steps"

    peak = PEAK_MFMA_TFLOPS[args.dtype]
    roofline = None
    if not args.no_roofline and rank == 0:
        roofline = roofline_leg(sd, run, max(1, min(args.steps, 5)), peak, (B, S, args.dtype) == (1, 64, "fp16"))

    if rank == 0:
        steps_per_s = world * B * args.steps / wall      # image-steps per second (one unit = one denoising step of one image)
        flop_unit = {64: FLOP_PER_STEP, 96: 4.30e12}.get(S)       # algorithmic FLOP per image-step
        rccl_ranks = 1
        if use_dist:
            import torch.distributed as dist
            rccl_ranks = dist.get_world_size()
        dt = {"fp16": "f16", "bf16": "bf16", "fp8": "f8e4m3"}[args.dtype]
        out = {
            "metric": "unet_denoise_steps_per_sec", "value": round(steps_per_s, 2), "unit": "steps/s", "n_gpus": world, "rccl_ranks": rccl_ranks,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dt, "data": "synthetic",
            "config": {"workload": f"SD1.5 UNet single denoise step (CFG batch 2 + DDIM update), {S}x{S}x4 latent, {B} image{'s' if B > 1 else ''} per GPU, {args.dtype} conv/linear, 50-step DDIM schedule",
                       "global_batch": world * B, "latent": [4, S, S], "parallelism": f"dp{world} (batch-sharded, RCCL weight broadcast only, no per-step collective)",
                       "launch": "eager" if args.eager else "hipGraph"},
            "device_ms_per_step": round(ms.value / args.steps, 4),
            "step_tflops": round(flop_unit * B * args.steps / (ms.value * 1e-3) / 1e12, 1) if flop_unit else None,
            "step_mfma_frac": round(flop_unit * B * args.steps / (ms.value * 1e-3) / 1e12 / peak, 4) if flop_unit else None,
            "weights": {"bytes": arena_bytes, "synth_s": round(t_gen, 2), "bcast_s": round(t_bcast, 4), "bcast_via": ("tf_bcast (C-ABI, librccl)" if args.comm == "tf" else "torch.distributed nccl") if use_dist else None},
            "startup": startup,
            "roofline": roofline,
        }
        if world == 1 and (B, S, args.dtype) == (1, 64, "fp16") and not args.no_e2e:
            out["e2e"] = e2e_leg(sd, args.e2e_images, seed)
        if world == 1 and (B, S, args.dtype) == (1, 64, "fp16") and not args.no_config5:
            out["config5"] = config5_leg(wstate, args.config5_steps, seed)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_steps)
        print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
