#!/usr/bin/env python3
"""UNet denoising steps/sec, SD1.5 512x512 (64x64x4 latent), one image per GPU, CFG batch 2, fp16.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU.  Rank 0 generates the synthetic weights and broadcasts the packed fp16 weight arena once
over RCCL (torch.distributed "nccl"); there is NO per-step collective: every rank runs its own image's DDIM
trajectory (weak scaling, value = all images' steps / max-over-ranks time).

A "step" = one StableDiffusion.__call__ of the reference (variants/sd.py:56-59): CFG duplicate, one UNet forward
at batch 2, CFG combine + DDIM update, replayed as one HIP graph.  Inputs are resident in HBM before timing.
Extra objects on the JSON line: "roofline" (the implicit-GEMM conv/linear kernel family, timed per launch with
HIP events on its own stream in an instrumented eager pass) and "cpu_baseline" (the CPU oracle = the torch-CPU op
path the reference's tests compare against, timed on this box's host cores; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = 1.6088e12          # algorithmic FLOPs of one step (conv 887.89 G + linear 466.49 G + SDPA 252.10 G + 2.28 G)
PEAK_MFMA_TFLOPS = 2500.0          # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of the HIP-graph replay")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--tune-cache", default="", help="file to load/save the GEMM autotuner's per-shape choices (optional)")
    ap.add_argument("--images", type=int, default=1, help="images per GPU (UNet batch = 2x); default 1 = BASELINE config 2")
    ap.add_argument("--latent", type=int, default=64, help="latent height = width; default 64 (512x512 images); 96 = 768x768 (config 5)")
    return ap.parse_args()


def build_weight_arena(unet, rank, world, device_index):
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
    arena = torch.empty(off, dtype=torch.uint8, device=f"cuda:{device_index}")
    base = arena.data_ptr()
    t0 = time.time()
    if rank == 0:
        from concurrent.futures import ThreadPoolExecutor

        def gen(k):
            return k, pack_tensor(synth_tensor(0, k, shapes[k]))
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            for k, w in ex.map(gen, list(shapes)):
                hip.tf_memcpy(base + offs[k], w.ctypes.data, w.nbytes, 1)
    t_gen = time.time() - t0
    t_bcast = 0.0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.time()
        broadcast_arena(arena, src=0)          # the ONLY collective of the path (RCCL over xGMI)
        torch.cuda.synchronize()
        t_bcast = time.time() - t1
    state = {k: DeviceArray(base + offs[k], shapes[k], np.float16, None, base=arena) for k in shapes}
    update_state(unet, state, "")
    return arena, off, t_gen, t_bcast


def cpu_baseline(steps):
    """The reference's CPU op path (F.conv2d / GroupNorm / F.layer_norm / F.linear / SDPA, fp32) = oracle.sd_step,
    on a bounded sample: `steps` full denoising steps of the same workload after one warm-up."""
    import torch
    import oracle
    from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict
    cores = min(os.cpu_count() or 1, 16)
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
    return {"value": 1.0 / sec, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} full SD1.5 denoising steps (CFG batch 2, 64x64 latent) after 1 warm-up, torch-CPU fp32 oracle, median {sec:.2f} s/step"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    assert world == args.gpus or not use_dist, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)

    import tinyfusers_amd.storage.tensor as T
    from tinyfusers_amd.native import hip, lib
    from tinyfusers_amd.storage.synth import synth_normal
    from tinyfusers_amd.variants.sd import StableDiffusion
    import ctypes
    T.ensure_init(local_rank)

    sd = StableDiffusion()
    arena, arena_bytes, t_gen, t_bcast = build_weight_arena(sd.model.diffusion_model, rank, world, local_rank)
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
    sd.compile(unc, ctx, lat)
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
    final = lat.numpy()
    assert np.isfinite(final).all(), "non-finite latent after the timed steps"

    roofline = None
    if not args.no_roofline and rank == 0:
        # dominant kernel family = k_igemm (conv2d + linear, 84 % of the step's FLOPs): per-launch HIP events
        n_inst = max(1, min(args.steps, 5))
        lib.tf_prof_enable(1)
        run(n_inst, True)
        sd.synchronize()
        gms, gfl, gl = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
        hip.tf_prof_read(ctypes.byref(gms), ctypes.byref(gfl), ctypes.byref(gl))
        lib.tf_prof_enable(0)
        ach = gfl.value / (gms.value * 1e-3) / 1e12 if gms.value > 0 else 0.0
        # HBM bytes per launch of this kernel family from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 per the
        # gfx950 correction + WRITE_SIZE, separate runs of this same command; tools/pmc_summary.py): cannot be sampled live
        traffic = None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            traffic = round(pj["k_igemm"]["hbm_bytes_per_launch"])
        except Exception:
            pass
        roofline = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_TFLOPS, 4),
                    "traffic": traffic, "traffic_unit": "HBM bytes per launch (profiles/r01_pmc_traffic.json)",
                    "kernel": "k_igemm<BM,BN> (implicit-GEMM conv2d + linear)",
                    "launches_per_step": gl.value / n_inst, "avg_launch_us": round(gms.value * 1e3 / max(1, gl.value), 2),
                    "gemm_ms_per_step": round(gms.value / n_inst, 4), "gemm_flop_per_step": gfl.value / n_inst}

    if rank == 0:
        steps_per_s = world * B * args.steps / wall      # image-steps per second (one unit = one denoising step of one image)
        flop_unit = {64: FLOP_PER_STEP, 96: 4.30e12}.get(S)       # algorithmic FLOP per image-step
        out = {
            "metric": "unet_denoise_steps_per_sec", "value": round(steps_per_s, 2), "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"SD1.5 UNet single denoise step (CFG batch 2 + DDIM update), {S}x{S}x4 latent, {B} image{'s' if B > 1 else ''} per GPU, fp16, 50-step DDIM schedule",
                       "global_batch": world * B, "latent": [4, S, S], "parallelism": f"dp{world} (batch-sharded, RCCL weight broadcast only, no per-step collective)",
                       "launch": "eager" if args.eager else "hipGraph"},
            "device_ms_per_step": round(ms.value / args.steps, 4),
            "step_tflops": round(flop_unit * B * args.steps / (ms.value * 1e-3) / 1e12, 1) if flop_unit else None,
            "step_mfma_frac": round(flop_unit * B * args.steps / (ms.value * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS, 4) if flop_unit else None,
            "weights": {"bytes": arena_bytes, "synth_s": round(t_gen, 2), "bcast_s": round(t_bcast, 4)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_steps)
        print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
