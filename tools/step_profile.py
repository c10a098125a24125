#!/usr/bin/env python3
"""Per-shape GEMM/conv timing of one real SD1.5 step (HIP events around every k_igemm launch, eager)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip, lib
from tinyfusers_amd.storage.synth import synth_normal
from tinyfusers_amd.variants.sd import StableDiffusion
from bench import build_weight_arena

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_shapes.csv"
T.ensure_init(0)
sd = StableDiffusion()
arena = build_weight_arena(sd.model.diffusion_model, 0, 1, 0)
lat = sd.latent_from_numpy(synth_normal(1234, "sd.latent", (1, 4, 64, 64)))
ctx = T.DeviceArray.from_numpy(synth_normal(1234, "sd.context", (1, 77, 768)))
unc = T.DeviceArray.from_numpy(synth_normal(1234, "sd.uncond", (1, 77, 768)))
sd.compile(unc, ctx, lat)
lib.tf_prof_enable(1)
for i in range(5):
    sd.step(981.0, 0.5, 0.6, 7.5, eager=True)
sd.synchronize()
hip.tf_prof_dump(out.encode())
rows = [l.strip().split(",") for l in open(out)][1:]
rows.sort(key=lambda r: -float(r[9]))
tot = sum(float(r[9]) for r in rows)
print("total gemm ms/step: %.3f" % (tot / 5))
print("%6s %6s %6s %4s %8s %3s %3s %5s %9s %8s %7s" % ("M", "N", "K", "taps", "tile", "sk", "var", "n/st", "ms/step", "avg_us", "TF/s"))
for r in rows:
    print("%6s %6s %6s %4s %8s %3s %3s %5d %9.3f %8s %7s" % (r[0], r[1], r[2], r[3], r[4] + "x" + r[5], r[6], r[7], int(r[8]) // 5, float(r[9]) / 5, r[10], r[11]))
