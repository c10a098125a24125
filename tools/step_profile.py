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

# usage: step_profile.py [out.csv [images [latent [fp16|fp8 [tune-cache]]]]]
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_shapes.csv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
from tinyfusers_amd import config
config.set_dtype(sys.argv[4] if len(sys.argv) > 4 else "fp16")
if len(sys.argv) > 5:
    hip.tf_gemm_tune_load(sys.argv[5].encode())
T.ensure_init(0)
sd = StableDiffusion()
arena = build_weight_arena(sd.model.diffusion_model, 0, 1, 0)
lat = sd.latent_from_numpy(synth_normal(1234, "sd.latent", (B, 4, S, S)))
ctx = T.DeviceArray.from_numpy(synth_normal(1234, "sd.context", (B, 77, 768)))
unc = T.DeviceArray.from_numpy(synth_normal(1234, "sd.uncond", (B, 77, 768)))
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
