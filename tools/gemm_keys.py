#!/usr/bin/env python3
"""GPU box: which rows of the GEMM tuning table do BASELINE config 2 (1 image, 64 x 64, fp16) and config 5 (4 images, 96 x 96; fp16 and the
fp8 policy) consult?  Runs one eager UNet step of each in TABLE-ONLY mode (tf_gemm_autotune(2): a shape without a row is an error, as on every
rank of a multi-GPU run) with tf_gemm_tune_trace on and writes the keys to tests/golden/gemm_keys.json -- the fixture the world-2 CPU test
checks the shipped table against (tests/test_dist_cpu.py).   usage: python tools/gemm_keys.py [out.json]   (--tune: mode 1, report what was missing)"""
import contextlib, io, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd import config
from tinyfusers_amd.native import hip, lib
from tinyfusers_amd.storage.state import unet_param_shapes, update_state
from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict
from tinyfusers_amd.variants.sd import StableDiffusion

tune = "--tune" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
out_path = args[0] if args else os.path.join(ROOT, "tests", "golden", "gemm_keys.json")
T.ensure_init(0)
W = None
res, missing, failed = {}, {}, []
for name, B, S, dt in (("config2_fp16", 1, 64, "fp16"), ("config5_fp16", 4, 96, "fp16"), ("config5_fp8", 4, 96, "fp8")):
    config.set_dtype(dt)
    sd = StableDiffusion(init=False)
    if W is None:
        W = synth_state_dict(unet_param_shapes(sd.model.diffusion_model), 0)
    with contextlib.redirect_stdout(io.StringIO()):
        update_state(sd.model.diffusion_model, W, "")
    lat = sd.latent_from_numpy(synth_normal(1234, "sd.latent", (B, 4, S, S)))
    ctx = T.DeviceArray.from_numpy(synth_normal(1234, "sd.context", (B, 77, 768)))
    unc = T.DeviceArray.from_numpy(synth_normal(1234, "sd.uncond", (B, 77, 768)))
    def one_step(mode):
        hip.tf_gemm_autotune(mode)
        hip.tf_gemm_tune_trace(1)
        try:
            sd(unc, ctx, lat, np.array([981]), sd.alphas_cumprod[981:982], sd.alphas_cumprod[961:962], np.array([7.5]))
            T.hip.tf_stream_sync(None)
        finally:
            hip.tf_gemm_tune_trace(0)
            hip.tf_gemm_autotune(1)
    try:
        one_step(1 if tune else 2)
    except RuntimeError as e:                              # table-only mode met a shape without a row: say so, then collect every key with tuning on
        print(f"{name}: {e}", flush=True)
        failed.append(name)
        one_step(1)
    tmp = out_path + ".tmp"
    hip.tf_gemm_tune_trace_dump(tmp.encode())
    rows = [[int(v) for v in ln.split()] for ln in open(tmp)]
    os.remove(tmp)
    res[name] = [r[:10] for r in rows]
    missing[name] = [r[:10] for r in rows if not r[10]]
    print(f"{name}: {len(rows)} shape keys, {len(missing[name])} without a row", flush=True)
    del sd
config.set_dtype("fp16")
json.dump(res, open(out_path, "w"))
if tune or failed:
    hip.tf_gemm_tune_save((out_path + ".tuned.txt").encode())
    print("table with the newly tuned rows:", out_path + ".tuned.txt")
print("wrote", out_path)
sys.exit(1 if failed else 0)
