import os, sys, io, contextlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.storage.state import param_shapes, update_state
from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict
from tinyfusers_amd.variants.sd import StableDiffusion
mode = sys.argv[1]
T.ensure_init(0)
model = StableDiffusion()
with contextlib.redirect_stdout(io.StringIO()):
    update_state(model, synth_state_dict(param_shapes(model), 0), "")
rng = np.random.default_rng(42)
prompt = np.full((1, 77), 49407, dtype=np.int64); prompt[0, 0] = 49406; prompt[0, 1:10] = rng.integers(0, 49406, 9)
empty = np.full((1, 77), 49407, dtype=np.int64); empty[0, 0] = 49406
tm = model.cond_stage_model.transformer.text_model
if mode.startswith("synth"):
    context = T.DeviceArray.from_numpy(synth_normal(42, "sd.context", (1, 77, 768))); unc = T.DeviceArray.from_numpy(synth_normal(42, "sd.uncond", (1, 77, 768)))
else:
    tm(prompt); T.hip.tf_stream_sync(None)
    context, unc = tm(prompt), tm(empty)
    T.hip.tf_stream_sync(None)
print("ctx absmax", np.abs(context.numpy()).max(), np.abs(unc.numpy()).max())
timesteps = list(range(1, 1000, 20)); alphas = model.alphas_cumprod[timesteps]
alphas_prev = np.concatenate((np.array([1.0]), alphas[:-1])).astype(np.float32)
latent = model.latent_from_numpy(synth_normal(42, "sd.latent", (1, 4, 64, 64)))
model.compile(unc, context, latent)
eager = mode.endswith("eager")
for n in range(4):
    T.hip.tf_memcpy(latent.ptr, np.ascontiguousarray(synth_normal(42 + n, "sd.latent", (1, 4, 64, 64))).ctypes.data, latent.nbytes, 1)
    bad = None
    for index, timestep in list(enumerate(timesteps))[::-1]:
        model.step(timestep, alphas[index], alphas_prev[index], 7.5, eager=eager)
        if mode.find("check") >= 0:
            model.synchronize()
            l = latent.numpy()
            if not np.isfinite(l).all() and bad is None:
                bad = index
    model.synchronize()
    l = latent.numpy()
    print(mode, n, "final finite", np.isfinite(l).all(), "first bad step", bad, flush=True)
