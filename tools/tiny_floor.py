import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.native import hip
from tinyfusers_amd.storage.state import param_shapes, update_state
from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict
from tinyfusers_amd.variants.sd import StableDiffusion
from tinyfusers_amd.vision.unet import TINY
T.ensure_init(0)
sd = StableDiffusion(TINY)
update_state(sd.model.diffusion_model, synth_state_dict(param_shapes(sd.model.diffusion_model), 5), "")
lat = sd.latent_from_numpy(synth_normal(5, "lat", (1, 4, 16, 16)))
ctx = T.DeviceArray.from_numpy(synth_normal(5, "c", (1, 13, 64))); unc = T.DeviceArray.from_numpy(synth_normal(5, "u", (1, 13, 64)))
sd.compile(unc, ctx, lat)
for eager in (False, True):
    for _ in range(5): sd.step(981.0, 0.5, 0.6, 7.5, eager=eager)
    sd.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n): sd.step(981.0, 0.5, 0.6, 7.5, eager=eager)
    sd.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(("eager" if eager else "graph"), "tiny step: %.3f ms" % (dt * 1e3))
