import os, sys, io, contextlib
os.environ["TF_POOL_POISON"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.storage.state import param_shapes, update_state
from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict
from tinyfusers_amd.variants.sd import StableDiffusion
from tinyfusers_amd.vision.resnet import ResBlock
from tinyfusers_amd.attention.attention import SpatialTransformer, CrossAttention, BasicTransformerBlock
from tinyfusers_amd.vision.conv2d import Conv2d
from tinyfusers_amd.ff.group_norm import GroupNorm
from tinyfusers_amd.ff.nn import GEGLU
from tinyfusers_amd.ff.linear import Linear
T.ensure_init(0)
first = []
def wrap(cls, name):
    orig = cls.__call__
    def f(self, *a, **k):
        y = orig(self, *a, **k)
        ys = y if isinstance(y, (tuple, list)) else (y,)
        for t in ys:
            if isinstance(t, T.DeviceArray) and t.dtype == np.float16:
                arr = t.numpy()
                if not np.isfinite(arr).all() and not first:
                    first.append(name)
                    ins = [x for x in a if isinstance(x, T.DeviceArray)]
                    print("FIRST NON-FINITE OUTPUT:", name, "out", t.shape, "nan frac %.4f" % (1 - np.isfinite(arr).mean()),
                          "| inputs finite:", [bool(np.isfinite(x.numpy()).all()) for x in ins], "kwargs", list(k), flush=True)
        return y
    cls.__call__ = f
for c, n in ((Conv2d, "Conv2d"), (GroupNorm, "GroupNorm"), (CrossAttention, "CrossAttention"), (GEGLU, "GEGLU"), (Linear, "Linear"),
             (ResBlock, "ResBlock"), (BasicTransformerBlock, "BasicTransformerBlock"), (SpatialTransformer, "SpatialTransformer")):
    wrap(c, n)
model = StableDiffusion()
with contextlib.redirect_stdout(io.StringIO()):
    update_state(model, synth_state_dict(param_shapes(model), 0), "")
context = T.DeviceArray.from_numpy(synth_normal(42, "sd.context", (1, 77, 768))); unc = T.DeviceArray.from_numpy(synth_normal(42, "sd.uncond", (1, 77, 768)))
latent = model.latent_from_numpy(synth_normal(42, "sd.latent", (1, 4, 64, 64)))
sp = model._step_params()
model._stream = T.Stream(); model._latent = latent
with T.use_stream(model._stream):
    model._ctx2 = model._stack_context(unc, context)
    for i in range(3):
        sp.set(981.0, 0.5, 0.6, 7.5)
        model._eager_step(sp)
        model._stream.synchronize()
        print("eager step", i, "latent finite", np.isfinite(latent.numpy()).all(), "first", first, flush=True)
