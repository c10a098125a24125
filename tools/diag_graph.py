"""Sampler-loop soak with diagnostics: compile() once, then N images x 50 graph steps + decode (the example/sd1.py flow on
synthetic weights).  Every latent must stay finite; on the first trajectory that does not, probe which of {graph replay,
eager step, a freshly captured graph} reproduces it from the same inputs and whether any persistent array went bad.

    python tools/diag_graph.py [images] [--trace | --determinism]

--trace: every array the step allocates is scanned for non-finite values by a stream-ordered kernel when it is handed back to
the pool (no host sync inside a step, works in graph replays too); the schedule then runs one step at a time and stops at the
first step that flags anything, listing the arrays in allocation order with the Python frames that allocated them.
--determinism: every eager step runs twice from the same latent with a checksum of every array it allocates; the first array
whose checksum differs between the two runs names the kernel that is not a pure function of its inputs.
"""
import contextlib
import gc
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if os.environ.get("TF_DIAG_TORCH"):       # torch bundles its own HIP runtime: loading it first makes libtinyfusers_hip.so bind to that one
    import torch
    if os.environ["TF_DIAG_TORCH"] == "2":
        torch.cuda.init()
import tinyfusers_amd.storage.tensor as T
from tinyfusers_amd.storage.state import param_shapes, update_state
from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict
from tinyfusers_amd.variants.sd import StableDiffusion

import traceback

trace = "--trace" in sys.argv or "--determinism" in sys.argv
determinism = "--determinism" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
images = int(args[0]) if args else 4
T.ensure_init(0)


class Tracer:
    MAX = 16384

    def __init__(self):
        self.flags = T.DeviceArray.zeros((self.MAX,), np.int64 if determinism else np.int32, "row")
        T.hip.tf_device_sync()
        self.live, self.tags, self.armed = {}, [], False
        empty0, free0 = T.DeviceArray.empty, T._pool_free
        me = self

        def empty(shape, dtype=np.float16, layout=None):
            a = empty0(shape, dtype, layout)
            if me.armed and len(me.tags) < me.MAX and np.dtype(dtype) in (np.dtype(np.float16), np.dtype(np.float32)):
                fr = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack(limit=9)[:-1] if "tinyfusers_amd" in f.filename]
                me.live[a.ptr] = (len(me.tags), a.nbytes, int(np.dtype(dtype) == np.float32))
                me.tags.append((a.shape, np.dtype(dtype).name, " < ".join(reversed(fr[-5:]))))
            return a

        def free(ptr, n):
            rec = me.live.pop(ptr, None)
            if rec is not None and me.armed and not sys.is_finalizing():
                if determinism:
                    T.hip.tf_debug_checksum(ptr, rec[1], me.flags.ptr + 8 * rec[0], T._sh())
                else:
                    T.hip.tf_debug_nonfinite(ptr, rec[1], rec[2], me.flags.ptr + 4 * rec[0], T._sh())
            free0(ptr, n)

        T.DeviceArray.empty = staticmethod(empty)
        T._pool_free = free

    def reset(self):
        self.live, self.tags = {}, []

    def clear(self, stream):
        T.hip.tf_memset_async(self.flags.ptr, 0, self.flags.nbytes, stream.handle)

    def report(self, tag):
        T.hip.tf_device_sync()
        f = np.nonzero(self.flags.numpy())[0]
        print(f"   [{tag}] arrays flagged non-finite: {len(f)} of {len(self.tags)} traced")
        for t in f[:14]:
            print(f"      #{t} {self.tags[t][0]} {self.tags[t][1]}  {self.tags[t][2]}")
        return len(f)


tracer = Tracer() if trace else None
model = StableDiffusion()
with contextlib.redirect_stdout(io.StringIO()):
    update_state(model, synth_state_dict(param_shapes(model), 0), "")
context = T.DeviceArray.from_numpy(synth_normal(42, "sd.context", (1, 77, 768)))
unc = T.DeviceArray.from_numpy(synth_normal(42, "sd.uncond", (1, 77, 768)))
timesteps = list(range(1, 1000, 20))
alphas = model.alphas_cumprod[timesteps]
alphas_prev = np.concatenate((np.array([1.0]), alphas[:-1])).astype(np.float32)
latent = model.latent_from_numpy(synth_normal(42, "sd.latent", (1, 4, 64, 64)))
if tracer and not determinism:
    tracer.armed = True
model.compile(unc, context, latent)
if tracer:
    print("traced allocations through compile():", len(tracer.tags), flush=True)
print("HIP runtime:", sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l}), flush=True)


def noise(n):
    return np.ascontiguousarray(synth_normal(42 + n, "sd.latent", (1, 4, 64, 64)))


def upload(n):
    model.set_latent(noise(n))


def finite():
    model.synchronize()
    T.hip.tf_device_sync()
    return bool(np.isfinite(latent.numpy()).all())


def scan(tag):
    bad = []
    for o in gc.get_objects():
        if isinstance(o, T.DeviceArray) and o.dtype in (np.float16, np.float32) and o.size > 0 and o.ptr not in T.pool().owned:
            a = o.numpy()
            if not np.isfinite(a).all():
                bad.append((o.shape, str(o.dtype)))
    print("  ", tag, "non-finite persistent arrays:", bad[:8], len(bad), flush=True)


def one_step(n, **kw):
    upload(n)
    model.step(981, alphas[49], alphas_prev[49], 7.5, **kw)
    return finite()


def determinism_soak():
    bad = 0
    tracer.armed = True
    for n in range(images):
        upload(n)
        for index, timestep in list(enumerate(timesteps))[::-1]:
            finite()
            x0 = latent.numpy().copy()
            runs = []
            for rep in range(2):
                model.set_latent(x0)
                tracer.reset()
                tracer.clear(model._stream)
                model.step(timestep, alphas[index], alphas_prev[index], 7.5, eager=True)
                with T.use_stream(model._stream):
                    model._keep = None
                finite()
                runs.append((tracer.flags.numpy()[:len(tracer.tags)].copy(), latent.numpy().copy(), list(tracer.tags)))
            (s0, l0, t0), (s1, l1, t1) = runs
            diff = np.nonzero(s0 != s1)[0] if len(s0) == len(s1) else np.arange(min(len(s0), len(s1)))
            if len(diff) or not np.array_equal(l0, l1):
                bad += 1
                print(f"image {n} step index {index}: two runs of the same step differ: {len(diff)} of {len(s0)} arrays, latent equal: {np.array_equal(l0, l1)},"
                      f" |latent| max {np.abs(l0).max():.3g} / {np.abs(l1).max():.3g}", flush=True)
                for t in diff[:10]:
                    print(f"      #{t} {t0[t][0]} {t0[t][1]}  {t0[t][2]}")
                if bad >= 4:
                    return bad
        print("image", n, "done, latent finite:", finite(), " |latent| max %.3g" % np.abs(latent.numpy()).max(), flush=True)
    return bad


if determinism:
    bad = determinism_soak()
    tracer.armed = False
    print("NONDETERMINISTIC" if bad else "DETERMINISTIC", flush=True)
    sys.exit(1 if bad else 0)

failed = 0
for n in range(images):
    upload(n)
    for index, timestep in list(enumerate(timesteps))[::-1]:
        if tracer:
            tracer.clear(model._stream)
        model.step(timestep, alphas[index], alphas_prev[index], 7.5)
        if tracer and (not finite() or np.any(tracer.flags.numpy())):
            print("image", n, "step index", index, "latent finite:", finite())
            tracer.report("graph replay")
            for k in range(3):
                upload(n)
                tracer.clear(model._stream)
                first = len(tracer.tags)
                model.step(981, alphas[49], alphas_prev[49], 7.5, eager=True)
                print("   eager step from fresh noise finite:", finite(), "(tags from #%d)" % first)
                tracer.report("eager step %d" % k)
            break
    ok = finite()
    print("image", n, "latent finite:", ok, flush=True)
    if not ok:
        failed += 1
        scan("after the failing trajectory:")
        print("   one graph step from fresh noise finite:", one_step(n))
        print("   again:", one_step(n))
        print("   one eager step finite:", one_step(n, eager=True))
        print("   graph step after the eager one:", one_step(n))
        model.compile(unc, context, latent)
        print("   graph step after re-capture:", one_step(n))
        upload(n)
        for index, timestep in list(enumerate(timesteps))[::-1]:
            model.step(timestep, alphas[index], alphas_prev[index], 7.5)
        print("   full trajectory with the new graph finite:", finite())
        break
    with T.use_stream(model._stream):
        img = model.decode(latent)
    print("   decoded image mean %.1f" % img.mean(), flush=True)
if tracer:
    tracer.armed = False
print("FAILED" if failed else "OK", flush=True)
sys.exit(1 if failed else 0)
