import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import tinyfusers_amd.storage.tensor as T
T.ensure_init(0)
from test_gpu_fp8 import rnd, raw
from oracle import fp8 as O8, ops as O
from tinyfusers_amd.ff import fp8
from tinyfusers_amd.ff.layer_norm import LayerNorm
from tinyfusers_amd.ff.nn import pack_geglu
m, c = 128, 320
x = rnd("f8.x", (1, m, c), 1.5) + 0.1
w1 = rnd("f8.w1", (8 * c, c), c ** -0.5); b1 = rnd("f8.b1", (8 * c,), 0.1)
dv = lambda a: T.DeviceArray.from_numpy(a, np.float16, "row")
ln = LayerNorm(c, init=False); ln.weight = dv(1 + rnd("f8.g", (c,), 0.1)); ln.bias = dv(rnd("f8.bt", (c,), 0.1))
h8 = fp8.layer_norm_fp8(dv(x), ln); hq = raw(h8).reshape(1, m, c)
wp, bp = pack_geglu(dv(w1), dv(b1)); w8, sc = fp8.pack_weight(wp, {})
for rep in range(3):
    hid8 = fp8.linear_fp8(h8, w8, sc, bp, act=1, out_features=4 * c, out_fp8=True)
    hidq = raw(hid8).reshape(m, 4 * c)
    hid16 = fp8.linear_fp8(h8, w8, sc, bp, act=1, out_features=4 * c, out_fp8=False).numpy().reshape(m, 4 * c)
    w1q = O8.quant_weight(w1)[0]
    lin = O.linear(torch.from_numpy(hq), w1q, b1).numpy().reshape(m, 8 * c)
    a, g = lin[:, :4 * c], lin[:, 4 * c:]
    want = (torch.from_numpy(a) * O.gelu(torch.from_numpy(g))).numpy()
    bad = np.argwhere(np.abs(hidq - want) > 0.1 * np.abs(want) + 5e-3)
    bad16 = np.argwhere(np.abs(hid16 - want) > 0.02 * np.abs(want) + 2e-3)
    print("rep", rep, "bad8", len(bad), "bad16", len(bad16))
    for r, cc in bad[:5]:
        print("  ", r, cc, "got8", hidq[r, cc], "got16", hid16[r, cc], "want", want[r, cc], "a", a[r, cc], "g", g[r, cc], "q(want)", float(O8.quant_act(np.array([want[r, cc]]))[0]))
print("---- weights")
deq = raw(w8).reshape(8 * c, c) * sc.numpy()[:, None]
# un-interleave packed rows back to [values ; gates]
n = 4 * c
idx = np.arange(2 * n).reshape(n // 16, 2, 16)
order = np.concatenate([idx[:, 0, :].reshape(-1), idx[:, 1, :].reshape(-1)])     # packed row index of logical row r
deq_l = deq[order]
wq, s = O8.quant_weight(w1)
wq = wq.numpy()
print("scale mismatch", np.abs(sc.numpy()[order] - s.numpy()).max())
d = np.abs(deq_l - wq)
print("weight mismatches", (d > 1e-7).sum(), "of", d.size, "max", d.max())
rows = np.unique(np.argwhere(d > 1e-7)[:, 0])
print("rows with mismatches", rows[:20], len(rows))
r = 1280 + 199
print("row", r, "mism", (d[r] > 1e-7).sum(), "w1 row max", np.abs(w1[r]).max(), "scale", s[r].item(), sc.numpy()[order][r])
bq = bp.numpy()[order]
print("bias mismatch", np.abs(bq - b1).max(), np.argmax(np.abs(bq - b1)))
