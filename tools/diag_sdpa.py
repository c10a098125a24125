"""Per-row diagnostic of the fused SDPA kernel against the CPU oracle (GPU box only):  python tools/diag_sdpa.py

For a handful of d = 40 shapes (the UNet's 64 x 64 level: self-attention 4096 x 4096, cross-attention 4096 x 77, plus small ragged
ones) prints the max error, the per-row scale factor <got, want> / <want, want> (a row whose softmax denominator or running maximum
went wrong shows as a factor != 1 while its direction is still right) and the first rows that are off by more than 2 %.  Written to
run down the 16-deep-tail variant of round 2 (wrong rows at Tk = 77); kept as the first thing to run when an SDPA test fails."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import tinyfusers_amd.storage.tensor as T
T.ensure_init(0)
from test_gpu_ops import rnd, dev
from oracle import ops as O
from tinyfusers_amd.attention.sdpa import scaled_dot_product_attention
for (b, nh, tq, tk, hs) in [(1, 2, 4096, 4096, 40), (1, 2, 128, 128, 40), (1, 2, 100, 77, 40), (2, 8, 4096, 77, 40), (2, 8, 4096, 4096, 40)]:
    q, k, v = rnd("sdpa.q", (b, nh, tq, hs)), rnd("sdpa.k", (b, nh, tk, hs)), rnd("sdpa.v", (b, nh, tk, hs))
    got = scaled_dot_product_attention(dev(T, q, "row"), dev(T, k, "row"), dev(T, v, "row")).numpy()
    want = O.scaled_dot_product_attention(q, k, v).numpy()
    err = np.abs(got - want)
    ratio = (got * want).sum(-1) / np.maximum((want * want).sum(-1), 1e-9)       # per-row scale factor
    print((b, nh, tq, tk, hs), "max err %.4f" % err.max(), "row scale min/max %.4f %.4f" % (ratio.min(), ratio.max()), "bad rows", int((np.abs(ratio - 1) > 0.02).sum()), "of", ratio.size,
          "first bad", np.argwhere(np.abs(ratio - 1) > 0.02)[:4].tolist())
