#!/usr/bin/env python3
"""Which conv / linear sites of the UNet may run on e4m3 operands within BASELINE config 5's gate (UNet rel-L2 <= 0.1 against the fp32
oracle)?  CPU study on the oracle (TEST INFRASTRUCTURE: lives under tests/ because it imports oracle/): every site class is switched to
the emulated fp8 GEMM -- e4m3 weights with one scale per output channel, e4m3 activations either at the fixed scale 1 ("unit", round 3)
or with one power-of-two E8M0 scale per 32 consecutive channels of a pixel / token ("mx", round 4: oracle.fp8.quant_act_mx) -- and the
UNet output is compared with the fp32 forward.  usage: python tests/fp8_policy_study.py [latent side, default 32]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle import fp8 as F8, ops
from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict

SITES = {
    "res3x3": lambda n: n.endswith((".in_layers.2.weight", ".out_layers.3.weight")),
    "ff1": lambda n: n.endswith(".ff.net.0.proj.weight"),
    "ff2": lambda n: n.endswith(".ff.net.2.weight"),
    "qkv1": lambda n: ".attn1.to_" in n and not n.endswith("to_out.0.weight"),
    "q2": lambda n: n.endswith(".attn2.to_q.weight"),
    "to_out": lambda n: n.endswith(".to_out.0.weight"),
    "proj_in": lambda n: n.endswith(".proj_in.weight"),
    "proj_out": lambda n: n.endswith(".proj_out.weight"),
    "skip1x1": lambda n: n.endswith(".skip_connection.weight"),
    "updown": lambda n: n.endswith((".op.weight", ".conv.weight")),
}


def run(side=32, seed=0):
    torch.set_num_threads(os.cpu_count() or 1)
    W = {k: torch.from_numpy(v.astype(np.float32)) for k, v in synth_state_dict(oracle.unet_param_shapes(oracle.SD15), 0).items()}
    names = {id(v): k for k, v in W.items()}
    x = synth_normal(1234, "sd.latent", (1, 4, side, side)); x = np.concatenate([x, x])
    ctx = np.concatenate([synth_normal(1234, "sd.uncond", (1, 77, 768)), synth_normal(1234, "sd.context", (1, 77, 768))])
    ref = oracle.unet_forward(x, np.array([981.0], np.float32), ctx, W).numpy()
    conv0, lin0 = ops.conv_2d, ops.linear
    wq = {}

    def qw(w):
        if id(w) not in wq:
            wq[id(w)] = F8.quant_weight(w)[0]
        return wq[id(w)]

    def study(active, act_mode, min_k=0):
        def site(w):
            n = names.get(id(w))
            if n is None:
                return None
            for s in active:
                if SITES[s](n):
                    return s
            return None

        def qa(t, channel_dim):
            return F8.quant_act(t) if act_mode == "unit" else F8.quant_act_mx(t, channel_dim)

        def conv(x, w, padding, stride, dilation):
            if site(w) and w.shape[1] % 64 == 0 and w.shape[1] >= min_k:
                return conv0(qa(ops.as_t(x), 1), qw(w), padding, stride, dilation)
            return conv0(x, w, padding, stride, dilation)

        def lin(x, w, b=None):
            if site(w) and w.shape[1] % 64 == 0 and w.shape[1] >= min_k:
                return lin0(qa(ops.as_t(x), -1), qw(w), b)
            return lin0(x, w, b)
        ops.conv_2d, ops.linear = conv, lin
        try:
            t0 = time.time()
            y = oracle.unet_forward(x, np.array([981.0], np.float32), ctx, W).numpy()
        finally:
            ops.conv_2d, ops.linear = conv0, lin0
        return float(np.linalg.norm(y - ref) / np.linalg.norm(ref)), time.time() - t0

    r3 = ["res3x3", "ff1", "ff2"]
    cases = [
        ("round-3 policy (res3x3 + ff, K >= 640 for ff), unit", r3, "unit"),
        ("same sites, mx", r3, "mx"),
        ("+ qkv1 + q2 (LayerNorm outputs)", r3 + ["qkv1", "q2"], "mx"),
        ("+ proj_in (GroupNorm output)", r3 + ["qkv1", "q2", "proj_in"], "mx"),
        ("+ to_out (attention output)", r3 + ["qkv1", "q2", "proj_in", "to_out"], "mx"),
        ("+ proj_out (residual stream)", r3 + ["qkv1", "q2", "proj_in", "to_out", "proj_out"], "mx"),
        ("+ skip1x1", r3 + ["qkv1", "q2", "proj_in", "to_out", "proj_out", "skip1x1"], "mx"),
        ("+ updown = every conv / linear with Cin % 64 == 0", list(SITES), "mx"),
        ("every site, unit scale", list(SITES), "unit"),
    ]
    for s in SITES:
        cases.append((f"only {s}, mx", [s], "mx"))
    for label, active, mode in cases:
        r, dt = study(active, mode)
        print(f"{r:8.4f}  {label}   ({dt:.1f} s)", flush=True)


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
