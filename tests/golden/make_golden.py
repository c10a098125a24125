#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python (read-only at /root/reference).

Runs only in the authoring container (the reference never travels to the GPU box).  The reference
cannot be imported as-is (cupy / cudnn / libcuda are absent and it has import defects),
so it is imported under stand-in modules, exactly as SURVEY 8(c) describes:
  * ``cupy``  -> numpy namespace (+ asnumpy, single, random.uniform(lo,hi,size,dtype), no-op
                 stream/device sync, RawModule(...).get_function(name) -> numpy row softmax that writes
                 ``att`` in place with the (grid, block, args, shared_mem) call signature);
  * ``cudnn`` -> create_handle() only; the two cuDNN-graph functions are replaced by the torch
                 functions the reference's own tests use as ground truth
                 (vision.conv2d.conv_2d -> F.conv2d, tests/conv2d.py:27;
                  ff.layer_norm.layer_norm -> F.layer_norm, tests/layer_norm.py:38);
  * ``tinyfusers.native*`` -> empty modules (ctypes CDLL('libcuda.so') cannot load);
  * ``tinyfusers.tensor.tensor`` aliased to ``tinyfusers.storage.tensor`` (D1).
Nothing from the reference is written anywhere: outputs are numeric arrays only.

Usage:  python tests/golden/make_golden.py [ops] [blocks] [unet] [unet50] [vae] [clip] [tokenizer]

``unet50`` runs the reference's StableDiffusion.__call__ over the WHOLE schedule of example/sd1.py:54-73 (50 steps) and keeps the
latent after every tenth step and the final one (unet50_sd15.npz).  For that long run the cupy stand-in sends the five array functions
that dominate the reference glue's CPU time (exp, tanh, matmul, dot, sqrt on large fp32 arrays) through multi-threaded torch instead of
single-threaded numpy -- ON by default when ``unet50`` is among the arguments (that is how the committed unet50_sd15.npz was produced),
off otherwise; TF_GOLDEN_FAST=0 / 1 overrides.  The reference's code is what runs either way.
"""
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True


def install_stubs():
    cp = types.ModuleType("cupy")
    for n in dir(np):
        if not n.startswith("_"):
            try:
                setattr(cp, n, getattr(np, n))
            except Exception:
                pass
    cp.asnumpy = lambda a: np.asarray(a)
    cp.single = np.single
    if os.environ.get("TF_GOLDEN_FAST", "1" if "unet50" in sys.argv[1:] else "0") != "0":
        def _big(a):
            return isinstance(a, np.ndarray) and a.dtype == np.float32 and a.size >= (1 << 16)

        def _unary(name):
            npf, tf_ = getattr(np, name), getattr(torch, name)

            def f(a, *r, **k):
                if _big(a) and not r and not k:
                    return tf_(torch.from_numpy(np.ascontiguousarray(a))).numpy()
                return npf(a, *r, **k)
            return f
        for name in ("exp", "tanh", "sqrt"):
            setattr(cp, name, _unary(name))

        def _mm(npf):
            def f(a, b, *r, **k):
                if _big(a) and isinstance(b, np.ndarray) and b.dtype == np.float32 and not r and not k and a.ndim >= 2 and b.ndim >= 2:
                    return torch.matmul(torch.from_numpy(np.ascontiguousarray(a)), torch.from_numpy(np.ascontiguousarray(b))).numpy()
                return npf(a, b, *r, **k)
            return f
        cp.matmul = _mm(np.matmul)
        cp.dot = _mm(np.dot)
    rnd = types.ModuleType("cupy.random")
    rnd.uniform = lambda lo, hi, size=None, dtype=np.float32: np.random.uniform(lo, hi, size).astype(dtype)
    rnd.randn = lambda *s: np.random.randn(*s)
    cp.random = rnd

    class _Stream:
        def use(self): return self
        def synchronize(self): pass

    class _Dev:
        def synchronize(self): pass

    cuda = types.ModuleType("cupy.cuda")
    cuda.get_current_stream = lambda: _Stream()
    cuda.Device = lambda *a: _Dev()
    cuda.runtime = types.SimpleNamespace(deviceSynchronize=lambda: None)
    cp.cuda = cuda

    def _softmax(grid=None, block=None, args=None, shared_mem=0):
        att, pre, n, c = args
        x = np.asarray(pre, dtype=np.float32).reshape(n, c)
        m = x.max(axis=-1, keepdims=True)
        e = np.exp(x - m)
        att[...] = (e / e.sum(axis=-1, keepdims=True)).reshape(att.shape)

    class _RawModule:
        def __init__(self, *a, **k): pass
        def get_function(self, name): return _softmax
    cp.RawModule = _RawModule
    sys.modules["cupy"] = cp
    sys.modules["cupy.random"] = rnd
    sys.modules["cupy.cuda"] = cuda

    cudnn = types.ModuleType("cudnn")
    cudnn.create_handle = lambda: None
    sys.modules["cudnn"] = cudnn

    for name in ("tinyfusers.native", "tinyfusers.native.cuda", "tinyfusers.native.cuda.ops",
                 "tinyfusers.native.nvrtc", "tinyfusers.native.nvrtc.ops",
                 "tinyfusers.native.cublas", "tinyfusers.native.cublas.ops"):
        m = types.ModuleType(name)
        m.cuda = m.cudart = m.nvrtc = m.cublas = None
        m.__path__ = []
        sys.modules[name] = m


def import_reference():
    install_stubs()
    os.chdir(REF)                       # attention/sdpa.py:7-14 opens its .cu relative to cwd
    sys.path.insert(0, REF)
    import tinyfusers.storage.tensor as st     # noqa
    sys.modules["tinyfusers.tensor"] = types.ModuleType("tinyfusers.tensor")
    sys.modules["tinyfusers.tensor.tensor"] = st
    import tinyfusers.vision.conv2d as vc
    import tinyfusers.ff.layer_norm as fl

    def conv_2d(X, W, padding, stride, dilation):
        return F.conv2d(torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)),
                        torch.from_numpy(np.ascontiguousarray(W, dtype=np.float32)), None,
                        tuple(stride), tuple(padding), tuple(dilation)).numpy()

    def layer_norm(x, scale, bias, eps):
        c = x.shape[-1]
        return F.layer_norm(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)), (c,),
                            torch.from_numpy(np.ascontiguousarray(scale.reshape(c), dtype=np.float32)),
                            torch.from_numpy(np.ascontiguousarray(bias.reshape(c), dtype=np.float32)),
                            float(np.asarray(eps).reshape(-1)[0])).numpy()
    vc.conv_2d = conv_2d
    fl.layer_norm = layer_norm
    import tinyfusers.attention.sdpa      # noqa  (reads its .cu relative to cwd at import)
    import tinyfusers.attention.attention  # noqa
    import tinyfusers.vision.unet          # noqa
    os.chdir(REPO)
    sys.path.insert(0, REPO)


def install(obj, weights, prefix=""):
    """Fill a reference module tree through the reference's own update_state walk."""
    from tinyfusers.storage.state import update_state
    sd = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in weights.items()}
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        update_state(obj, sd, prefix)
    # bias-free Linears (to_q/to_k/to_v) carry bias=None and are reported as skipped (state.py:17-19)
    skipped = [l for l in buf.getvalue().splitlines() if l.startswith("skipped")
               and not l.endswith((".to_q.bias", ".to_k.bias", ".to_v.bias"))]
    assert not skipped, skipped[:5]


def synth(shapes, seed=0):
    from tinyfusers_amd.storage.synth import synth_state_dict
    return {k: v.astype(np.float32) for k, v in synth_state_dict(shapes, seed).items()}


def rnd(name, shape, std=1.0, seed=7):
    from tinyfusers_amd.storage.synth import synth_normal
    return synth_normal(seed, name, shape, std)


def gen_ops():
    from tinyfusers.vision.conv2d import conv_2d, Conv2d
    from tinyfusers.vision.unet import Upsample, Downsample, timestep_embedding
    from tinyfusers.ff.group_norm import group_norm, GroupNorm
    from tinyfusers.ff.layer_norm import LayerNorm
    from tinyfusers.ff.linear import Linear
    from tinyfusers.ff.nn import GEGLU, FeedForward
    from tinyfusers.attention.sdpa import scaled_dot_product_attention
    from tinyfusers.storage.tensor import Tensor
    from tinyfusers.variants.sd import StableDiffusion, get_alphas_cumprod
    G = {}
    # conv_2d (vision/conv2d.py:9) -- 3x3 s1 p1, 3x3 s2 p1, 1x1
    x = rnd("conv.x", (2, 8, 9, 7)); w3 = rnd("conv.w3", (6, 8, 3, 3), 0.2); w1 = rnd("conv.w1", (6, 8, 1, 1), 0.3)
    G.update(conv_x=x, conv_w3=w3, conv_w1=w1,
             conv_y_3x3_s1=conv_2d(x, w3, [1, 1], [1, 1], [1, 1]),
             conv_y_3x3_s2=conv_2d(x, w3, [1, 1], [2, 2], [1, 1]),
             conv_y_1x1=conv_2d(x, w1, [0, 0], [1, 1], [1, 1]))
    m = Conv2d(8, 6, [3, 3], padding=[1, 1]); m.weight = w3; m.bias = rnd("conv.b", (6,), 0.1)
    G.update(conv_b=m.bias, conv_y_module=m(x))
    # tests/conv2d.py shape family (2x2 kernel, pad 0), down-scaled
    xt = rnd("convt.x", (1, 2, 50, 40)); wt = rnd("convt.w", (1, 2, 2, 2))
    G.update(convt_x=xt, convt_w=wt, convt_y=conv_2d(xt, wt, [0, 0], [1, 1], [1, 1]))
    # group_norm / GroupNorm (ff/group_norm.py:3, :13)
    xg = rnd("gn.x", (2, 64, 5, 3), 2.0) + 0.5
    gm = GroupNorm(32, 64); gm.weight = 1 + rnd("gn.w", (64,), 0.1); gm.bias = rnd("gn.b", (64,), 0.1)
    G.update(gn_x=xg, gn_w=gm.weight, gn_b=gm.bias, gn_y_plain=group_norm(xg, 32, 1e-5), gn_y_affine=gm(xg))
    # tests/group_norm.py family: (N, C, 2, 2), 2 groups
    xg2 = rnd("gn2.x", (8, 768, 2, 2))
    G.update(gn2_x=xg2, gn2_y=group_norm(xg2, 2, 1e-5))
    # LayerNorm (ff/layer_norm.py:34) -- B=1 only is meaningful in the reference (D4)
    xl = rnd("ln.x", (1, 7, 64), 1.5) + 0.25
    lm = LayerNorm(64); lm.weight = 1 + rnd("ln.w", (64,), 0.1); lm.bias = rnd("ln.b", (64,), 0.1)
    G.update(ln_x=xl, ln_w=lm.weight, ln_b=lm.bias, ln_y=lm(xl))
    # Linear (ff/linear.py:112)
    xi = rnd("lin.x", (2, 5, 64)); li = Linear(64, 48); li.weight = rnd("lin.w", (48, 64), 0.125); li.bias = rnd("lin.b", (48,), 0.1)
    G.update(lin_x=xi, lin_w=li.weight, lin_b=li.bias, lin_y=li(xi))
    li.bias = None
    G.update(lin_y_nobias=li(xi))
    # GEGLU / FeedForward (ff/nn.py:5, :14)
    ge = GEGLU(64, 128); ge.proj.weight = rnd("geglu.w", (256, 64), 0.125); ge.proj.bias = rnd("geglu.b", (256,), 0.1)
    G.update(geglu_w=ge.proj.weight, geglu_b=ge.proj.bias, geglu_y=ge(xi))
    # sdpa (attention/sdpa.py:53)
    q = rnd("sdpa.q", (2, 2, 16, 8)); k = rnd("sdpa.k", (2, 2, 16, 8)); v = rnd("sdpa.v", (2, 2, 16, 8))
    kc = rnd("sdpa.kc", (2, 2, 5, 8)); vc_ = rnd("sdpa.vc", (2, 2, 5, 8))
    G.update(sdpa_q=q, sdpa_k=k, sdpa_v=v, sdpa_kc=kc, sdpa_vc=vc_,
             sdpa_y_self=scaled_dot_product_attention(q, k, v), sdpa_y_cross=scaled_dot_product_attention(q, kc, vc_))
    mask = np.tril(np.ones((16, 16), dtype=bool))
    G.update(sdpa_y_causal=scaled_dot_product_attention(q, k, v, attn_mask=mask))
    # activations (storage/tensor.py:64-86)
    xa = rnd("act.x", (3, 50), 3.0)
    G.update(act_x=xa, act_sigmoid=Tensor.sigmoid(xa), act_silu=Tensor.silu(xa), act_gelu=Tensor.gelu(xa),
             act_quick_gelu=Tensor.quick_gelu(xa), act_swish=Tensor.swish(xa))
    # timestep embedding, up/down-sample (vision/unet.py:78-97)
    G.update(temb_981=timestep_embedding(np.array([981]), 320), temb_1=timestep_embedding(np.array([1]), 320))
    xu = rnd("ud.x", (2, 8, 4, 6))
    up = Upsample(8); up.conv.weight = rnd("up.w", (8, 8, 3, 3), 0.2); up.conv.bias = rnd("up.b", (8,), 0.1)
    dn = Downsample(8); dn.op.weight = rnd("dn.w", (8, 8, 3, 3), 0.2); dn.op.bias = rnd("dn.b", (8,), 0.1)
    G.update(ud_x=xu, up_w=up.conv.weight, up_b=up.conv.bias, up_y=up(xu), dn_w=dn.op.weight, dn_b=dn.op.bias, dn_y=dn(xu))
    # schedule + DDIM (variants/sd.py:14-25, :61-65)
    G.update(alphas_cumprod=get_alphas_cumprod())
    sd = StableDiffusion.__new__(StableDiffusion)
    xx = rnd("ddim.x", (1, 4, 8, 8)); ee = rnd("ddim.e", (1, 4, 8, 8))
    a_t = np.array([0.0413], dtype=np.float32); a_p = np.array([0.0502], dtype=np.float32)
    xp, px0 = sd.get_x_prev_and_pred_x0(xx, ee, a_t, a_p)
    G.update(ddim_x=xx, ddim_e=ee, ddim_a_t=a_t, ddim_a_prev=a_p, ddim_x_prev=xp, ddim_pred_x0=px0)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **{k: np.asarray(v, dtype=np.float32) for k, v in G.items()})
    print("ops.npz", len(G), "arrays")


def gen_blocks():
    from tinyfusers.vision.resnet import ResBlock
    from tinyfusers.attention.attention import CrossAttention, BasicTransformerBlock, SpatialTransformer
    from oracle.unet import UNetConfig, unet_param_shapes
    G = {}
    # Use the parameter-shape enumerator on a one-level toy config only to get names/shapes for a
    # ResBlock(64->128, emb 256) and a SpatialTransformer(64, ctx 32, 2 heads, d_head 32).
    shapes = {
        "res.in_layers.0.weight": (64,), "res.in_layers.0.bias": (64,),
        "res.in_layers.2.weight": (128, 64, 3, 3), "res.in_layers.2.bias": (128,),
        "res.emb_layers.1.weight": (128, 256), "res.emb_layers.1.bias": (128,),
        "res.out_layers.0.weight": (128,), "res.out_layers.0.bias": (128,),
        "res.out_layers.3.weight": (128, 128, 3, 3), "res.out_layers.3.bias": (128,),
        "res.skip_connection.weight": (128, 64, 1, 1), "res.skip_connection.bias": (128,),
    }
    c, cd = 64, 32
    st = {"st.norm.weight": (c,), "st.norm.bias": (c,), "st.proj_in.weight": (c, c, 1, 1), "st.proj_in.bias": (c,),
          "st.proj_out.weight": (c, c, 1, 1), "st.proj_out.bias": (c,)}
    t = "st.transformer_blocks.0"
    for a, d in (("attn1", c), ("attn2", cd)):
        st.update({f"{t}.{a}.to_q.weight": (c, c), f"{t}.{a}.to_k.weight": (c, d), f"{t}.{a}.to_v.weight": (c, d),
                   f"{t}.{a}.to_out.0.weight": (c, c), f"{t}.{a}.to_out.0.bias": (c,)})
    st.update({f"{t}.ff.net.0.proj.weight": (8 * c, c), f"{t}.ff.net.0.proj.bias": (8 * c,),
               f"{t}.ff.net.2.weight": (c, 4 * c), f"{t}.ff.net.2.bias": (c,)})
    for n in ("norm1", "norm2", "norm3"):
        st.update({f"{t}.{n}.weight": (c,), f"{t}.{n}.bias": (c,)})
    shapes.update(st)
    W = synth(shapes, seed=3)
    rb = ResBlock(64, 256, 128); install(rb, W, "res")
    x = rnd("res.x", (2, 64, 6, 5)); emb = rnd("res.emb", (1, 256))
    G.update(res_x=x, res_emb=emb, res_y=rb(x, emb))
    sp = SpatialTransformer(64, 32, 2, 32); install(sp, W, "st")
    xs = rnd("st.x", (2, 64, 4, 4)); ctx = rnd("st.ctx", (2, 5, 32))
    G.update(st_x=xs, st_ctx=ctx, st_y=sp(xs, ctx))
    blk = sp.transformer_blocks[0]
    xt = rnd("blk.x", (2, 16, 64))
    G.update(blk_x=xt, blk_y=blk(xt, ctx), attn1_y=blk.attn1(xt), attn2_y=blk.attn2(xt, ctx), ff_y=blk.ff(xt))
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **{k: np.asarray(v, dtype=np.float32) for k, v in G.items()})
    print("blocks.npz", len(G), "arrays (weights: synth seed 3, regenerated by name)")


def gen_unet():
    from collections import namedtuple
    from tinyfusers.vision.unet import UNetModel
    from tinyfusers.variants.sd import StableDiffusion, get_alphas_cumprod
    from oracle.unet import unet_param_shapes, SD15
    t0 = time.time()
    W = synth(unet_param_shapes(SD15), seed=0)
    unet = UNetModel(); install(unet, W, "")
    print("weights installed %.1fs" % (time.time() - t0))
    del W
    taps = {}
    calls = []

    class Rec:
        def __call__(self, x, t, c):
            y = unet(x, t, c)
            calls.append((np.array(x), np.array(t), np.array(c), np.array(y)))
            return y
    sd = StableDiffusion.__new__(StableDiffusion)
    sd.alphas_cumprod = get_alphas_cumprod()
    sd.model = namedtuple("DiffusionModel", ["diffusion_model"])(diffusion_model=Rec())
    latent = rnd("sd.latent", (1, 4, 64, 64), seed=1234)
    ctx = rnd("sd.context", (1, 77, 768), seed=1234); unc = rnd("sd.uncond", (1, 77, 768), seed=1234)
    timesteps = list(range(1, 1000, 1000 // 50))
    alphas = sd.alphas_cumprod[timesteps]
    alphas_prev = np.concatenate((np.array([1.0]), alphas[:-1])).astype(np.float32)
    G = dict(timesteps=np.array(timesteps, dtype=np.float32), alphas=alphas, alphas_prev=alphas_prev)
    x = latent
    for n, index in enumerate((49, 48)):
        t1 = time.time()
        tid = np.array([index])
        x = sd(unc, ctx, x, np.array([timesteps[index]]), alphas[tid], alphas_prev[tid], np.array([7.5]))
        G[f"x_after_step{n}"] = np.array(x)
        G[f"unet_out_step{n}"] = calls[-1][3]
        print("step", n, "%.1fs" % (time.time() - t1), float(np.abs(x).mean()))
    np.savez_compressed(os.path.join(HERE, "unet_sd15.npz"), **{k: np.asarray(v, dtype=np.float32) for k, v in G.items()})
    print("unet_sd15.npz written (inputs: synth seed 1234 by name; weights: synth seed 0)")


def gen_unet50():
    """The whole 50-step schedule of example/sd1.py:54-73 through the reference's StableDiffusion.__call__ (variants/sd.py:56-59) and
    UNetModel, same seeds / weights / contexts as gen_unet: the latent after every tenth step and the final one."""
    from collections import namedtuple
    from tinyfusers.vision.unet import UNetModel
    from tinyfusers.variants.sd import StableDiffusion, get_alphas_cumprod
    from oracle.unet import unet_param_shapes, SD15
    W = synth(unet_param_shapes(SD15), seed=0)
    unet = UNetModel(); install(unet, W, "")
    del W
    sd = StableDiffusion.__new__(StableDiffusion)
    sd.alphas_cumprod = get_alphas_cumprod()
    sd.model = namedtuple("DiffusionModel", ["diffusion_model"])(diffusion_model=unet)
    latent = rnd("sd.latent", (1, 4, 64, 64), seed=1234)
    ctx = rnd("sd.context", (1, 77, 768), seed=1234); unc = rnd("sd.uncond", (1, 77, 768), seed=1234)
    timesteps = list(range(1, 1000, 1000 // 50))
    alphas = sd.alphas_cumprod[timesteps]
    alphas_prev = np.concatenate((np.array([1.0]), alphas[:-1])).astype(np.float32)
    G = dict(timesteps=np.array(timesteps, dtype=np.float32), alphas=alphas, alphas_prev=alphas_prev)
    x = latent
    out = os.path.join(HERE, "unet50_sd15.npz")
    for n, index in enumerate(range(49, -1, -1)):          # example/sd1.py:68: indices high -> low
        t1 = time.time()
        tid = np.array([index])
        x = sd(unc, ctx, x, np.array([timesteps[index]]), alphas[tid], alphas_prev[tid], np.array([7.5]))
        x = np.asarray(x, dtype=np.float32)
        if (n + 1) % 10 == 0 or n < 2:
            G[f"x_after_step{n}"] = np.array(x)
            np.savez_compressed(out, **{k: np.asarray(v, dtype=np.float32) for k, v in G.items()})
        print("step", n, "index", index, "%.1fs" % (time.time() - t1), "mean|x| %.4f" % float(np.abs(x).mean()), flush=True)
    print("unet50_sd15.npz written (inputs: synth seed 1234 by name; weights: synth seed 0)")


def gen_vae():
    """StableDiffusion.decode (variants/sd.py:48-54) through the reference's Decoder / AttnBlock / ResnetBlock."""
    from types import SimpleNamespace
    from tinyfusers.vae.decoder import Decoder
    from tinyfusers.vision.conv2d import Conv2d
    from tinyfusers.variants.sd import StableDiffusion
    from oracle.vae import vae_decoder_param_shapes
    W = synth(vae_decoder_param_shapes(), seed=0)
    fsm = SimpleNamespace(decoder=Decoder(), post_quant_conv=Conv2d(4, 4, kernel_size=[1, 1]))
    install(fsm, W, "first_stage_model")
    sd = StableDiffusion.__new__(StableDiffusion)
    sd.first_stage_model = fsm
    latent = rnd("vae.latent", (1, 4, 64, 64), 0.18215 * 0.8, seed=1234)
    t0 = time.time()
    import io, contextlib
    taps = {}
    dec = fsm.decoder
    orig = dec.__class__.__call__
    def tapped(self, x):
        y = orig(self, x); taps["pre"] = np.array(y); return y
    dec.__class__.__call__ = tapped
    with contextlib.redirect_stdout(io.StringIO()):
        img = sd.decode(latent)
    dec.__class__.__call__ = orig
    print("decode %.1fs" % (time.time() - t0), img.shape, img.dtype, float(img.mean()))
    np.savez_compressed(os.path.join(HERE, "vae_sd15.npz"), pre_sub=taps["pre"][:, :, ::4, ::4].astype(np.float32),
                        img_sub=np.asarray(img)[::4, ::4].astype(np.uint8), pre_mean=np.float32(taps["pre"].mean()), pre_std=np.float32(taps["pre"].std()))
    print("vae_sd15.npz written")


def gen_clip():
    """CLIPEncoder (12 x CLIPEncoderLayer) + final LayerNorm of the reference (vae/encoder.py:39-81) on synthetic hidden
    states, under the reference's own causal mask.  The embedding front end is NOT taken from the reference: its
    Embedding is defective; see oracle/clip.py."""
    from tinyfusers.vae.encoder import CLIPTextTransformer
    from oracle.clip import clip_param_shapes
    pre = "cond_stage_model.transformer.text_model."
    shapes = {k: v for k, v in clip_param_shapes(pre).items() if ".embeddings." not in k}
    W = synth(shapes, seed=0)
    t = CLIPTextTransformer()
    holder = types.SimpleNamespace(encoder=t.encoder, final_layer_norm=t.final_layer_norm)
    install(holder, W, pre[:-1])
    h = rnd("clip.hidden", (1, 77, 768), 0.05, seed=1234)
    mask = np.triu(np.full((1, 1, 77, 77), float("-inf")), k=1).astype(np.float32)       # vae/encoder.py:79
    t0 = time.time()
    l0 = t.encoder.layers[0](h, mask)
    y = t.final_layer_norm(t.encoder(h, mask))
    print("clip encoder %.1fs" % (time.time() - t0), y.shape, float(np.abs(y).mean()))
    np.savez_compressed(os.path.join(HERE, "clip_text.npz"), layer0=np.asarray(l0, dtype=np.float32)[:, ::4], out=np.asarray(y, dtype=np.float32))
    print("clip_text.npz written (hidden: synth 'clip.hidden' seed 1234 std 0.05; weights: synth seed 0)")


TOKENIZER_TEXTS = [
    "a horse sized cat eating a bagel", "", "  Multiple   spaces\tand\nnewlines  ", "It's the cat's pyjamas, isn't it? They've won!",
    "UPPER lower MiXeD", "naive cafe \u00e9l\u00e8ve \u00fcber stra\u00dfe", "emoji \U0001F600 and symbols #$%&*()[]{}", "<|startoftext|>quoted<|endoftext|> specials",
    "aaaaaaaaaaaaaaaaaaaa bbbbbbbbbb abababababab", "the quick brown fox jumps over the lazy dog " * 12,
    "supercalifragilisticexpialidocious antidisestablishmentarianism", "1234567890 3.14159 2024-10-03", "x",
]


def toy_merges(n_merges=400):
    """A small byte-level BPE merge table trained here on a fixed paragraph (most frequent pair first, ties broken by
    the pair itself): DATA for the tokenizer fixture, in the format of bpe_simple_vocab_16e6.txt.gz."""
    from collections import Counter
    from tinyfusers_amd.tokenizer.clip import byte_symbols
    table, _ = byte_symbols()
    corpus = (" ".join(TOKENIZER_TEXTS) + " the cat sat on the mat and ate the bagel while a horse watched the lazy dog jump over "
              "the quick brown fox again and again because eating is something that cats horses and dogs all like doing").lower().split()
    words = Counter(tuple(table[b] for b in w.encode("utf-8"))[:-1] + (table[w.encode("utf-8")[-1]] + "</w>",) for w in corpus)
    merges = []
    for _ in range(n_merges):
        pairs = Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        best = min(pairs, key=lambda p: (-pairs[p], p))
        merges.append(best)
        nw = Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            nw[tuple(out)] += c
        words = nw
    return merges


def gen_tokenizer():
    """ClipTokenizer of the reference (tokenizer/clip.py) on a toy merge table: its module calls tinygrad's fetch() for the
    real table while being imported, so tinygrad.helpers is a stand-in whose fetch returns the toy file."""
    import gzip, json
    path = os.path.join(HERE, "clip_bpe_toy.txt.gz")
    merges = toy_merges()
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(("#version: toy table for tests, %d merges\n" % len(merges) + "\n".join(a + " " + b for a, b in merges)).encode("utf-8"))
    tg = types.ModuleType("tinygrad"); th = types.ModuleType("tinygrad.helpers")
    th.fetch = lambda url, name=None: path
    tg.helpers = th
    sys.modules["tinygrad"], sys.modules["tinygrad.helpers"] = tg, th
    from tinyfusers.tokenizer.clip import ClipTokenizer
    tok = ClipTokenizer()
    out = {t: tok.encode(t) for t in TOKENIZER_TEXTS}
    with open(os.path.join(HERE, "clip_tokens.json"), "w") as f:
        json.dump({"texts": TOKENIZER_TEXTS, "ids": [out[t] for t in TOKENIZER_TEXTS], "vocab_size": len(tok.encoder)}, f)
    print("clip_bpe_toy.txt.gz (%d merges), clip_tokens.json (%d texts, vocab %d)" % (len(merges), len(out), len(tok.encoder)))


if __name__ == "__main__":
    what = sys.argv[1:] or ["ops", "blocks"]
    import_reference()
    if "unet50" in what: gen_unet50()
    if "ops" in what: gen_ops()
    if "blocks" in what: gen_blocks()
    if "unet" in what: gen_unet()
    if "vae" in what: gen_vae()
    if "clip" in what: gen_clip()
    if "tokenizer" in what: gen_tokenizer()
