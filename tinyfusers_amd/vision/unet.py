"""UNetModel / Upsample / Downsample / timestep_embedding -- mirrors tinyfusers/vision/unet.py:9-97.

Same module tree and attribute names as the reference (they are the LDM checkpoint keys that update_state
walks), generated from a channel plan so that a down-scaled graph can be built for tests; the default plan
is exactly the SD-1.x one hard-coded in vision/unet.py:12-49.

Step-level batching that the reference does not do (results unchanged):
  * the 22 ResBlock ``emb_layers`` Linear(SiLU(emb)) run as ONE weight-streaming GEMV over a device-side
    concatenated weight at the top of the step;
  * the 32 cross-attention to_k / to_v projections of the context run as ONE GEMM (N = sum 2C) -- every
    layer then reads its K|V column slice through SDPA strides;
  * the channel concat of :72 is never materialised (GroupNorm / conv read both tensors), the nearest-2x
    upsample of :81-83 is folded into the following conv's gather.
"""
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from ..attention.attention import SpatialTransformer, _concat_rows
from ..ff.group_norm import GroupNorm
from ..ff.linear import Linear, gemv_f16, linear_any, linear_bf16, linear_f16, to_f16
from ..native import hip
from .. import config
from ..storage.tensor import Branch, DeviceArray, Tensor, _sh, asarray, bfloat16, is_bfloat16
from .conv2d import Conv2d
from .resnet import ResBlock


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    model_channels: int = 320
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    attention_levels: Tuple[int, ...] = (0, 1, 2)
    n_heads: int = 8
    context_dim: int = 768


SD15 = UNetConfig()
TINY = UNetConfig(model_channels=64, channel_mult=(1, 2, 2), attention_levels=(0, 1), n_heads=2, context_dim=64)


class Upsample:
    def __init__(self, channels, init=True):
        self.conv = Conv2d(channels, channels, kernel_size=[3, 3], padding=[1, 1], init=init)

    def __call__(self, x, out_gn=0, out_norm=None):
        return self.conv(x, upsample=True, gn=out_gn, out_norm=out_norm)       # nearest-2x (unet.py:81-83) folded into the conv gather


class Downsample:
    def __init__(self, channels, init=True):
        self.op = Conv2d(channels, channels, stride=[2, 2], kernel_size=[3, 3], padding=[1, 1], init=init)

    def __call__(self, x, out_gn=0, out_norm=None):
        return self.op(x, gn=out_gn, out_norm=out_norm)


class StepParams:
    """Device-resident per-step scalars [timestep, a_t, a_prev, guidance] (fp32).  ``set`` is a 1-thread kernel
    whose arguments carry the values, so it is stream-ordered with the step graph that reads them."""

    def __init__(self):
        self.dev = DeviceArray.empty((4,), np.float32, "row")
        self.set(0.0)

    def set(self, timestep, a_t=1.0, a_prev=1.0, guidance=1.0):
        hip.tf_set_step_params(self.dev.ptr, float(timestep), float(a_t), float(a_prev), float(guidance), _sh())
        return self


def _as_params(timesteps):
    if isinstance(timesteps, StepParams):
        return timesteps
    if isinstance(timesteps, DeviceArray):
        sp = StepParams.__new__(StepParams)
        sp.dev = timesteps
        return sp
    t = float(np.asarray(timesteps, dtype=np.float32).reshape(-1)[0])
    return StepParams().set(t)


def timestep_embedding(timesteps, dim, max_period=10000):
    """unet.py:92-97: (1, dim) = [cos(t f), sin(t f)].  timesteps: host scalar/array, or StepParams / a device
    fp32 array whose element 0 is the timestep."""
    sp = _as_params(timesteps)
    if config.is_bf16():
        out = DeviceArray.empty((1, dim), bfloat16, "row")
        hip.tf_timestep_embedding_bf16(out.ptr, sp.dev.ptr, dim, float(max_period), _sh())
        out._base = sp
        return out
    out = DeviceArray.empty((1, dim), np.float16, "row")
    hip.tf_timestep_embedding_f16(out.ptr, sp.dev.ptr, dim, float(max_period), _sh())
    out._base = sp
    return out


class UNetModel:
    def __init__(self, cfg: UNetConfig = SD15, init=False):
        self.cfg = cfg
        mc, emb = cfg.model_channels, cfg.model_channels * 4
        nh, cd = cfg.n_heads, cfg.context_dim
        R = lambda i, o: ResBlock(i, emb, o, init=init)
        S = lambda c: SpatialTransformer(c, cd, nh, c // nh, init=init)
        self.time_embed = [Linear(mc, emb, init=init), Tensor.silu, Linear(emb, emb, init=init)]
        self.input_blocks = [[Conv2d(cfg.in_channels, mc, kernel_size=[3, 3], padding=[1, 1], init=init)]]
        chans, ch, nlev = [mc], mc, len(cfg.channel_mult)
        for lev, mult in enumerate(cfg.channel_mult):
            for _ in range(cfg.num_res_blocks):
                blk = [R(ch, mc * mult)]
                ch = mc * mult
                if lev in cfg.attention_levels:
                    blk.append(S(ch))
                self.input_blocks.append(blk)
                chans.append(ch)
            if lev != nlev - 1:
                self.input_blocks.append([Downsample(ch, init=init)])
                chans.append(ch)
        self.middle_block = [R(ch, ch), S(ch), R(ch, ch)]
        self.output_blocks = []
        for lev in reversed(range(nlev)):
            mult = cfg.channel_mult[lev]
            for i in range(cfg.num_res_blocks + 1):
                blk = [R(ch + chans.pop(), mc * mult)]
                ch = mc * mult
                if lev in cfg.attention_levels:
                    blk.append(S(ch))
                if lev > 0 and i == cfg.num_res_blocks:
                    blk.append(Upsample(ch, init=init))
                self.output_blocks.append(blk)
        self.out = [GroupNorm(32, mc, init=init), Tensor.silu, Conv2d(mc, cfg.out_channels, kernel_size=[3, 3], padding=[1, 1], init=init)]
        self._batched = None

    # -- step-level batched projections (built once per weight set, on the device)
    def _all(self, kind):
        blocks = [bb for b in self.input_blocks for bb in b] + list(self.middle_block) + [bb for b in self.output_blocks for bb in b]
        return [bb for bb in blocks if isinstance(bb, kind)]

    def _prepare(self):
        res, sts = self._all(ResBlock), self._all(SpatialTransformer)
        key = tuple(r.emb_layers[1].weight.wkey for r in res) + tuple(r.emb_layers[1].bias.wkey for r in res) \
            + tuple(w.wkey for s in sts for w in (s.transformer_blocks[0].attn2.to_k.weight, s.transformer_blocks[0].attn2.to_v.weight))
        if self._batched is not None and self._batched["key"] == key:
            return self._batched
        emb_w = _concat_rows([r.emb_layers[1].weight for r in res])
        emb_b = _concat_rows([r.emb_layers[1].bias.view((r.emb_layers[1].bias.size, 1)) for r in res]).view((emb_w.shape[0],))
        emb_off, off = {}, 0
        for r in res:
            emb_off[id(r)] = (off, r.emb_layers[1].weight.shape[0]); off += r.emb_layers[1].weight.shape[0]
        kv_w, kv_off, off = None, {}, 0
        if sts:
            ws = []
            for s in sts:
                a = s.transformer_blocks[0].attn2
                ws += [a.to_k.weight, a.to_v.weight]
                kv_off[id(s)] = off; off += 2 * a.to_k.weight.shape[0]
            kv_w = _concat_rows(ws)
        self._batched = dict(key=key, emb_w=emb_w, emb_b=emb_b, emb_off=emb_off, kv_w=kv_w, kv_off=kv_off, kv_n=off)
        return self._batched

    def time_embedding_all(self, timesteps):
        """(emb, emb_all): the time-embedding chain of unet.py:54-56 and every ResBlock's Linear(SiLU(emb)) (resnet.py:28) as one row.  Both
        depend on the timestep alone -- not on the latent, the context or the batch -- so a sampler may compute the row of a timestep once
        and keep it (variants/sd.py here: StableDiffusion.compile(timesteps=...))."""
        bt = self._prepare()
        t_emb = timestep_embedding(timesteps, self.cfg.model_channels)
        emb = self.time_embed[2](self.time_embed[0](t_emb), silu_input=True)          # Linear -> SiLU -> Linear
        return emb, gemv_f16(emb, bt["emb_w"], bt["emb_b"], silu_input=True)          # every ResBlock's Linear(SiLU(emb))

    def context_kv(self, context):
        """Every cross-attention's K|V projection of the (stacked) context as ONE GEMM (attention/attention.py:35-36 for all 16 blocks):
        depends on the context alone, so a sampler computes it when the context changes, not once per step."""
        bt = self._prepare()
        return linear_any(context, bt["kv_w"]) if bt["kv_w"] is not None else None

    def weights_key(self):
        """Identity of everything time_embedding_all / context_kv depend on: a cached row is stale once any of these weights is replaced."""
        return self._prepare()["key"] + tuple(m.weight.wkey for m in (self.time_embed[0], self.time_embed[2])) + tuple(m.bias.wkey for m in (self.time_embed[0], self.time_embed[2]))

    def step_shared(self, timesteps, context):
        """What one denoising step computes ONCE for every sample: the time-embedding chain (the same timestep for the whole batch) and
        the cross-attention K|V projection of all contexts.  __call__ computes them itself unless handed the result (``shared``)."""
        emb, emb_all = self.time_embedding_all(timesteps)
        return emb, emb_all, self.context_kv(context)

    def __call__(self, x, timesteps=None, context=None, shared=None):
        cfg = self.cfg
        bt = self._prepare()
        br = None
        if shared is not None:
            emb, emb_all, kv_all = shared          # kv_all: this call's rows (b, tk, kv_n) of the step-level projection; emb may be None (emb_all is what the ResBlocks read)
        else:
            if config.parallel_branches and bt["kv_w"] is not None:
                br = Branch()                          # the context projection is independent of the time-embedding chain
                with br:
                    kv_all = linear_f16(context, bt["kv_w"])      # (either element type since round 5)
            emb, emb_all = self.time_embedding_all(timesteps)                            # Linear -> SiLU -> Linear, then every ResBlock's Linear(SiLU(emb))
            if br is None:
                kv_all = self.context_kv(context)                                         # every attn2's K|V of the context

        def run(x, bb, nxt, force_gn=0):
            # nxt = the module that reads this one's output as a single tensor (None across a concat): when it opens
            # with a GroupNorm, the statistics are produced by this module's last conv.  force_gn: the output (also) enters
            # an equal-split concat whose GroupNorm merges the two producers' partials (tf_group_norm_apply2_f16)
            gn = nxt.num_groups if isinstance(nxt, GroupNorm) else 32 if isinstance(nxt, (ResBlock, SpatialTransformer)) else force_gn
            # ... and where that conv runs split-K its reduce applies the norm too: (the GroupNorm module that opens nxt, silu)
            on = (nxt, True) if isinstance(nxt, GroupNorm) else (nxt.in_layers[0], True) if isinstance(nxt, ResBlock) \
                else (nxt.norm, False) if isinstance(nxt, SpatialTransformer) else None
            if isinstance(bb, ResBlock):
                off, n = bt["emb_off"][id(bb)]
                return bb(x, emb, emb_out=emb_all.view((emb_all.shape[0], n), "row", off), out_gn=gn, out_norm=on)
            if isinstance(bb, SpatialTransformer):
                c = bb.proj_in.weight.shape[0]
                return bb(x, context, kv=KVSlice(kv_all, bt["kv_off"][id(bb)], c, bt["kv_n"]), out_gn=gn, out_norm=on)
            if isinstance(bb, (Downsample, Upsample)):
                return bb(x, out_gn=gn, out_norm=on)
            if isinstance(bb, Conv2d):
                return bb(x, gn=gn, out_norm=on)               # conv_in: its output feeds the first ResBlock's GroupNorm and the last skip concat
            return bb(x)

        saved_inputs = []
        joined = br is None
        seq = [bb for b in self.input_blocks for bb in b] + list(self.middle_block)
        ends = set()                                # indices after which the tensor is also saved for a skip concat
        i = 0
        for b in self.input_blocks:
            i += len(b)
            ends.add(i - 1)
        for i, bb in enumerate(seq):
            nxt = seq[i + 1] if i + 1 < len(seq) else None     # the middle block's output enters a concat
            if not joined and isinstance(bb, SpatialTransformer):
                br.join(); joined = True            # first consumer of kv_all
            # tensors saved for (or entering) the output path's concat carry 32-group statistics when the concat is an equal split
            want = config.concat_stats and (i in ends or nxt is None) and x.size <= config.concat_stats_max_elems
            x = run(x, bb, nxt, force_gn=32 if want else 0)
            if i in ends:
                saved_inputs.append(x)
        if not joined:
            br.join()
        for bi, b in enumerate(self.output_blocks):
            x = (x, saved_inputs.pop())            # channel concat (unet.py:72), consumed un-materialised
            for j, bb in enumerate(b):
                last = bi == len(self.output_blocks) - 1 and j == len(b) - 1
                nxt = b[j + 1] if j + 1 < len(b) else (self.out[0] if last else None)
                cout = bb.conv.weight.shape[0] if isinstance(bb, Upsample) else (bb.out_layers[3].weight.shape[0] if isinstance(bb, ResBlock) else bb.proj_out.weight.shape[0])
                # the output enters the next concat: emit its statistics in sub-groups as wide as the 32 groups of the saved partner
                # (32 sub-groups for an equal split, 64 for the 2:1 splits), so that the concat's GroupNorm is apply-only
                fg = 0
                if config.concat_stats and nxt is None and saved_inputs and saved_inputs[-1].size <= config.concat_stats_max_elems:
                    c2 = saved_inputs[-1].shape[1]
                    sub = c2 // 32
                    if c2 % 32 == 0 and sub >= 4 and cout % sub == 0 and ((cout + c2) // 32) % sub == 0 and cout // sub <= 256:
                        fg = cout // sub
                x = run(x, bb, nxt, force_gn=fg)
        return self.out[2](self.out[0](x, silu=True))


class KVSlice:
    """Column slice [off, off+2C) of the step-level (b, tk, kv_n) K|V projection."""
    __slots__ = ("arr", "off", "c", "ld")

    def __init__(self, arr, off, c, ld):
        self.arr, self.off, self.c, self.ld = arr, off, c, ld
