"""StableDiffusion sampler object -- mirrors tinyfusers/variants/sd.py:7-65.

Same call surface: ``sd(unconditional_context, context, latent, timestep, alphas, alphas_prev, guidance)``
returns x_{t-1}.  Differences, all MI355X-first:
  * no host round trip / device syncs per step (variants/sd.py:34-41); CFG duplicate, CFG combine and the DDIM
    update are two tiny kernels; the latent state stays fp32 NCHW on the device;
  * batch generalised from the reference's hard-coded 1 (D8) to [uncond x B ; cond x B];
  * ``compile()`` captures the whole step (one launch per fused op) into one HIP graph replayed per step.
Beyond the UNet denoising path the next rows are built too: first_stage_model (VAE decode side, 8(f1))
and cond_stage_model.transformer.text_model (CLIP text encoder, 8(f2)), under the reference's attribute names so that
update_state walks the same LDM checkpoint keys.
"""
import ctypes
from collections import namedtuple

import numpy as np

from .. import config
from ..native import hip
from ..storage.tensor import Branch, DeviceArray, Stream, _sh, asarray, bfloat16, pool, use_stream
from ..vision.unet import SD15, StepParams, UNetModel


def get_alphas_cumprod(beta_start=0.00085, beta_end=0.0120, n_training_steps=1000):
    """variants/sd.py:61-65 (host fp32: a 1000-entry table, not device work)."""
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n_training_steps, dtype=np.float32) ** 2
    alphas = 1.0 - betas
    return np.cumprod(alphas, axis=0)


def _scalar(v):
    return float(np.asarray(v.numpy() if isinstance(v, DeviceArray) else v, dtype=np.float32).reshape(-1)[0])


class StableDiffusion:
    def __init__(self, cfg=SD15, init=False):
        self.alphas_cumprod = get_alphas_cumprod()
        self.model = namedtuple("DiffusionModel", ["diffusion_model"])(diffusion_model=UNetModel(cfg, init=init))
        from ..vae.vae import AutoencoderKL
        self.first_stage_model = AutoencoderKL(init=init, init_encoder=False) if cfg is SD15 else None   # the sampler uses the decode side only: the encoder stays an empty tree until update_state fills it
        self.cond_stage_model = None
        if cfg is SD15:                  # variants/sd.py:12: cond_stage_model.transformer.text_model
            from ..vae.encoder import CLIPTextTransformer
            self.cond_stage_model = namedtuple("CondStageModel", ["transformer"])(
                transformer=namedtuple("Transformer", ["text_model"])(text_model=CLIPTextTransformer(init=init)))
        self._params = None
        self._graph = None

    # -- reference surface -------------------------------------------------------------------------
    def get_model_output(self, unconditional_context, context, latent, timestep, unconditional_guidance_scale, params=None):
        """variants/sd.py:27-46.  Returns the raw UNet output for [uncond x B ; cond x B] (2B,4,H,W) -- the CFG
        combine is fused with the DDIM update in __call__ (use cfg_combine() to get e_t on its own)."""
        b, c, h, w = latent.shape
        x2 = self._cfg_duplicate(latent)
        ctx = self._stack_context(unconditional_context, context)
        sp = params if params is not None else self._step_params().set(_scalar(timestep), 1.0, 1.0, _scalar(unconditional_guidance_scale))
        return self.model.diffusion_model(x2, sp, ctx)

    def get_x_prev_and_pred_x0(self, x, e_t, a_t, a_prev):
        """variants/sd.py:14-25 on host arrays (kept for API parity / tests; the device path is tf_cfg_ddim_step_f32)."""
        x, e_t = np.asarray(x, dtype=np.float32), np.asarray(e_t, dtype=np.float32)
        a_t, a_prev = np.float32(_scalar(a_t)), np.float32(_scalar(a_prev))
        sqrt_one_minus_at = np.sqrt(1 - a_t)
        pred_x0 = (x - sqrt_one_minus_at * e_t) / np.sqrt(a_t)
        dir_xt = np.sqrt(1.0 - a_prev) * e_t
        return np.sqrt(a_prev) * pred_x0 + dir_xt, pred_x0

    def __call__(self, unconditional_context, context, latent, timestep, alphas, alphas_prev, guidance):
        """variants/sd.py:56-59: one denoising step; latent (B,4,H,W) fp32 device array -> new latent."""
        sp = self._step_params().set(_scalar(timestep), _scalar(alphas), _scalar(alphas_prev), _scalar(guidance))
        out = self.get_model_output(unconditional_context, context, latent, timestep, guidance, params=sp)
        b, c, h, w = latent.shape
        x_prev = DeviceArray.empty(latent.shape, np.float32, "row")
        hip.tf_memcpy_async(x_prev.ptr, latent.ptr, latent.nbytes, 3, _sh())
        (hip.tf_cfg_ddim_step_bf16 if config.is_bf16() else hip.tf_cfg_ddim_step_f32)(x_prev.ptr, out.ptr, sp.dev.ptr, b, c, h, w, _sh())
        return x_prev

    def decode(self, x):
        """variants/sd.py:48-54: post_quant_conv(x / 0.18215) -> Decoder -> (x+1)/2 -> clip -> uint8 (H, W, 3).
        x: fp32 NCHW latent (1,4,h,w) on the device.  Returns a host uint8 array."""
        b, c, h, w = x.shape
        assert b == 1, "decode: batch 1 (the reference reshapes to (3,512,512), variants/sd.py:52)"
        z16 = DeviceArray.empty((b, c, h, w), np.float16, "row")
        hip.tf_scale_cast_f32_to_f16(z16.ptr, x.ptr, 1.0 / 0.18215, x.size, _sh())
        z = DeviceArray.empty((b, c, h, w), np.float16, "nhwc")
        hip.tf_nchw_to_nhwc_f16(z.ptr, z16.ptr, b, c, h, w, _sh())
        y = self.first_stage_model.decoder(self.first_stage_model.post_quant_conv(z))      # (1,3,8h,8w) NHWC
        n = y.size
        out = DeviceArray.empty((n,), np.uint8, "row")
        hip.tf_image_to_u8(out.ptr, y.ptr, n, _sh())
        hip.tf_stream_sync(_sh())
        host = np.empty((y.shape[2], y.shape[3], y.shape[1]), dtype=np.uint8)
        hip.tf_memcpy(host.ctypes.data, out.ptr, n, 2)
        return host

    # -- helpers ---------------------------------------------------------------------------------
    def _step_params(self):
        if self._params is None:
            self._params = StepParams()
        return self._params

    @staticmethod
    def _cfg_duplicate(latent):
        """variants/sd.py:31: the latent for both halves of the CFG pair, in the step's 16-bit type (NHWC)."""
        b, c, h, w = latent.shape
        x2 = DeviceArray.empty((2 * b, c, h, w), bfloat16 if config.is_bf16() else np.float16, "nhwc")
        (hip.tf_cfg_duplicate_bf16 if config.is_bf16() else hip.tf_cfg_duplicate_f16)(x2.ptr, latent.ptr, b, c, h, w, _sh())
        return x2

    @staticmethod
    def _stack_context(unconditional_context, context):
        b, t, d = context.shape
        ctx = DeviceArray.empty((2 * b, t, d), context.dtype, "row")
        hip.tf_memcpy_async(ctx.ptr, unconditional_context.ptr, context.nbytes, 3, _sh())
        hip.tf_memcpy_async(ctx.ptr + context.nbytes, context.ptr, context.nbytes, 3, _sh())
        if config.is_bf16():
            from ..ff.linear import to_bf16
            ctx = to_bf16(ctx)                                 # (the bfloat16 step takes fp16 or bfloat16 contexts)
        return ctx

    @staticmethod
    def latent_from_numpy(x):
        """(B,4,H,W) host array -> fp32 NCHW device latent (the sampler state)."""
        return DeviceArray.from_numpy(np.ascontiguousarray(x, dtype=np.float32), np.float32, "row")

    def set_latent(self, x):
        """Start a new image: host noise (B,4,H,W) -> the latent buffer the compiled step updates in place.  Ordered after
        every step already queued (the sampler stream is drained first)."""
        self.synchronize()
        self._latent.copy_from_numpy(x)
        return self._latent

    # -- whole-step HIP graph ------------------------------------------------------------------------
    def compile(self, unconditional_context, context, latent, stream=None, warmup=2, timesteps=None):
        """Capture one denoising step for these (static) buffers into a HIP graph.  Afterwards
        ``step(timestep, a_t, a_prev, guidance)`` updates ``latent`` in place with one graph launch.

        Step-invariant work stays out of the captured step (config.hoist_step_invariants): the cross-attention K|V projection of the
        context runs here and in ``set_context`` -- it depends on the context alone -- and the time-embedding row of a timestep (the MLP
        of unet.py:54-56 + the 22 ResBlock projections of resnet.py:28) is computed once per distinct timestep, kept in a table, and
        handed to the replay by the launch that sets the step scalars.  ``timesteps``: the schedule, to fill the table up front.

        The captured step reads PRIVATE buffers (the stacked context and, hoisted, its K|V projection): ``set_context`` is the only supported
        way to change the prompts of a compiled sampler -- writing into the arrays handed to ``compile`` (or into ``_ctx2``) changes nothing the
        cross-attention reads.  The hoisted tables are keyed by ``weights_key()``: eager steps (``step(..., eager=True)``) follow weights replaced
        after ``compile``; the captured graph holds the addresses of the weights it was captured with, so a new weight set needs ``compile`` again."""
        if config.is_bf16() and (config.parallel_branches or config.cfg_parallel):
            raise RuntimeError("StableDiffusion.compile: the bfloat16 step has no parallel-branch / two-chain CFG form (TF_PARALLEL_BRANCHES / TF_CFG_PARALLEL are fp16-only experiments)")
        self._stream = stream or Stream()
        self._latent, self._unc, self._ctx = latent, unconditional_context, context
        sp = self._step_params()
        unet = self.model.diffusion_model
        with use_stream(self._stream):
            self._ctx2 = self._stack_context(unconditional_context, context)
            self._kv_all, self._kv_key, self._emb_cur, self._emb_rows, self._emb_key = None, None, None, {}, None
            if config.hoist_step_invariants and not config.cfg_parallel:
                self._kv_all, self._kv_key = unet.context_kv(self._ctx2), unet.weights_key()
                _, row = unet.time_embedding_all(sp.set(981.0))
                self._emb_cur = DeviceArray.empty(row.shape, row.dtype, "row")       # what the captured step reads (fp16, or bfloat16 bits in the bf16 step)
                assert self._emb_cur.nbytes % 16 == 0
                hip.tf_memcpy_async(self._emb_cur.ptr, row.ptr, row.nbytes, 3, _sh())   # (the warm-up steps below run at t = 981)
                self._keep_row = row
                for t in (timesteps if timesteps is not None else ()):
                    self._emb_row(float(t))
            saved = DeviceArray.empty(latent.shape, np.float32, "row")
            hip.tf_memcpy_async(saved.ptr, latent.ptr, latent.nbytes, 3, _sh())
            for _ in range(warmup):                 # warms the pool and builds the lazily packed weights
                sp.set(981.0, 0.5, 0.6, 7.5)
                self._eager_step(sp)
            hip.tf_memcpy_async(latent.ptr, saved.ptr, latent.nbytes, 3, _sh())
            self._stream.synchronize()
            if getattr(self, "_graph_blocks", None):          # a previous graph of this model: its buffers go back to the pool
                hip.tf_graph_destroy(self._graph)
                pool().disown(self._graph_blocks)
                self._graph, self._graph_blocks = None, None
            pool().begin_capture()
            g, ok = ctypes.c_void_p(), False
            try:
                hip.tf_graph_begin_capture(self._stream.handle)
                self._eager_step(sp)
                hip.tf_graph_end_capture(self._stream.handle, ctypes.byref(g))
                ok = True
            finally:
                blocks = pool().end_capture()    # every block the captured step touches now belongs to the graph
                if not ok:
                    # an op raised inside the capture (pool frozen, untuned shape, ...): leave capture mode so that the stream
                    # stays usable, and give the blocks back -- there is no graph to own them
                    hip.tf_graph_abort_capture(self._stream.handle)
                    self._keep = None
                    pool().disown(blocks)
            self._graph, self._graph_blocks = g, blocks
        return self

    def _eager_step(self, sp):
        b, c, h, w = self._latent.shape
        x2 = self._cfg_duplicate(self._latent)
        unet = self.model.diffusion_model
        if config.cfg_parallel:
            # the unconditional and the conditional half of the CFG pair (variants/sd.py:31-32) as two independent UNet chains: one runs
            # as a side branch of the step (its own stream / graph branch), so the launch gaps and partly filled grids of one chain are
            # covered by the other; what both share (time embedding, context K|V) is computed once in front of the fork
            emb, emb_all, kv_all = unet.step_shared(sp, self._ctx2)
            half = lambda a, i: a.view((b,) + a.shape[1:], a.layout, i * b * (a.size // a.shape[0])) if a is not None else None
            br = Branch()
            with br:
                out_u = unet(half(x2, 0), sp, None, shared=(emb, emb_all, half(kv_all, 0)))
            out_c = unet(half(x2, 1), sp, None, shared=(emb, emb_all, half(kv_all, 1)))
            br.join()
            hip.tf_cfg_ddim_step2_f32(self._latent.ptr, out_u.ptr, out_c.ptr, sp.dev.ptr, b, c, h, w, _sh())
            self._keep = (x2, out_u, out_c, emb, emb_all, kv_all)
            return
        if getattr(self, "_emb_cur", None) is not None:
            out = unet(x2, sp, self._ctx2, shared=(None, self._emb_cur, self._kv_all))     # (step() has put this timestep's row into _emb_cur)
        else:
            out = unet(x2, sp, self._ctx2)
        (hip.tf_cfg_ddim_step_bf16 if config.is_bf16() else hip.tf_cfg_ddim_step_f32)(self._latent.ptr, out.ptr, sp.dev.ptr, b, c, h, w, _sh())
        self._keep = (x2, out)      # graph nodes reference these blocks: keep them out of the pool

    def step(self, timestep, a_t, a_prev, guidance, eager=False):
        """One denoising step on the sampler stream, asynchronous.  The stream switch is un-ordered on purpose: an event
        edge between two graph launches costs 0.2 ms per step (measured, tools/ab3.sh), and consecutive steps are ordered by
        the stream itself.  Work on other streams that touches the latent goes through set_latent() / synchronize()."""
        sp = self._params
        with use_stream(self._stream, ordered=False):
            if getattr(self, "_emb_cur", None) is not None:
                key = self.model.diffusion_model.weights_key()
                if getattr(self, "_kv_all", None) is not None and self._kv_key != key:
                    # a to_k / to_v (or any hoisted) weight was replaced since compile(): the K|V projection the graph reads is stale -- refresh it in place
                    kv = self.model.diffusion_model.context_kv(self._ctx2)
                    hip.tf_memcpy_async(self._kv_all.ptr, kv.ptr, kv.nbytes, 3, _sh())
                    self._kv_tmp, self._kv_key = kv, key
                row = self._emb_row(float(timestep), key)     # (computed on this stream the first time a timestep is seen)
                hip.tf_set_step_params_copy(sp.dev.ptr, float(timestep), float(a_t), float(a_prev), float(guidance), self._emb_cur.ptr, row.ptr, row.nbytes, _sh())
            else:
                sp.set(timestep, a_t, a_prev, guidance)
            if eager or self._graph is None:
                self._eager_step(sp)
            else:
                hip.tf_graph_launch(self._graph, self._stream.handle)

    def _emb_row(self, t, key=None):
        """The cached time-embedding row of timestep t (keyed by the weights it was computed from; at most 1024 rows are kept)."""
        unet = self.model.diffusion_model
        key = unet.weights_key() if key is None else key
        if self._emb_key != key:
            self._emb_rows, self._emb_key = {}, key
        row = self._emb_rows.get(t)
        if row is None:
            if len(self._emb_rows) >= 1024:
                self._emb_rows.clear()
            tmp = StepParams().set(t)
            row = self._emb_rows[t] = unet.time_embedding_all(tmp)[1]
            row._base = (row._base, tmp)
        return row

    def set_context(self, unconditional_context, context):
        """New prompts for the compiled step: refresh the stacked context in place (the captured graph reads these buffers) and the
        cross-attention K|V projection that was hoisted out of the step.  Ordered on the sampler stream."""
        with use_stream(self._stream):
            new = self._stack_context(unconditional_context, context)       # (in the step's 16-bit type)
            assert new.nbytes == self._ctx2.nbytes, "set_context: the contexts must have the shape the step was compiled for"
            hip.tf_memcpy_async(self._ctx2.ptr, new.ptr, new.nbytes, 3, _sh())
            self._ctx_tmp = new                                    # (referenced until the copy has run)
            if getattr(self, "_kv_all", None) is not None:
                kv = self.model.diffusion_model.context_kv(self._ctx2)
                hip.tf_memcpy_async(self._kv_all.ptr, kv.ptr, kv.nbytes, 3, _sh())
                self._kv_tmp, self._kv_key = kv, self.model.diffusion_model.weights_key()   # (referenced until the copy has run)
        self._unc, self._ctx = unconditional_context, context

    def synchronize(self):
        self._stream.synchronize()
