"""Process-wide switches of the drop-in path.  The boolean fusions can be turned off from the environment for A/B runs
(TF_FUSE_LAYER_NORM=0, TF_FOLD_SKIP=0, TF_FOLD_PROJ_OUT=0, TF_PARALLEL_BRANCHES=1); results stay within the stated tolerance."""
import os


def _flag(name, default):
    v = os.environ.get(name)
    return default if v is None else v not in ("0", "", "false", "False")


# Head-merge layout after SDPA in CrossAttention:
#   "reference_exact": (b,h,t,d) reshaped straight to (b,-1,h*d) with no transpose back -- what
#                      attention/attention.py:38-39 actually computes; the default, so results equal the reference's.
#   "intended"       : the LDM / tinygrad head merge (transpose(0,2,1,3) first), for real SD weights.
# It is purely an output-stride choice of the SDPA kernel; FLOPs and bytes are identical.
head_merge = "reference_exact"

# LayerNorm -> Linear pairs of the transformer blocks run as ONE GEMM on the raw activations (tf_linear_ln_f16:
# row statistics from the streamed fragments, gamma/beta folded into the weights once).  False = separate
# tf_layer_norm_f16 launches (the unfused reference structure).
fuse_layer_norm = _flag("TF_FUSE_LAYER_NORM", True)

# Independent sub-chains of the step (a ResBlock's 1x1 skip projection next to its GroupNorm -> conv -> GroupNorm main
# path; the context K|V projection next to the time-embedding MLP) run as parallel branches of the step graph on a side
# stream.  Results are bit-identical either way; False serialises everything on one stream.
# MEASURED SLOWER on MI355X (ROCm 7.2): 15 fork/join pairs per step cost 5.30 ms/step against 5.12 ms serial -- every
# cross-stream graph edge is a barrier packet + signal round trip of several microseconds, more than the ~10 us kernels it
# hides.  Kept as a switch for larger batches, off by default.
parallel_branches = _flag("TF_PARALLEL_BRANCHES", False)

# ResBlock: fold the 1x1 skip_connection into the last 3x3 conv as extra K columns (one launch instead of two plus a
# residual read); False runs the reference's two convs.
fold_skip_projection = _flag("TF_FOLD_SKIP", True)

# SpatialTransformer: fold the last FeedForward Linear (4C -> C) and proj_out (1x1 conv) into one GEMM with K = 5C over the
# pair (GEGLU output, FF input); weights folded once on the host in fp32.  False runs the reference's two GEMMs.
fold_proj_out = _flag("TF_FOLD_PROJ_OUT", True)

# GroupNorm over an equal-split channel concat (output path of the UNet) from the 32-group partials of the two producers
# (pairs of groups merge) instead of a statistics pass over the concat.
concat_stats = _flag("TF_CONCAT_STATS", True)
# ... for tensors up to this many elements: the producers then carry a statistics epilogue (and cannot run on the persistent short-K kernel, which has none),
# which pays at batch 1 (a statistics launch saved per concat) and costs 0.17 ms per step on config 5's shape, where the statistics pass over a concat is
# a 20-us HBM-bound launch (profiles/r04_ab.txt)
concat_stats_max_elems = int(os.environ.get("TF_CONCAT_STATS_MAX_ELEMS", str(4 << 20)))

# GroupNorm (+ SiLU) in front of a convolution applied inside the conv launch (tf_conv2d_gn_f16): the loader waves normalise the
# activation pieces in LDS.  False = a GroupNorm-apply launch in front of every such conv (the unfused reference structure).
fuse_group_norm = _flag("TF_FUSE_GROUP_NORM", True)
# ... up to this many input elements (rows x channels).  The fused form normalises a block's activation rows once per n-tile, so its extra work grows
# with rows x channels x n-tiles while the launch it saves is a constant 8.5 us: a win at batch 1 (BASELINE config 2: 0.65-2.6 M elements per proj_in),
# a loss on config 5's shape (5.9-23.6 M: 73 against 26 + 6 us at 4608 x 1280, the step 0.37 ms slower with it: profiles/r04_ab.txt).
fuse_group_norm_max_elems = int(os.environ.get("TF_FUSE_GROUP_NORM_MAX_ELEMS", str(4 << 20)))
# ... also for the 3x3 convolutions of the ResBlocks (patch kernel).  Correct and covered by the GPU tests, but MEASURED SLOWER on MI355X
# (profiles/r02_gn_in_conv.txt): every block normalises its whole input patch for all its n-tiles (4-8x the elements k_gn_apply
# touches) on loader waves whose LDS-DMA issue is the kernel's critical path: +12...20 us per conv against the 8.5 us launch saved.
# On by default only for the 1x1 proj_in convolutions (+3 us against 8.5 us saved).
fuse_group_norm_3x3 = _flag("TF_FUSE_GROUP_NORM_3X3", False)

# conv -> GroupNorm (-> SiLU) behind a split-K shape: the reduce kernel that already owns the statistics also writes the normalised
# tensor (k_splitk_reduce_gn_apply, tf_conv2d_fused_norm_f16).  False = reduce launch + GroupNorm-apply launch.
# MEASURED (profiles/r02_ab_fusions.txt, same box): 13.8 us per launch against 8.0 + 10.8 us for the two launches it replaces, 33 per step.
# With its own tuner keys (and a penalty for unsplit shapes) the GEMMs in front moved to slower configurations and the step lost
# 0.03-0.05 ms; with the plain keys (same tile as without it) the step gains 0.01 ms and 33 launches: on.
fuse_reduce_norm = _flag("TF_FUSE_REDUCE_NORM", True)

# Sampler: run the unconditional and the conditional half of the CFG pair as two independent UNet chains (batch B each) on two
# streams / graph branches instead of one chain at batch 2B (StableDiffusion._eager_step).  Same arithmetic per sample.
cfg_parallel = _flag("TF_CFG_PARALLEL", False)

# Compiled sampler (StableDiffusion.compile): work that does not depend on the latent stays out of the captured step -- the cross-attention
# K|V projection of the context (a function of the context: computed in compile() / set_context()) and the time-embedding row of a timestep
# (unet.py:54-56 + the 22 ResBlock projections: a function of t, computed once per distinct timestep and handed to the replay by the launch
# that sets the step scalars).  Bit-identical results; 5 launches (~55 us) fewer per step.  False = the reference's per-step recomputation.
hoist_step_invariants = _flag("TF_HOIST_STEP_INVARIANTS", True)

# "bf16" (round 4): every 16-bit tensor of the step holds bfloat16 -- install the weights AFTER set_dtype("bf16") (update_state then makes bfloat16 leaves);
# conv / linear / GEGLU on the bf16 MFMA, norms and the sampler's small kernels in bfloat16, the attention core on the fp16 kernel behind a conversion;
# the plain per-op structure (no LayerNorm fold, no split-K, no statistics riding on the convs): a correct bfloat16 step, not a tuned one.
# Operand type of the conv / linear GEMMs: "fp16" (default; BASELINE configs 2-4) or "fp8" (config 5: OCP e4m3 weights with
# per-output-channel scales packed once, e4m3 activations with a per-tensor scale quantised by the loader side, fp32 accumulate,
# fp16 residual stream).  Set through set_dtype() before the first forward.
dtype = "fp16"


def set_dtype(name):
    global dtype
    if name not in ("fp16", "fp8", "bf16"):
        raise ValueError(f"config.set_dtype: unknown dtype {name!r}")
    if name == "bf16" and (parallel_branches or cfg_parallel):
        # those two experiments (both measured slower, both off by default) have fp16-only side paths: refuse the combination instead of
        # producing wrong latents
        raise ValueError("config.set_dtype('bf16'): not with TF_PARALLEL_BRANCHES / TF_CFG_PARALLEL (fp16-only experiments)")
    dtype = name


def is_bf16():
    return dtype == "bf16"
