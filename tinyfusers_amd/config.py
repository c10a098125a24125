"""Process-wide switches of the drop-in path."""

# Head-merge layout after SDPA in CrossAttention:
#   "reference_exact": (b,h,t,d) reshaped straight to (b,-1,h*d) with no transpose back -- what
#                      attention/attention.py:38-39 actually computes; the default, so results equal the reference's.
#   "intended"       : the LDM / tinygrad head merge (transpose(0,2,1,3) first), for real SD weights.
# It is purely an output-stride choice of the SDPA kernel; FLOPs and bytes are identical.
head_merge = "reference_exact"

# LayerNorm -> Linear pairs of the transformer blocks run as ONE GEMM on the raw activations (tf_linear_ln_f16:
# row statistics from the streamed fragments, gamma/beta folded into the weights once).  False = separate
# tf_layer_norm_f16 launches (the unfused reference structure).
fuse_layer_norm = True
