"""Names/shapes of the toy ResBlock(64->128, emb 256) and SpatialTransformer(64, ctx 32, 2 heads x 32)
whose weights (synth seed 3) produced blocks.npz.  Data only; shared by make_golden.py and the tests."""
c, cd = 64, 32
BLOCK_SHAPES = {
    "res.in_layers.0.weight": (64,), "res.in_layers.0.bias": (64,),
    "res.in_layers.2.weight": (128, 64, 3, 3), "res.in_layers.2.bias": (128,),
    "res.emb_layers.1.weight": (128, 256), "res.emb_layers.1.bias": (128,),
    "res.out_layers.0.weight": (128,), "res.out_layers.0.bias": (128,),
    "res.out_layers.3.weight": (128, 128, 3, 3), "res.out_layers.3.bias": (128,),
    "res.skip_connection.weight": (128, 64, 1, 1), "res.skip_connection.bias": (128,),
    "st.norm.weight": (c,), "st.norm.bias": (c,), "st.proj_in.weight": (c, c, 1, 1), "st.proj_in.bias": (c,),
    "st.proj_out.weight": (c, c, 1, 1), "st.proj_out.bias": (c,),
}
_t = "st.transformer_blocks.0"
for _a, _d in (("attn1", c), ("attn2", cd)):
    BLOCK_SHAPES.update({f"{_t}.{_a}.to_q.weight": (c, c), f"{_t}.{_a}.to_k.weight": (c, _d), f"{_t}.{_a}.to_v.weight": (c, _d),
                         f"{_t}.{_a}.to_out.0.weight": (c, c), f"{_t}.{_a}.to_out.0.bias": (c,)})
BLOCK_SHAPES.update({f"{_t}.ff.net.0.proj.weight": (8 * c, c), f"{_t}.ff.net.0.proj.bias": (8 * c,),
                     f"{_t}.ff.net.2.weight": (c, 4 * c), f"{_t}.ff.net.2.bias": (c,)})
for _n in ("norm1", "norm2", "norm3"):
    BLOCK_SHAPES.update({f"{_t}.{_n}.weight": (c,), f"{_t}.{_n}.bias": (c,)})
