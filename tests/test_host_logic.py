"""CPU tests of the host side: module tree / weight-name contract, synthetic generator, schedule, masks, sharding."""
import numpy as np
import pytest

import oracle


def test_update_state_names_match_reference_walk():
    """The product's module tree must yield exactly the LDM key set the reference's update_state walk produces
    (the oracle's enumerator was validated against the reference itself by make_golden.py: no skipped keys)."""
    from tinyfusers_amd.storage.state import unet_param_shapes
    from tinyfusers_amd.vision.unet import SD15, TINY, UNetModel
    for cfg, ocfg in ((SD15, oracle.SD15), (TINY, oracle.TINY)):
        assert unet_param_shapes(UNetModel(cfg, init=False)) == oracle.unet_param_shapes(ocfg)


def test_synth_is_deterministic_and_order_independent():
    from tinyfusers_amd.storage.synth import synth_normal, synth_tensor
    a = synth_tensor(0, "input_blocks.1.0.in_layers.2.weight", (320, 320, 3, 3))
    b = synth_tensor(0, "input_blocks.1.0.in_layers.2.weight", (320, 320, 3, 3))
    assert a.dtype == np.float16 and np.array_equal(a, b)
    assert abs(float(a.astype(np.float32).std()) - (320 * 9) ** -0.5) < 2e-3
    g = synth_tensor(0, "out.0.weight", (320,)).astype(np.float32)
    assert abs(g.mean() - 1.0) < 0.05
    assert not np.array_equal(synth_normal(1, "x", (8,)), synth_normal(2, "x", (8,)))


def test_schedule_matches_reference_golden(golden):
    from tinyfusers_amd.variants.sd import get_alphas_cumprod
    ac = get_alphas_cumprod()
    np.testing.assert_allclose(ac, golden["ops"]["alphas_cumprod"], rtol=1e-6)
    assert abs(ac[0] - 0.99915) < 1e-5 and abs(ac[999] - 0.0046601) < 1e-6      # SURVEY 3.3 probe values
    ts, al, ap = oracle.sampler_schedule(50)
    assert ts[0] == 1 and ts[-1] == 981 and len(ts) == 50 and ap[0] == 1.0


def test_host_ddim_matches_reference_golden(golden):
    from tinyfusers_amd.variants.sd import StableDiffusion
    g = golden["ops"]
    sd = StableDiffusion.__new__(StableDiffusion)
    xp, p0 = sd.get_x_prev_and_pred_x0(g["ddim_x"], g["ddim_e"], g["ddim_a_t"], g["ddim_a_prev"])
    np.testing.assert_allclose(xp, g["ddim_x_prev"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(p0, g["ddim_pred_x0"], rtol=1e-5, atol=1e-6)


def test_causal_mask_detection():
    from tinyfusers_amd.attention.sdpa import _is_causal_mask
    m = np.tril(np.ones((7, 7), dtype=bool))
    assert _is_causal_mask(m, 7, 7)
    add = np.where(m, 0.0, -np.inf).astype(np.float32)
    assert _is_causal_mask(add[None, None], 7, 7)
    assert not _is_causal_mask(np.ones((7, 7), dtype=bool), 7, 7)


def test_shard_range_and_arena_plan():
    from tinyfusers_amd.dist import pack_tensor, plan_arena, shard_range
    for g, w in ((8, 8), (32, 8), (5, 2), (1, 4)):
        parts = [shard_range(g, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == g
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
    offs, total = plan_arena(oracle.unet_param_shapes(oracle.SD15))
    assert total == 1719055872 and all(o % 256 == 0 for o in offs.values())     # 1.72 GB fp16
    w = np.arange(2 * 3 * 2 * 2, dtype=np.float16).reshape(2, 3, 2, 2)
    assert pack_tensor(w).shape == (2, 2, 2, 3) and pack_tensor(w)[1, 0, 1, 2] == w[1, 2, 0, 1]


def test_state_walk_cuts_cycles_but_visits_a_shared_module_under_both_names():
    """_slots (storage/state.py:4-23's recursive walk): a module two parents share is a leaf owner under BOTH dotted names (the
    round-4 visited set dropped the second name), while a back-reference to an ancestor does not make the walk spin."""
    from tinyfusers_amd.storage.state import _slots, param_shapes
    from tinyfusers_amd.ff.linear import Linear

    class Box:
        pass
    shared = Linear(8, 4, init=False)
    root, a, b = Box(), Box(), Box()
    a.proj, b.proj = shared, shared
    root.a, root.b = a, b
    a.parent = root                                        # a public back-reference: a cycle
    root.items = [shared, {"again": shared}]
    names = [n for n, *_ in _slots(root)]
    assert sorted(names) == sorted(f"{p}.{leaf}" for p in ("a.proj", "b.proj", "items.0", "items.1.again") for leaf in ("weight", "bias"))
    assert param_shapes(root)["b.proj.weight"] == (4, 8) and param_shapes(root)["a.proj.bias"] == (4,)
