#!/usr/bin/env python3
"""Sampler driver -- mirrors example/sd1.py:34-79 of the reference: load weights, run the prompt's token ids through the
CLIP text encoder for the two contexts (:44-49), timesteps = range(1,1000,1000//steps), reversed loop, VAE decode.
No checkpoint or BPE vocabulary exists offline, so by default the weights are the seeded synthetic ones and
the "prompt" is a seeded list of token ids; ``--ckpt file.ckpt|file.safetensors`` reads real LDM weights through
storage/unpicker.py + update_state instead (example/sd1.py:40-41), ``--vocab bpe_simple_vocab_16e6.txt.gz --prompt "..."``
tokenizes a real prompt (tokenizer/clip.py).
BASELINE config 3: full 50-step sampler, batch 1, end-to-end img/s.

    python -m example.sd1 --steps 50 [--ckpt sd-v1-4.ckpt] [--out rendered.npy]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Run the SD-1.x sampler on MI355X (synthetic weights)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--guidance", type=float, default=7.5)
    ap.add_argument("--images", type=int, default=3, help="images to time after the first (compile + warm-up) one")
    ap.add_argument("--out", default="")
    ap.add_argument("--ckpt", default="", help="LDM checkpoint (.ckpt torch zip or .safetensors); default: synthetic weights")
    ap.add_argument("--vocab", default="", help="local bpe_simple_vocab_16e6.txt.gz for the prompt; default: seeded token ids")
    ap.add_argument("--prompt", default="a horse sized cat eating a bagel")
    args = ap.parse_args()

    import tinyfusers_amd.storage.tensor as T
    from tinyfusers_amd.storage.state import param_shapes, update_state
    from tinyfusers_amd.storage.synth import synth_normal, synth_state_dict
    from tinyfusers_amd.variants.sd import StableDiffusion

    T.ensure_init(0)
    model = StableDiffusion()
    t0 = time.time()
    import io, contextlib
    if args.ckpt:
        from tinyfusers_amd.storage.unpicker import load_checkpoint
        state = load_checkpoint(args.ckpt)                       # memory-mapped; update_state streams tensor by tensor
    else:
        state = synth_state_dict(param_shapes(model), 0)       # UNet + VAE decoder + CLIP text encoder, by LDM name
    with contextlib.redirect_stdout(io.StringIO()):
        update_state(model, state, "")
    del state
    print(f"weights installed in {time.time() - t0:.1f}s")
    # run through CLIP to get the contexts (example/sd1.py:44-49); token ids stand in for tokenizer.encode(prompt)
    if args.vocab:
        from tinyfusers_amd.tokenizer.clip import ClipTokenizer
        tokenizer = ClipTokenizer(args.vocab)
        prompt, empty = np.array([tokenizer.encode(args.prompt)]), np.array([tokenizer.encode("")])
    else:
        rng = np.random.default_rng(args.seed)
        n_words = 9
        prompt = np.full((1, 77), 49407, dtype=np.int64); prompt[0, 0] = 49406; prompt[0, 1:1 + n_words] = rng.integers(0, 49406, n_words)
        empty = np.full((1, 77), 49407, dtype=np.int64); empty[0, 0] = 49406
    text_model = model.cond_stage_model.transformer.text_model
    text_model(prompt)                                           # first call folds the LayerNorms / fuses q|k|v once
    T.hip.tf_stream_sync(None)
    t0 = time.perf_counter()
    context = text_model(prompt)
    unconditional_context = text_model(empty)
    T.hip.tf_stream_sync(None)
    print(f"CLIP context: {context.shape}, unconditional CLIP context: {unconditional_context.shape}  ({1e3 * (time.perf_counter() - t0):.2f} ms for both)")
    timesteps = list(range(1, 1000, 1000 // args.steps))
    alphas = model.alphas_cumprod[timesteps]
    alphas_prev = np.concatenate((np.array([1.0]), alphas[:-1])).astype(np.float32)
    latent = model.latent_from_numpy(synth_normal(args.seed, "sd.latent", (1, 4, 64, 64)))
    model.compile(unconditional_context, context, latent)
    times = []
    for n in range(args.images + 1):
        model.set_latent(synth_normal(args.seed + n, "sd.latent", (1, 4, 64, 64)))
        t0 = time.perf_counter()
        for index, timestep in list(enumerate(timesteps))[::-1]:
            model.step(timestep, alphas[index], alphas_prev[index], args.guidance)
        model.synchronize()
        t1 = time.perf_counter()
        assert np.isfinite(latent.numpy()).all(), f"image {n}: the sampler produced a non-finite latent"
        with T.use_stream(model._stream):
            x = model.decode(latent)
        t2 = time.perf_counter()
        times.append((t1 - t0, t2 - t1))
        print(f"image {n}: {args.steps} steps {1e3 * (t1 - t0):.1f} ms ({args.steps / (t1 - t0):.1f} steps/s), decode {1e3 * (t2 - t1):.1f} ms, image {x.shape} mean {x.mean():.1f}")
    s, d = np.median([t[0] for t in times[1:]]), np.median([t[1] for t in times[1:]])
    print(f"end-to-end (sampler + VAE decode, batch 1): {1.0 / (s + d):.3f} img/s  [{args.steps} steps {1e3 * s:.1f} ms + decode {1e3 * d:.1f} ms]")
    if args.out:
        np.save(args.out, x)
