"""bench.py as the driver calls it: `python bench.py --gpus N` must start its N ranks itself (the parent makes no GPU call) and
relay rank 0's single JSON line; a WORLD_SIZE that disagrees with --gpus is an error.  Rehearsed on CPU through --dry-run
(gloo: arena broadcast + image shard + max-over-ranks, no kernels)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_bench_gpus2_launches_two_ranks_dry_run():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dry_run"] is True
    assert out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 2
    # max over ranks: rank 1 sleeps 20 ms, rank 0 10 ms
    assert out["ms_per_step"] * 3 >= 19.0
    assert out["weights"]["bytes"] > 0 and "bcast_s" in out["weights"]


def test_bench_config5_geometry_on_eight_ranks_dry_run():
    """BASELINE config 5 as the driver would start it -- 8 ranks, 4 images per rank (global batch 32), 96 x 96 latents, fp8 -- through
    the launcher and the whole N > 1 plumbing on CPU: one arena broadcast, every rank its own image shard and seeds, max over ranks."""
    r = _run(["--gpus", "8", "--images", "4", "--latent", "96", "--dtype", "fp8", "--steps", "2", "--warmup", "0", "--dry-run"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["dry_run"] is True and out["scaling"] == "weak"
    c = out["config"]
    assert c["global_batch"] == 32 and c["images_per_rank"] == 4 and c["latent"] == [4, 96, 96] and c["requested_dtype"] == "fp8" and c["parallelism"] == "dp8"
    assert out["ms_per_step"] * 2 >= 79.0                  # the slowest rank (rank 7 sleeps 80 ms) sets the time


def test_bench_dry_run_single_rank():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1


def test_bench_world_size_mismatch_is_an_error():
    # started "by torch.distributed.run" with 2 ranks but told --gpus 4: must fail loudly, not print an n_gpus=1 line
    r = _run(["--gpus", "4", "--dry-run"], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]


def test_launcher_parent_never_imports_torch():
    """The parent of an N-rank run may not touch the GPU: it must not even import torch before starting the ranks."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def dry_run")]
    assert "import torch" not in head
    body = src[src.index("def main"):]
    assert body.index("launch_ranks(args, argv)") < body.index("import torch")


def test_launcher_refuses_more_ranks_than_visible_gpus():
    """`bench.py --gpus N` on a host that shows fewer than N GPUs: ONE named error from the parent, before any rank starts.  The
    parent counts KFD topology nodes (no HIP call) cut by *_VISIBLE_DEVICES; where the count is unknown (no KFD sysfs: this container) there is no check."""
    sys.path.insert(0, ROOT)
    import bench
    have = bench.visible_gpus()
    if have is None:
        # no KFD here: the function must say "unknown", and an environment list alone must not invent devices
        os.environ["HIP_VISIBLE_DEVICES"] = "0,1"
        try:
            assert bench.visible_gpus() is None
        finally:
            del os.environ["HIP_VISIBLE_DEVICES"]
        # rehearse the refusal with a fake topology
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            for i, simd in enumerate((0, 1024, 1024)):            # node 0 = the CPU
                os.makedirs(os.path.join(d, str(i)))
                open(os.path.join(d, str(i), "properties"), "w").write(f"cpu_cores_count 0\nsimd_count {simd}\n")
            real_listdir, real_open = os.listdir, open
            base = "/sys/class/kfd/kfd/topology/nodes"
            import builtins
            try:
                os.listdir = lambda p=".": real_listdir(d if p == base else p)
                builtins.open = lambda f, *a, **k: real_open(f.replace(base, d) if isinstance(f, str) else f, *a, **k)
                assert bench.visible_gpus() == 2
                os.environ["ROCR_VISIBLE_DEVICES"] = "1"
                assert bench.visible_gpus() == 1
                rc = bench.launch_ranks(bench.parse(["--gpus", "2"]), ["--gpus", "2"])
                assert rc == 2
            finally:
                os.listdir, builtins.open = real_listdir, real_open
                os.environ.pop("ROCR_VISIBLE_DEVICES", None)
    else:
        r = _run(["--gpus", str(have + 1), "--steps", "1", "--warmup", "0"])
        assert r.returncode == 2 and f"only {have} GPU" in r.stderr
        assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
