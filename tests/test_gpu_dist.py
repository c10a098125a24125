"""The N > 1 plumbing of bench.py on the one-GPU box: one rank under torch.distributed.run with the collective forced on
(TF_BENCH_FORCE_DIST), once with torch.distributed carrying the weight-arena broadcast and once with the C-ABI's own
tf_comm_unique_id / tf_comm_init_rank / tf_bcast (bench.py --comm tf, tinyfusers_amd.dist.TfComm); and the shipped tuning table in
table-only mode (what every rank of a multi-GPU run uses) on BASELINE config 2's step."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench_one_rank(comm):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, TF_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-e2e", "--no-config5", "--comm", comm]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])


@pytest.mark.parametrize("comm", ["tf", "torch"])
def test_bench_one_rank_with_the_collective_forced(comm):
    out = _bench_one_rank(comm)
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["value"] > 0
    assert out["weights"]["bcast_s"] > 0                      # the broadcast ran
    assert out["weights"]["bcast_via"].startswith("tf_bcast" if comm == "tf" else "torch.distributed")


def test_table_only_mode_runs_config2_and_names_a_missing_shape():
    """tf_gemm_autotune(2): the shipped table serves every GEMM of BASELINE config 2's step without tuning; a shape it does not hold is an
    error that names the shape (so that no rank of a multi-GPU run ever times kernels on its own)."""
    import ctypes
    import tinyfusers_amd.storage.tensor as T
    from tinyfusers_amd.native import hip, lib
    from tinyfusers_amd.ff.linear import linear_f16 as linear
    T.ensure_init(0)
    x = T.DeviceArray.from_numpy(np.ones((24, 72), np.float16)); w = T.DeviceArray.from_numpy(np.ones((40, 72), np.float16))
    hip.tf_gemm_autotune(2)
    try:
        with pytest.raises(RuntimeError, match=r"M=24 N=40 K=72 .* not in the tuning table"):
            linear(x, w, None)
    finally:
        hip.tf_gemm_autotune(1)
    y = linear(x, w, None).numpy()
    assert np.array_equal(y, np.full((24, 40), 72, np.float16))
