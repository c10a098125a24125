"""The ablation tools (pp_dbg.py, gemm_dbg.py, geglu_dbg.py, launch_floor.py, sdpa_dbg.py) time kernels that skip work and return wrong
results by design.  Those instances are not in the shipped library: they live in a second one built from the same sources with
-DTF_ABLATION (`python -m tinyfusers_amd.build --ablation`), which these tools load explicitly through TF_LIB_PATH -- call use() BEFORE
importing tinyfusers_amd."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tinyfusers_amd", "lib", "libtinyfusers_hip_ablation.so")


def use():
    if not os.path.exists(LIB):
        raise SystemExit("ablation library missing: run `python -m tinyfusers_amd.build --ablation` first")
    os.environ["TF_LIB_PATH"] = LIB
    return LIB
