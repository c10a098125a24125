import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes about a minute (full-size oracle vs reference trajectory)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    return {n: np.load(os.path.join(d, n + ".npz")) for n in ("ops", "blocks", "unet_sd15", "unet50_sd15")
            if os.path.exists(os.path.join(d, n + ".npz"))}
