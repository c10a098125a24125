"""CPU oracle for the tinyfusers SD-1.x UNet denoising path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: it may be imported
by ``tests/``, by ``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg, always
as the *checker*, never as the thing measured or shipped.  The product path
(``tinyfusers_amd``) never imports it and fails loudly if the HIP library is missing.

Parity status: PINNED.  Every function here is a restatement (numpy / torch-CPU, fp32) of a
reference function, cited file:line against ``/root/reference``, and is checked against outputs
of the reference's own Python run in the authoring container (``tests/golden/make_golden.py``
imports the reference under cupy->numpy / cudnn->torch stand-in modules and writes
``tests/golden/*.npz``); ``tests/test_oracle_golden.py`` re-checks the oracle against those
fixtures on every CPU test run.
"""
from .ops import *  # noqa: F401,F403
from .unet import *  # noqa: F401,F403
from .vae import *  # noqa: F401,F403
from .clip import *  # noqa: F401,F403
