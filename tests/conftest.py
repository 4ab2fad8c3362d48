import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import truely_amd  # noqa: E402  (import shim for the hyphenated package directory)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, loudly
    from truely_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libtrl_oracle.so")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def state_dicts():
    return truely_amd.weights.synthetic_state_dicts(0)


@pytest.fixture(scope="session")
def blob(state_dicts):
    return truely_amd.weights.pack_state_dicts(*state_dicts)


@pytest.fixture(scope="session")
def oracle(blob):
    from oracle.oracle import Oracle
    return Oracle(blob)


@pytest.fixture(scope="session")
def engine(blob):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from truely_amd.engine import Engine
    return Engine(blob)


@pytest.fixture(scope="session")
def engine_generic(blob):
    """Same weights, PNet through the generic layer kernels (validation path)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from truely_amd.engine import Engine
    return Engine(blob, pnet_mode=1)


def frames_small(n=6, H=180, W=320, seed=3):
    return truely_amd.synthetic.synthetic_frames(n, H, W, seed=seed)
