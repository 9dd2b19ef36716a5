import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a GPU test whose CPU-oracle side takes minutes (still part of -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _oracle_lib():
    """The C part of the oracle is built on demand (gcc only; seconds)."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "libccn_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    yield


@pytest.fixture(autouse=True)
def _seed_everything():
    """torch's initial seed is drawn at random per process: every test starts from the same global generator
    state, so that modules built without an explicit seed get the same parameters in every run."""
    import numpy as np
    import torch
    torch.manual_seed(20261003)
    np.random.seed(20261003)
    yield
