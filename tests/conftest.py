import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only; skipped elsewhere)")


def golden(name):
    import numpy as np

    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def gpu_ctx():
    """A model-less context for operator-level calls."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from bevgen_amd.runtime import Context

    ctx = Context(None)
    yield ctx
    ctx.close()
