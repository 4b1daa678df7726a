import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    # the CPU oracle's operators are small: on a 256-thread host torch's default (one thread per logical core) is several times SLOWER than 32 threads
    # (bench.py's cpu_baseline tunes the same knob); results do not depend on the thread count on the fixtures
    try:
        import torch

        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    except Exception:
        pass
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
