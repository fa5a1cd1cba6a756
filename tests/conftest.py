import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_device_memory_between_tests():
    """A net owns tens (configs[4] at 1024 pairs: ~177) of GiB of device buffers through reference cycles (parameters <->
    parameter store, autograd node <-> net): they are freed only by a cycle collection, which Python schedules by host
    allocation counts, not by device memory.  Collect after every test so that the next large model finds the HBM free."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    except Exception:
        pass
