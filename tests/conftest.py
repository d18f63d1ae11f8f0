import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    # the product has no CPU path: a missing extension must fail, not skip
    from ndjir_amd import lib
    lib.load()
    return torch.device("cuda:0")


def pytest_collection_modifyitems(config, items):
    """NDJIR_TEST_SHUFFLE=<seed>: run the collected tests in a seeded random order (finds state that leaks from one test
    into another: the process-global registries of accumulate-in-place gradient buffers, packed-weight caches, tile settings)."""
    seed = os.environ.get("NDJIR_TEST_SHUFFLE")
    if seed:
        import random
        random.Random(int(seed)).shuffle(items)
