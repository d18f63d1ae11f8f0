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


@pytest.fixture(autouse=True)
def _poison_free_memory(request):
    """NDJIR_TEST_POISON=<GiB>: before every GPU test, fill that much of the allocator's free memory with NaN and hand it
    back -- the test's `torch.empty` buffers then start as NaN, so a kernel that reads memory nobody wrote shows up as a
    parity failure instead of passing on zeros."""
    gib = os.environ.get("NDJIR_TEST_POISON")
    if gib and request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
            chunks = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(int(gib))]      # 1 GiB each
            del chunks
    yield
