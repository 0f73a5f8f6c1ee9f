import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.lib()
    return O


@pytest.fixture(scope="session")
def built_lib():
    """libststhip.so, built on demand (hipcc cross-compiles without a GPU)."""
    from stencilstream_amd import capi

    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return capi.load()


@pytest.fixture(scope="session")
def gpu(built_lib):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("a test marked gpu ran without a GPU; the HIP path has no CPU fallback")
    from stencilstream_amd import capi

    capi.init(0)
    return torch.device("cuda:0")
