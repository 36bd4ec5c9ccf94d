import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure libmgx.so and liboracle.so exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from tests.oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def torch_mod():
    import torch
    return torch


@pytest.fixture(scope="session")
def gpu_ctx(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mini_amd
    ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
    yield ctx
