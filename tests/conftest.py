import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lab: needs the lab library (MGX_LIB=mini_amd/libmgx_lab.so, built by "
                                       "`python __graft_entry__.py --lab`): experiment shapes that are not in the product")


LAB_ENV_KEYS = ("MGX_BFS_FLAGS", "MGX_BFS_DENSE_DIAG", "MGX_BFS_BUILD_DIAG")


def needs_lab(env):
    """does this set of environment switches select a shape that only the lab library holds?"""
    return any(k in env for k in LAB_ENV_KEYS)


def skip_unless_lab(env=None):
    import mini_amd
    if (env is None or needs_lab(env)) and not mini_amd.lib.mgx_build_is_lab():
        pytest.skip("lab shape: run with MGX_LIB=mini_amd/libmgx_lab.so")


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
