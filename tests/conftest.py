import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_api

    oracle_api.lib()
    return oracle_api


@pytest.fixture(scope="session")
def pt():
    import rust_pathtracer_amd as pt

    return pt


@pytest.fixture(scope="session")
def gpu_ctx(pt):
    """One device context for the whole GPU test session (fails loudly when there is no GPU)."""
    ctx = pt.Context(0)
    yield ctx
    ctx.close()
