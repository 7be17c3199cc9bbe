import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ctx():
    """one GPU context for the whole session (gpu tests only)"""
    import pgvector_amd
    c = pgvector_amd.api.Context(0, stream=0)  # the default stream: ordered with torch tensors made in tests
    yield c
    c.close()
