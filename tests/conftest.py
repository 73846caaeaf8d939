import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import knzlib
    return knzlib.Oracle()


@pytest.fixture(scope="session")
def ref():
    import knzlib
    try:
        return knzlib.Ref()
    except (RuntimeError, OSError):
        pytest.skip("reference build (oracle/_ref) not available")


@pytest.fixture(scope="session")
def golden():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")))


@pytest.fixture(scope="session")
def hip():
    """Device context. Fails loudly (no CPU fallback) when the library or the GPU is missing."""
    import importlib
    import knzlib
    knzlib.load_pkg()
    hipapi = importlib.import_module("kanzi_amd.hipapi")
    return hipapi.Context(0)
