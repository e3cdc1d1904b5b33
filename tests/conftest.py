import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from helm_amd import _native
        return _native.hip.helm_hip_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests FAIL (not skip) on a GPU-less box if explicitly selected with -m gpu:
    # the product path has no CPU fallback and must say so loudly.
    pass
