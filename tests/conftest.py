import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (gfx950) device; run with -m gpu")


def _have_gpu():
    try:
        import heracles_amd

        return heracles_amd.device_count() > 0
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def rng():
    # same seed as the reference's session fixture (tests/conftest.py:20-22)
    return np.random.default_rng(50)


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "reference_numpy.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle import hxoracle

    hxoracle.lib()
    return hxoracle
