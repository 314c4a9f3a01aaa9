import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# HX_TEST_HERACLES_SHIM=1 (set by tests/test_production_config.py for its child run, in the build container only): make
# ``heracles.core`` / ``heracles.result`` importable from /root/reference through a bare package object (the
# package's own __init__ needs fitsio) BEFORE heracles_amd is imported, so that heracles_amd.core takes the branch a user with an
# installed Heracles gets: the reference's own TocDict / toc_match / update_metadata / Result.  Nothing of the reference is copied.
if os.environ.get("HX_TEST_HERACLES_SHIM") == "1":
    import types

    if not os.path.isdir("/root/reference/heracles"):
        raise RuntimeError("HX_TEST_HERACLES_SHIM=1 without /root/reference")
    _pkg = types.ModuleType("heracles")
    _pkg.__path__ = ["/root/reference/heracles"]
    sys.modules["heracles"] = _pkg


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
