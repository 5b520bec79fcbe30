import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    """True when the HIP engine library loads and sees a device (the gpu-marked tests need both)."""
    try:
        import raft_rs_amd
        return raft_rs_amd.load_library().rg_device_count() > 0
    except Exception:  # noqa: BLE001 -- library not built / no HIP runtime
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without an MI355X skips the gpu-marked tests instead of failing in rg_create.
    An explicit `-m gpu` run is NOT softened: there a missing device or library must fail loudly."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no MI355X visible (gpu-marked tests run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def rg():
    """The product package; the HIP library must already be built (python -m raft_rs_amd.build)."""
    import raft_rs_amd
    raft_rs_amd.load_library()
    return raft_rs_amd
