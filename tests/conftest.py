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
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def gpu():
    """The product package with its HIP library loaded on a machine that has a GPU."""
    import iq_tool_amd
    lib = iq_tool_amd.load()          # raises if libiqgpu.so is missing: no fallback
    if lib.iqgpu_device_count() < 1:
        pytest.fail("test marked gpu but no HIP device is visible")
    return iq_tool_amd
