import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle
    return pyoracle.lib()


@pytest.fixture(scope="session")
def gpu():
    """The product library on a gfx950 device; the tests FAIL (not skip) when it is unusable."""
    import slide_slam_amd as s
    s.device_check()
    return s
