import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oraclelib

    return oraclelib.Oracle()


@pytest.fixture(scope="session")
def reference():
    import oraclelib

    if not oraclelib.Reference.available():
        pytest.skip("compiled reference (oracle/_ref/libmcref.so) not built here")
    return oraclelib.Reference()


@pytest.fixture(scope="session")
def mcrt():
    import minecraftskin_raytracer_amd as M

    return M


@pytest.fixture(scope="session")
def gpu(mcrt):
    if mcrt.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (libmcrt has no CPU fallback)")
    return 0
