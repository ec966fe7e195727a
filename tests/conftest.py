import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "ref: needs the compiled reference oracle/_ref/libptref.so (skipped where it is absent)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle as po
    return po.Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle import pyoracle as po
    if not po.Ref.available():
        pytest.skip("oracle/_ref/libptref.so not built here (no /root/reference); golden fixtures cover this")
    return po.Ref()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def renderer():
    import pathtracer_amd as pt
    if pt.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible")
    r = pt.Renderer(0)
    yield r
    r.close()
