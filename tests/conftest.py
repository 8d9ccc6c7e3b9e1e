import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def extractor_vga():
    from slam_amd.extractor import ORBextractor
    ex = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=8)
    yield ex
    ex.close()
