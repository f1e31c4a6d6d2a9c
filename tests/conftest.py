import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


def pytest_collection_modifyitems(config, items):
    # a hung GPU test must not eat the box: 5 minutes per test at most (pytest-timeout)
    for item in items:
        if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(300))
