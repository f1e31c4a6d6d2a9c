import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# One HIP / HSA runtime per process: PyTorch ships its own libamdhip64 + libhsa-runtime64, libi3rc_hip.so links
# /opt/rocm's.  Whichever is loaded first serves both (same SONAMEs) -- but only torch-first works: with ROCm's
# runtime already holding the device, torch's later initialisation finds "No HIP GPUs".  The tests that hand torch
# tensors / streams to the C ABI (and bench.py, multigpu.py) therefore import torch before the library is loaded.
try:
    import torch  # noqa: F401
except ImportError:   # the library itself does not need it
    pass


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
