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


# ---- statistical retries are not silent --------------------------------------------------------------------------
# The 3-sigma parity tests re-examine a first miss once on an independent, larger sample (test_gpu_parity._parity,
# test_gpu_baseline_configs._two_stage).  Every such first miss is recorded here, raised as a warning and listed in a
# summary line at the end of the session: a slowly growing bias shows first as stage-1 misses.
STAGE1_MISSES = []


def record_stage1_miss(what, detail):
    import warnings

    STAGE1_MISSES.append((what, str(detail)[:300]))
    warnings.warn(f"statistical stage-1 miss in {what}: {str(detail)[:300]} (re-examined on an independent larger sample)",
                  stacklevel=2)


def pytest_terminal_summary(terminalreporter):
    tr = terminalreporter
    tr.write_sep("-", f"statistical stage-1 misses this session: {len(STAGE1_MISSES)}")
    for what, detail in STAGE1_MISSES:
        tr.write_line(f"  stage-1 miss: {what}: {detail}")
