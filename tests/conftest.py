import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_report_header(config):
    """Which GPU the run is on (the serial number: two GPUs of one pool did not behave alike in round 4, tests/test_gpu_00_selfcheck.py)."""
    try:
        import subprocess
        out = subprocess.run(["rocm-smi", "--showserial"], capture_output=True, text=True, timeout=20).stdout
        serials = [ln.split(":")[-1].strip() for ln in out.splitlines() if "Serial Number:" in ln]
        return "GPU serial: " + (", ".join(serials) if serials else "none found")
    except Exception:
        return "GPU serial: rocm-smi not available"
