import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (an entry point: a hardware queue per stream, before the runtime starts)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libs():
    """Build the oracle (.so, gcc) and the product library (hipcc cross-compiles
    for gfx950 without a GPU) if they are not there yet."""
    import subprocess

    if not os.path.exists(os.path.join(ROOT, "oracle", "liborc_diff.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(os.path.join(ROOT, "grav1synth_amd", "libg1s_diff.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grav1synth_amd", "csrc")])
