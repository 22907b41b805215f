import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the native pieces once (no-op when up to date)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "mtsv_tools_amd", "csrc"), "-j8"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libmtsv_oracle.so"])
    if os.path.exists("/root/reference/ssw/src/ssw.c") and not os.path.exists(
            os.path.join(ROOT, "oracle", "_ref", "libssw_ref.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
