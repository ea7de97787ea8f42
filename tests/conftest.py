import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "global-motion-estimation_amd"), REPO, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def _ensure_native_build():
    """A fresh checkout has no libgme_hip.so (build products are not tracked): build it once if
    hipcc is here.  On the GPU box the prebuilt library travels with the snapshot."""
    import shutil
    import subprocess
    lib = os.path.join(REPO, "global-motion-estimation_amd", "lib", "libgme_hip.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "global-motion-estimation_amd", "csrc"), "-j4"])


def pytest_configure(config):
    _ensure_native_build()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than a few seconds")


@pytest.fixture(scope="session")
def golden():
    from helpers import Golden
    return Golden()
