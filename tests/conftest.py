import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def hiplib():
    """The product library.  GPU tests fail loudly (never skip) when it is missing."""
    import openwurli_amd
    if not os.path.exists(openwurli_amd.library_path()):
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "build.sh")])
    return openwurli_amd.load_library()


@pytest.fixture(scope="session")
def hiplib_host(hiplib):
    """Same library, for tests that only call its host-side entry points (WAV writer, quantisers): they need no device."""
    return hiplib
