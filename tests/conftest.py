import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "image-retrieval-wavelet_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    """The C restatement is test infrastructure: build it on demand (gcc only, < 1 s)."""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture
def diag(monkeypatch):
    """monkeypatch with the DIAGNOSTIC build of the library active (libwvhash_diag.so: same objects, linked with
    csrc/tune_diag.cpp): the WV_* switches that pin one kernel variant / code path exist only there -- the release
    library never reads the environment."""
    from wvhash import _lib
    with _lib.diagnostic():
        yield monkeypatch
