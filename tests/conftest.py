import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

PKG = "rt-depth-map_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built libraries: build them once (what __graft_entry__.build() does); hipcc cross-compiles
    # gfx950 without a GPU.  The PRODUCT never builds or falls back on its own: a missing library is an OSError there.
    lib = os.path.join(ROOT, PKG, "lib", "librtdm_hip.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(ROOT, PKG, "lib", "host_selftest")):
        import subprocess
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, PKG)])


def load(sub=None):
    """The package directory has a hyphen in its name, so it is imported through importlib."""
    return importlib.import_module(PKG + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def synth():
    return load("synth")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
