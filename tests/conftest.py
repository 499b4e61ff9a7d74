import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "2d-lb_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); builds oracle/_build/libd2q9_oracle.so on first use."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def lbhip():
    """The product library; built by __graft_entry__.build() / 2d-lb_amd/build.py."""
    from LB_D2Q9 import _native
    if not os.path.exists(_native.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("lb_build", os.path.join(ROOT, "2d-lb_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    return _native.lib()
