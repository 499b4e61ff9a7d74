"""The drop-in boundary without a GPU: liblbhip.so loads, exports every symbol include/lb_hip.h
declares, reports errors through status codes, and the product never touches the oracle."""
import ctypes as ct
import os
import re

import pytest

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "lb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lb_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lbhip):
    from LB_D2Q9 import _native
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lbhip, n), "liblbhip.so does not export %s" % n
    assert sorted(_native.EXPORTS) == names          # the binding covers exactly the header


def test_abi_version_matches_header(lbhip):
    text = open(os.path.join(ROOT, "include", "lb_hip.h")).read()
    assert lbhip.lb_abi_version() == int(re.search(r"#define LB_ABI_VERSION (\d+)", text).group(1))


def test_params_struct_layout_matches_header():
    from LB_D2Q9 import _native
    assert ct.sizeof(_native.LbParams) == 16 * 4     # 6 int32 + 5 float + flags + 4 reserved
    assert _native.LbParams.flags.offset == 44


def test_errors_are_status_codes_not_exceptions(lbhip):
    from LB_D2Q9 import _native
    assert lbhip.lb_create(None, None) == -1                         # LB_ERR_ARG
    assert b"null" in lbhip.lb_last_error()
    p = _native.LbParams()
    p.nx, p.ny, p.local_ny, p.omega = 1, 1, 1, 1.0
    h = ct.c_void_p()
    assert lbhip.lb_create(ct.byref(p), ct.byref(h)) == -1 and not h.value
    p.nx, p.ny, p.local_ny, p.omega = 8, 8, 8, 2.5
    assert lbhip.lb_create(ct.byref(p), ct.byref(h)) == -1 and b"omega" in lbhip.lb_last_error()
    assert lbhip.lb_run(None, 1) == -1 and lbhip.lb_destroy(None) == 0


def test_no_silent_cpu_fallback(lbhip):
    """Without a GPU the product must fail loudly (this test is skipped on the GPU box)."""
    from LB_D2Q9 import _native
    from LB_D2Q9.simulation import Simulation
    if lbhip.lb_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_native.LbError):
        Simulation(16, 16, 1.0)
    with pytest.raises(_native.LbError):
        _native.device_count()


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "2d-lb_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(base, f)).read()
                for line in text.splitlines():
                    if re.search(r"^\s*(from|import)\s+.*oracle", line) or "d2q9_oracle.so" in line:
                        raise AssertionError("%s references the oracle: %s" % (f, line.strip()))
