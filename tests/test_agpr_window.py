"""k_deep's row in flight lives in a fixed window of accumulation registers that only its asm blocks may write and only the copies
behind its hand-written wait may read (csrc/kernels_deep.h: deep_row_issue / deep_row_take).  The compiler is told the window is
clobbered, not that it is reserved: the device code of the built objects is checked for strays (tools/check_agpr_window.py)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_agpr_window", os.path.join(ROOT, "tools", "check_agpr_window.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("unit", ["deep6.o", "deep7.o"])
def test_nothing_but_the_asm_blocks_touches_the_window(unit):
    obj = os.path.join(ROOT, "2d-lb_amd", "build", unit)
    if not os.path.exists(obj):
        pytest.skip("objects not built here (__graft_entry__.build() leaves them in 2d-lb_amd/build)")
    tool = _tool()
    if not os.path.exists(tool.LLVM + "/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    loads, reads, strays = tool.check(obj)
    assert loads > 0 and reads > 0, "the hand-waited gather is not in this object"
    assert not strays, strays[:5]
