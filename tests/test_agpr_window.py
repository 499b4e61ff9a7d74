"""k_deep's row in flight lives in a fixed window of accumulation registers that only its asm blocks may write and only the copies
behind its hand-written wait may read, and that wait counts the stores of one steady iteration (csrc/kernels_deep.h: deep_row_issue /
deep_row_take).  The compiler is told the window is clobbered, not that it is reserved: the device code of the LIBRARY THAT RUNS is
disassembled and checked for strays and for the store counts (tools/check_agpr_window.py; build.py runs the same check on every
product build and refuses the build on a finding).  Runs wherever the library and llvm-objdump exist -- here and on the GPU box."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_agpr_window", os.path.join(ROOT, "tools", "check_agpr_window.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_nothing_but_the_asm_blocks_touches_the_window_and_the_waits_count_the_stores():
    lib = os.path.join(ROOT, "2d-lb_amd", "LB_D2Q9", "liblbhip.so")
    tool = _tool()
    assert os.path.exists(lib), "library not built (__graft_entry__.build())"
    if not os.path.exists(tool.LLVM + "/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    loads, reads, strays, waits, problems = tool.check_all(lib)
    assert loads > 0 and reads > 0 and waits > 0, "the hand-waited gather is not in this library"
    assert not strays, strays[:5]
    assert not problems, problems[:5]


def test_the_checker_sees_a_stray_and_a_miscount():
    """the checker itself: a made-up disassembly with one instruction writing the window, and a kernel one store short"""
    tool = _tool()
    good = ["0000000000001000 <k_deep_ok>:"]
    good += ["\tbuffer_load_dwordx4 a[%d:%d], v1, s[0:3], 0 offen // 000000001000: 00" % (tool.LO + 4 * k, tool.LO + 3 + 4 * k) for k in range(9)]
    good += ["\ts_waitcnt vmcnt(9) // 000000001100: 00", "\tv_accvgpr_read_b32 v5, a%d // 000000001104: 00" % tool.LO]
    good += ["\tbuffer_store_dwordx4 v[0:3], v1, s[0:3], 0 offen nt // 000000001200: 00"] * 9
    good += ["\tbuffer_store_dwordx4 v[0:3], v1, s[0:3], 0 offen // 000000001300: 00"] * 9
    text = "\n".join(good)
    assert tool.check(None, text) == (9, 1, []) and tool.check_waits(None, text) == (1, [])
    stray = text + "\n\tv_accvgpr_write_b32 a%d, v3" % (tool.LO + 8) + "  // 000000001400: 00"
    assert len(tool.check(None, stray)[2]) == 1
    short = "\n".join(good[:-1])
    assert len(tool.check_waits(None, short)[1]) == 1
    extra = text + "\n\tglobal_load_dword v7, v[2:3], off // 000000001500: 00"
    assert len(tool.check_waits(None, extra)[1]) == 1
