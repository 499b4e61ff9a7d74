"""bench.py and the tools the GPU runs call are only ever executed on the GPU box: a name that is not defined anywhere (a slip of an
edit) would show up there, minutes later, as a missing bench line.  Checked here with the compiler's symbol tables: every name a function
treats as GLOBAL must exist in its module or in builtins."""
import builtins
import os
import symtable

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["bench.py", "__graft_entry__.py", "tools/slab_proxy.py", "tools/timeline.py", "tools/peer_ranks_check.py", "tools/run_case.py",
         "2d-lb_amd/LB_D2Q9/slabs.py", "2d-lb_amd/LB_D2Q9/simulation.py", "2d-lb_amd/LB_D2Q9/_native.py"]


def _undefined(path):
    src = open(path).read()
    top = symtable.symtable(src, path, "exec")
    module_names = set(s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace())
    bad = []

    def walk(tab):
        for child in tab.get_children():
            if child.get_type() in ("function", "class"):
                for s in child.get_symbols():
                    if s.is_global() and s.is_referenced() and not s.is_assigned():
                        n = s.get_name()
                        if n not in module_names and not hasattr(builtins, n) and n not in ("__file__", "__name__", "__doc__"):
                            bad.append("%s: %s" % (child.get_name(), n))
            walk(child)
    walk(top)
    return bad


@pytest.mark.parametrize("rel", FILES)
def test_every_global_name_a_function_uses_exists(rel):
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        pytest.skip("no such file")
    assert _undefined(path) == []
