"""The C ABI used from plain C: examples/poiseuille_c_abi.c includes include/lb_hip.h, links liblbhip.so (no Python, no
torch in that process) and checks a Poiseuille profile.  Building it is a CPU test, running it needs the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "2d-lb_amd", "LB_D2Q9")


def build(tmp_path):
    exe = str(tmp_path / "poiseuille_c_abi")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "poiseuille_c_abi.c"), "-L", LIBDIR, "-llbhip",
                           "-Wl,-rpath," + LIBDIR, "-lm", "-o", exe])
    return exe


def test_c_example_compiles_and_links_against_the_abi(lbhip, tmp_path):
    exe = build(tmp_path)
    assert os.path.exists(exe)
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "liblbhip.so" in needed


@pytest.mark.gpu
def test_c_example_runs_a_poiseuille_pipe(lbhip, tmp_path):
    exe = build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "pipe" in out.stdout and "% of the peak" in out.stdout
