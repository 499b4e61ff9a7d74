#!/usr/bin/env python3
"""A/B of LB_DIAG switches of the diagnostic build on k_step4, whole processes alternated (timing only).
    python tools/layout_ab.py 0 65536 131072"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ablate

lib = os.path.join(ROOT, "2d-lb_amd", "LB_D2Q9", "liblbhip_diag.so")
diags = [int(a) for a in sys.argv[1:]] or [0]
for rep in range(3):
    for diag in diags:
        env = dict(os.environ, LB_LIB=lib, LB_DIAG=str(diag))
        out = subprocess.run([sys.executable, "-c", ablate.CHILD, "8192", "353"], env=env, capture_output=True, text=True)
        print(diag, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-300:], flush=True)
