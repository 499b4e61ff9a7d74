import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ablate
lib = os.path.join(ROOT, "2d-lb_amd", "LB_D2Q9", "liblbhip_diag.so")
for rep in range(4):
    for diag in (0, 32768):
        env = dict(os.environ, LB_LIB=lib, LB_DIAG=str(diag))
        out = subprocess.run([sys.executable, "-c", ablate.CHILD, "8192", "353"], env=env, capture_output=True, text=True)
        print(diag, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-300:], flush=True)
