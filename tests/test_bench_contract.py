"""bench.py's host-side pieces without a GPU: the synthetic workload, the CPU-baseline leg, and the
refusal to run (rather than fall back) when no GPU is present."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_shear_layer_shape_and_slab_consistency():
    import bench
    rho, u, v = bench.shear_layer(64, 48, 0, 48)
    for a in (rho, u, v):
        assert a.shape == (64, 48) and a.dtype == np.float32 and a.flags.f_contiguous
    assert np.all(rho == 1) and abs(u).max() <= 0.04 + 1e-7 and abs(v).max() <= 4.1e-5
    assert u[:, 12].max() == pytest.approx(0., abs=1e-7)            # shear layers at ny/4 and 3ny/4
    r2, u2, v2 = bench.shear_layer(64, 48, 16, 16)                   # a slab sees the same rows
    assert np.array_equal(u2, u[:, 16:32]) and np.array_equal(v2, v[:, 16:32])


def test_cpu_baseline_leg_reports_the_contract_keys():
    import bench
    r = bench.cpu_baseline(budget_s=0.5, n=128, all_cores_budget_s=0.3)
    assert set(r) == {"value", "unit", "cores", "kind", "sample", "all_cores"}
    assert r["all_cores"]["cores"] >= 1 and r["all_cores"]["value"] > 0
    assert r["kind"] == "port" and r["cores"] == 1 and r["unit"] == "MLUPS" and r["value"] > 0.1


def test_bench_refuses_to_run_without_a_gpu(lbhip):
    if lbhip.lb_device_count() > 0:
        pytest.skip("a GPU is visible")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU" in (p.stderr + p.stdout)
    assert not p.stdout.strip().startswith("{")


def test_committed_bench_line_has_the_contract_fields():
    path = os.path.join(ROOT, "profiles", "r01_bench_n1.json")
    line = [l for l in open(path) if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    assert d["roofline"]["frac"] == pytest.approx(d["roofline"]["achieved"] / 8000.0, abs=1e-3)
    assert "workload" in d["config"] and d["vs_baseline"] is None
    assert d["roofline"]["steps_per_launch"] in (1, 2, 3, 4)
