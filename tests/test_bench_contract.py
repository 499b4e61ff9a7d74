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
    r = bench.cpu_baseline(budgets=((64, 0.2), (128, 0.4)), all_cores_budget_s=0.3)
    assert set(r) == {"value", "unit", "cores", "kind", "sample", "sizes", "all_cores", "product_cpu_backend"}
    own = r["product_cpu_backend"]                                     # the product's own CPU backend, timed beside the oracle's port
    assert own["kind"] == "own" and own["cores"] == 1 and own["value"] > 0.1 and "LB_DEVICE_CPU" in own["sample"]
    assert r["all_cores"]["cores"] >= 1 and r["all_cores"]["value"] > 0
    assert r["kind"] == "port" and r["cores"] == 1 and r["unit"] == "MLUPS" and r["value"] > 0.1
    assert "numpy2=True" in r["sample"]                                # the mode pinned bit-exact to the imported reference
    assert [s["grid"] for s in r["sizes"]] == [[64, 64], [128, 128]] and r["value"] == r["sizes"][-1]["value"]


def test_pmc_traffic_lookup_says_why_when_there_is_no_profile():
    import bench
    b, src, per = bench.load_pmc_traffic(8192, [4, 4, 6, 6])
    assert b and "committed profile" in src and sorted(per) == [4, 6] and b == pytest.approx((per[4] + per[6]) / 2.0)
    b, src, per = bench.load_pmc_traffic("c9/123", [4])
    assert b is None and per is None and "no --pmc profile" in src and "8192/4" in src


def test_bench_refuses_to_run_without_a_gpu(lbhip):
    if lbhip.lb_device_count() > 0:
        pytest.skip("a GPU is visible")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU" in (p.stderr + p.stdout)
    assert not p.stdout.strip().startswith("{")


def test_bench_with_several_gpus_spawns_ranks_and_fails_loudly_without_gpus(lbhip):
    """`python bench.py --gpus 2` with no launcher around it (as the driver calls it) starts its own rank processes;
    without GPUs every rank refuses, and the parent reports the failure instead of hanging or printing a line."""
    if lbhip.lb_device_count() > 0:
        pytest.skip("a GPU is visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "rank" in p.stderr and "GPU" in p.stderr
    assert not any(l.startswith("{") for l in p.stdout.splitlines())


def test_bench_config_flag_is_validated_before_anything_touches_a_gpu():
    """--config 2 | 3 | 5 are single-GPU cases: asking for several ranks is refused up front (exit status != 0, no line)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--gpus", "2", "--steps", "4", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "single-GPU" in (p.stderr + p.stdout)
    assert not any(l.startswith("{") for l in p.stdout.splitlines())
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "7"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "invalid choice" in p.stderr


def test_committed_per_configuration_lines_name_their_workload():
    """profiles/r03_bench_c{2,3,4,5}.json: one line per single-GPU configuration of BASELINE.json, each with a roofline object."""
    want = {2: ("1024x1024", "cavity", 72.0), 3: ("4096x4096", "Kelvin-Helmholtz", 72.0), 4: ("8192x8192", "shear layer", 72.0),
            5: ("4096x4096", "porous", 73.0)}
    for c, (grid, word, bpc) in want.items():
        d = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_c%d.json" % c)).read())
        assert d["config"]["baseline_config"] == c and grid in d["config"]["workload"] and word in d["config"]["workload"]
        r = d["roofline"]
        n = d["config"]["grid"][0]
        assert r["bound"] == "hbm" and 0 < r["frac"] <= 1 and r["frac"] == pytest.approx(r["achieved"] / 8000.0, abs=1e-3)
        assert r["algorithmic_bytes_per_launch"] == pytest.approx(bpc * n * n, rel=1e-6)
        assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9, rel=2e-3)
        assert d["health"]["n_nonfinite"] == 0 and d["value"] > 0 and d["unit"] == "MLUPS"
        assert r["traffic"] is None or "committed profile" in r["traffic_source"]


def _committed_bench_lines():
    import glob
    out = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json"))):
        rnd = int(os.path.basename(path)[1:3])
        lines = [l for l in open(path) if l.startswith("{")]
        if lines:
            out.append((rnd, path, json.loads(lines[-1])))
    return out


def test_committed_bench_line_has_the_contract_fields():
    lines = _committed_bench_lines()
    assert lines
    for rnd, path, d in lines:
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                  "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, (path, k)
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0
        assert r["frac"] == pytest.approx(r["achieved"] / 8000.0, abs=1e-3)
        assert "workload" in d["config"] and d["vs_baseline"] is None
        assert r["steps_per_launch"] in (1, 2, 3, 4, 5, 6, 7)
        if rnd >= 2:
            # a fraction of the HBM roofline is a fraction: the bytes a launch must move (72 B x cells, whatever
            # the number of fused time steps) over its duration; the 72 B x UPDATES figure lives under another name
            assert 0 < r["frac"] <= 1.0, path
            n = d["config"]["grid"][0]
            assert r["algorithmic_bytes_per_launch"] == pytest.approx(72.0 * n * n / d["n_gpus"], rel=0.01)
            assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9, rel=2e-3)
            assert r["effective_GBps"] == pytest.approx(r["achieved"] * r["steps_per_launch"], rel=2e-3)
            assert r["frac_of_measured_copy"] == pytest.approx(r["achieved"] / 6290.0, abs=1e-3)
            if r["traffic"] is not None:
                assert "committed profile" in r["traffic_source"]
                assert r["traffic_frac"] == pytest.approx(r["traffic"] / (r["launch_ms"] * 1e-3) / 1e9 / 8000.0, abs=2e-3)
            t = d["timing"]
            assert t["blocks"] >= 5 and t["timed_s"] >= 0.5 and t["statistic"] == "median"
            assert t["min_ms_per_step"] <= d["ms_per_step"] <= t["max_ms_per_step"]
        if rnd >= 4:
            # the other single-GPU configurations of BASELINE.json ride behind the headline, on the same clock
            assert d["methodology"] == ("r3-block-avg" if rnd == 4 else "r5-block-avg-plan-priced")
            oc = {o["config"]: o for o in d["other_configs"] if o["config"] != "reference_case"}
            assert sorted(oc) == [2, 3, 5]
            if rnd >= 5:
                # the reference's one published benchmark (3751 x 1251 Pipe_Flow_Cylinder, 1000 steps), both drop-in classes
                rc = {o["path"]: o for o in d["other_configs"] if o["config"] == "reference_case"}
                assert sorted(rc) == ["cython", "opencl"]
                for o in rc.values():
                    for k in ("workload", "value", "unit", "seconds", "steps", "kernel", "steps_per_launch", "roofline_frac", "reference_published_MLUPS"):
                        assert k in o, (path, k)
                    assert o["steps"] == 1000 and "3751 x 1251" in o["workload"] and o["value"] > 10 * o["reference_published_MLUPS"]
                    assert o["value"] == pytest.approx(3751 * 1251 * 1000 / o["seconds"] / 1e6, rel=2e-3)
                c = d["compute"]
                assert c["bound"] == "valu" and c["frac"] == pytest.approx(d["value"] * 1e6 * 91.0 / 157.3e12, abs=2e-3)
            for c, o in oc.items():
                for k in ("workload", "value", "unit", "ms_per_step", "launch_ms", "roofline_frac", "kernel", "steps_per_launch", "health"):
                    assert k in o, (path, c, k)
                assert o["unit"] == "MLUPS" and o["value"] > 0 and 0 < o["roofline_frac"] <= 1 and o["health"]["n_nonfinite"] == 0
                n = {2: 1024, 3: 4096, 5: 4096}[c]
                # (ms_per_step / launch_ms are rounded in the line: allow for that rounding at 1024^2, where a step is 6 us)
                assert o["value"] == pytest.approx(n * n / (o["ms_per_step"] * 1e-3) / 1e6, rel=max(2e-3, 6e-5 / o["ms_per_step"]))
                assert o["roofline_frac"] == pytest.approx(o["bytes_per_cell_per_launch"] * n * n / (o["launch_ms"] * 1e-3) / 1e9 / 8000.0,
                                                           abs=max(2e-3, 6e-5 / o["launch_ms"]))
            assert "cavity" in oc[2]["workload"] and "Kelvin-Helmholtz" in oc[3]["workload"] and "porous" in oc[5]["workload"]
            own = d["cpu_baseline"]["product_cpu_backend"]
            assert own["kind"] == "own" and own["cores"] == 1 and own["value"] > 0
