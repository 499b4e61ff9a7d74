#!/usr/bin/env python3
"""Per-wave timeline of one k_step4 / k_deep launch (diagnostic build, LB_DIAG bit 12): when each wave started and ended, on which
XCD / CU / SIMD.  Usage (GPU box): [LB_TIMELINE_BC=pipe] [LB_TIMELINE_DEPTH=6|7] python tools/wave_timeline.py [n] [waves_per_cu]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "2d-lb_amd"), ROOT]
os.environ["LB_LIB"] = os.environ.get("LB_TIMELINE_LIB") or os.path.join(ROOT, "2d-lb_amd", "LB_D2Q9", "liblbhip_diag.so")
os.environ["LB_DIAG"] = os.environ.get("LB_DIAG", "4096")
if len(sys.argv) > 2:
    os.environ["LB_STEP2_WAVES_PER_CU"] = sys.argv[2]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    ny = int(os.environ.get("LB_TIMELINE_NY", n))
    from LB_D2Q9.simulation import Simulation
    from bench import shear_layer
    # (the records travel in the rho array: no launch stores rho, u, v on this handle, and LB_DIAG bit 12 keeps lb_get_macro
    #  from rebuilding them)
    sim = Simulation(n, ny, 1.7, bc=os.environ.get("LB_TIMELINE_BC", "periodic"), inlet_rho=1.0005)
    depth = int(os.environ.get("LB_TIMELINE_DEPTH", "4"))
    deep2 = os.environ.get("LB_TIMELINE_DEEP2") == "1"             # k_deep2<7>: four waves per item (front / back x down / up)
    sim.set_variant({4: 353, 6: 353 | 4096 | 16384, 7: 353 | 4096 | 16384 | 32768}[depth] | (65536 if deep2 else 0))
    sim.init_equilibrium(*shear_layer(n, ny, 0, ny))
    sim.run(2 * depth)
    sim.run(depth)                                # the launch whose timeline is read
    raw = sim.get_fields(("rho",))["rho"]
    u = np.ascontiguousarray(raw.T).view(np.uint32).reshape(-1)          # device order: [y][x]
    # device rows are pitch floats long; host rows nx: with nx % 64 == 0 they coincide
    strips = (n + 255) // 256 if depth == 4 else (n + 239) // 240      # (k_step5, k_deep: 240 apart)
    wpc = int(os.environ.get("LB_STEP2_WAVES_PER_CU", "8" if depth == 4 else "4"))
    cap = 256 * wpc // 2                          # an item = a pair of segments = one workgroup of two waves (up / down)
    segs = max(cap // strips, 1)
    seg_rows = max(-(-ny // segs), 8)
    segs = -(-ny // seg_rows)
    items = 2 * strips * segs
    if deep2:
        return deep2_report(u, strips, segs, n, ny)
    # (boxes with walls: the first / last strip march shorter segments, their extra items follow: read what is there)
    items = min(items + 8 * segs, u.size // 8)
    rec = u[:8 * items].reshape(items, 8).astype(np.int64)
    t0 = rec[:, 0] | (rec[:, 1] << 32)
    t1 = rec[:, 2] | (rec[:, 3] << 32)
    ok = (rec[:, 6] == np.arange(items)) & (t1 > t0)
    print("grid %d x %d: wave records %d, valid %d, pairs per strip %d, rows per pair %d" % (n, ny, items, int(ok.sum()), segs, seg_rows))
    t0, t1, rec = t0[ok], t1[ok], rec[ok]
    base = t0.min()
    start, end = (t0 - base) / 100.0, (t1 - base) / 100.0            # microseconds
    xcc = rec[:, 4] & 15
    print("launch span %.1f us; wave start: median %.1f max %.1f us; wave end: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f us"
          % (end.max(), np.median(start), start.max(), end.min(), np.percentile(end, 10), np.median(end),
             np.percentile(end, 90), end.max()))
    print("mean residency of a wave slot: %.1f %% of the launch" % (100 * (end - start).mean() / end.max()))
    for x in range(8):
        m = xcc == x
        if m.any():
            print("XCD %d: %4d waves, end median %.1f  p90 %.1f  max %.1f us" % (x, m.sum(), np.median(end[m]), np.percentile(end[m], 90), end[m].max()))
    hw = rec[:, 5]
    simd, wave_slot, cu = (hw >> 4) & 3, hw & 15, (hw >> 8) & 15          # HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13]
    for sd in range(4):
        m = simd == sd
        if m.any():
            print("SIMD %d: %4d waves, end median %.1f us; share of upward waves: %.2f" % (sd, m.sum(), np.median(end[m]), (rec[m, 6] % 2).mean()))
    for ws in sorted(set(wave_slot.tolist())):
        m = wave_slot == ws
        print("wave slot %d: %4d waves, end median %.1f us" % (ws, m.sum(), np.median(end[m])))
    it_id = rec[:, 6] >> 1                                            # the pair; bit 0 = direction (0 down, 1 up)
    for d, name in ((0, "down"), (1, "up")):
        m = (rec[:, 6] & 1) == d
        if m.any():
            print("%-4s waves: start median %.1f us, end median %.1f us, duration median %.1f us" % (
                name, np.median(start[m]), np.median(end[m]), np.median((end - start)[m])))
    nseg_main = int(os.environ.get("LB_TIMELINE_NSEGS", segs))       # segments of the interior strips (= segs unless walls)
    sx = np.where(it_id < strips * nseg_main, it_id % strips, np.where((it_id - strips * nseg_main) & 1, strips - 1, 0))
    sy = np.where(it_id < strips * nseg_main, it_id // strips, nseg_main + ((it_id - strips * nseg_main) >> 1))
    order = np.argsort(-end)[:24]
    print("slowest waves (end us, strip, segment, XCD, CU, SIMD, slot):",
          " ".join("(%.0f,%d,%d,%d,%d,%d,%d)" % (end[i], sx[i], sy[i], xcc[i], cu[i], simd[i], wave_slot[i]) for i in order))
    print("end by strip (max us):", " ".join("%.0f" % end[sx == i].max() for i in range(strips)))
    print("end by strip (median us):", " ".join("%.0f" % np.median(end[sx == i]) for i in range(strips)))
    print("end by segment (median us), first 16:", " ".join("%.0f" % np.median(end[sy == i]) for i in range(min(segs, 16))))


def deep2_report(u, strips, segs, n, ny):
    items = 4 * strips * segs
    rec = u[:8 * items].reshape(items, 8).astype(np.int64)
    t0 = rec[:, 0] | (rec[:, 1] << 32)
    t1 = rec[:, 2] | (rec[:, 3] << 32)
    ok = (rec[:, 6] == np.arange(items)) & (t1 > t0)
    print("k_deep2, grid %d x %d: wave records %d, valid %d, pairs per strip %d" % (n, ny, items, int(ok.sum()), segs))
    t0, t1, rec = t0[ok], t1[ok], rec[ok]
    base = t0.min()
    start, end = (t0 - base) / 100.0, (t1 - base) / 100.0
    print("launch span %.1f us; wave start: median %.1f p90 %.1f max %.1f us; wave end: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f us"
          % (end.max(), np.median(start), np.percentile(start, 90), start.max(), end.min(), np.percentile(end, 10), np.median(end),
             np.percentile(end, 90), end.max()))
    print("mean residency of a wave: %.1f %% of the launch" % (100 * (end - start).mean() / end.max()))
    role = rec[:, 6] & 3
    for r, name in enumerate(("front down", "front up", "back down", "back up")):
        m = role == r
        print("%-10s: %4d waves, start median %.1f, end median %.1f p90 %.1f max %.1f us, duration median %.1f us" % (
            name, m.sum(), np.median(start[m]), np.median(end[m]), np.percentile(end[m], 90), end[m].max(), np.median((end - start)[m])))
    hw = rec[:, 5]
    simd, cu, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, rec[:, 4] & 15
    key = ((xcc * 8 + se) * 16 + cu) * 2 + ((hw >> 12) & 1)
    per_cu = {}
    for k, e in zip(key.tolist(), end.tolist()):
        per_cu.setdefault(k, []).append(e)
    counts = np.array([len(v) for v in per_cu.values()])
    print("CUs holding waves: %d; waves per CU: %s" % (len(per_cu), dict(zip(*np.unique(counts, return_counts=True)))))
    for c in sorted(set(counts.tolist())):
        ends = np.array([max(v) for v in per_cu.values() if len(v) == c])
        print("  CUs with %d waves: last wave ends median %.1f max %.1f us" % (c, np.median(ends), ends.max()))
    hist, edges = np.histogram(end, bins=12)
    print("wave ends, histogram:", " ".join("%.0f-%.0f:%d" % (edges[i], edges[i + 1], hist[i]) for i in range(len(hist))))
    hist, edges = np.histogram(start, bins=8)
    print("wave starts, histogram:", " ".join("%.0f-%.0f:%d" % (edges[i], edges[i + 1], hist[i]) for i in range(len(hist))))


if __name__ == "__main__":
    main()
