#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/gpu_profile.sh into profiles/: a kernel-time table
(from --kernel-trace) and HBM traffic per launch (from the two --pmc passes), corrected as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes: on gfx950 FETCH_SIZE reports half
the bytes of a wide coalesced read, so the read side is calibrated on the k_copy4 launches of the
same run, whose byte count is known; WRITE_SIZE is taken as is (checked the same way).

    python tools/pmc_summary.py gpurun_out/prof_<tag> <round-tag> <grid-side> [<BASELINE config, 1-based; default 4>]

Configurations other than 4 are keyed "c<config>/<side>/<steps per launch>" in profiles/pmc_traffic.json (bench.py looks them up
that way); a kernel instantiated with an obstacle mask is priced at 73 B per cell.
"""
import csv
import glob
import json
import os
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows_of(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit("no file matches " + pattern)
    files.sort(key=os.path.getmtime)
    with open(files[-1]) as fh:            # newest, if an older run left files in the same directory
        return list(csv.DictReader(fh))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    src, tag, side = sys.argv[1], sys.argv[2], int(sys.argv[3])
    config = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    kt = rows_of(os.path.join(src, "kt", "**", "*_kernel_trace.csv"))
    # Launches of one marching kernel differ in geometry while lb_autotune samples its candidates (8 or 4 waves
    # per CU): the table keeps them apart by grid size, "name" alone = the geometry with the most launches, i.e.
    # the one the timed region runs.
    by_geo = {}
    for r in kt:
        by_geo.setdefault(short(r["Kernel_Name"]), {}).setdefault(r.get("Grid_Size_X", r.get("Grid_Size", "?")), []).append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    dur = {}
    for k, geos in by_geo.items():
        main = max(geos, key=lambda g: len(geos[g]))
        dur[k] = geos[main]
        for g, v in geos.items():
            if g != main:
                dur["%s [other geometry: grid %s]" % (k, g)] = v
    vg = {short(r["Kernel_Name"]): (r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"]) for r in kt}
    pmc = {}
    for which, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for r in rows_of(os.path.join(src, which, "**", "*_counter_collection.csv")):
            if r["Counter_Name"] == cname:
                pmc.setdefault((short(r["Kernel_Name"]), cname), []).append(float(r["Counter_Value"]))

    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    lines = ["# rocprofv3 summary %s (grid %dx%d)" % (tag, side, side), "",
             "Command: `tools/gpu_profile.sh` = `rocprofv3 --kernel-trace --stats` and two `--pmc` passes "
             "(FETCH_SIZE, WRITE_SIZE) around `bench.py --config %d --steps 20 --warmup 5 --calibrate 5`." % config, "",
             "## Kernel time (--kernel-trace)", "",
             "| kernel | calls | avg us | min us | max us | VGPR | SGPR | scratch | LDS |", "|---|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        lines.append("| %s | %d | %.1f | %.1f | %.1f | %s | %s | %s | %s |" % (
            k, len(v), st.mean(v) / 1e3, min(v) / 1e3, max(v) / 1e3, *vg[k.split(" [")[0]]))

    # the same table as CSV, one row per kernel AND launch geometry (rocprofv3's own --stats file merges the geometries
    # lb_autotune samples into one average: 929 us where the timed region's launches take 870)
    with open(os.path.join(ROOT, "profiles", "%s_kernel_stats_by_geometry.csv" % tag), "w") as fh:
        fh.write("kernel,grid_size_x,calls,avg_us,min_us,max_us,total_us\n")
        for k, geos in sorted(by_geo.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
            for g, v in sorted(geos.items(), key=lambda gv: -len(gv[1])):
                fh.write('"%s",%s,%d,%.1f,%.1f,%.1f,%.1f\n' % (k, g, len(v), st.mean(v) / 1e3, min(v) / 1e3, max(v) / 1e3, sum(v) / 1e3))

    # the bench line of the same (kernel-trace) run: its HIP-event launch time must agree with the table
    try:
        import re
        log = open(os.path.join(src, "kt.log")).read()
        m_l, m_v, m_k = re.search(r'"launch_ms": ([0-9.]+)', log), re.search(r'"value": ([0-9.]+)', log), re.search(r'"kernel": "(k_[a-z]+\d?(?:<\d>)?)', log)
        if m_l and m_v:
            lines += ["", "bench.py inside this run (HIP events over the timed region): %s MLUPS, launch_ms %s of %s."
                      % (m_v.group(1), m_l.group(1), m_k.group(1) if m_k else "the hot kernel")]
    except OSError:
        pass

    # calibration on the copy kernel (bytes known: lattice allocation read once, written once)
    copy_bytes = (9 * (side + 28) * ((side + 63) // 64 * 64) + 1024) * 4      # 14 ghost rows per side (round 5; 10 before), 2 x 512 guard floats
    copy_fetch = st.mean(pmc[("k_copy4<false>", "FETCH_SIZE")]) * 1024
    copy_write = st.mean(pmc[("k_copy4<false>", "WRITE_SIZE")]) * 1024
    fetch_corr = copy_bytes / copy_fetch
    write_corr = copy_bytes / copy_write
    lines += ["", "## HBM traffic (--pmc, per launch)", "",
              "Calibration on `k_copy4<false>` (reads %.0f B, writes the same): FETCH_SIZE x 1024 = %.4g B -> "
              "read correction x%.3f; WRITE_SIZE x 1024 = %.4g B -> write correction x%.3f." % (
                  copy_bytes, copy_fetch, fetch_corr, copy_write, write_corr), "",
              "| kernel | FETCH_SIZE (KiB) | WRITE_SIZE (KiB) | HBM read B (corrected) | HBM write B | total B | compulsory B (72 B x cells; 73 with a mask) | ratio |",
              "|---|---|---|---|---|---|---|---|"]
    out = {}
    for k in sorted(dur):
        if (k, "FETCH_SIZE") not in pmc or not (k.startswith("k_step") or k.startswith("k_deep") or k.startswith("k_tile4")):
            continue
        f = st.mean(pmc[(k, "FETCH_SIZE")])
        w = st.mean(pmc[(k, "WRITE_SIZE")])
        rd, wr = f * 1024 * fetch_corr, w * 1024 * write_corr
        macro = k.split(",")[2].strip() == "true"      # k_step<BC, MASK, MACRO, ...> / k_step2<BC, MASK, MACRO, NTS> / k_tile4<BC, MASK, MACRO, ...>
        masked = k.split(",")[1].strip() == "true"
        # time steps per launch: k_tile4: 4; k_deep<BC, MASK, MACRO, D, RW, PFD>: D; k_stepN: N
        spl = 4 if k.startswith("k_tile4") else (int(k.split(",")[3].strip(" >")) if k.startswith("k_deep") else (int(k[6]) if k[6:7].isdigit() else 1))
        alg = (73.0 if masked else 72.0) * side * side + (12.0 * side * side if macro else 0.0)      # compulsory bytes of one launch, whatever spl
        lines.append("| %s | %.4g | %.4g | %.4g | %.4g | %.4g | %.4g | %.3f |" % (k, f, w, rd, wr, rd + wr, alg, (rd + wr) / alg))
        key = ("%d/%d" % (side, spl)) if config == 4 else ("c%d/%d/%d" % (config, side, spl))
        # (k_step4 and k_tile4 both advance four steps: the key goes to the one the run launched more often -- the other is one
        #  of lb_autotune's samples -- and the other is kept under "<key>:k_tile4" / "<key>:k_step4")
        if not macro and key in out and out[key]["calls_profiled"] >= len(dur[k]):
            key += ":" + k.split("<")[0]
        elif not macro and key in out:
            out[key + ":" + out[key]["kernel"].split("<")[0]] = out[key]
        if not macro:
            # (also under "<key>:<kernel family>": k_deep<7> and k_deep2<7> both advance seven steps and either may be a line's kernel --
            #  bench.py asks for its own first)
            fam_key = ("%d/%d" % (side, spl) if config == 4 else "c%d/%d/%d" % (config, side, spl)) + ":" + k.split("<")[0]
            out[fam_key] = {"calls_profiled": len(dur[k]), "kernel": k, "steps_per_launch": spl, "hbm_bytes_per_launch": round(rd + wr),
                            "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr), "algorithmic_bytes": alg,
                            "fetch_correction": round(fetch_corr, 4), "avg_launch_us_profiled": round(st.mean(dur[k]) / 1e3, 1),
                            "source": "profiles/%s_rocprof_summary.md" % tag}
            out[key] = {"calls_profiled": len(dur[k]),"kernel": k, "steps_per_launch": spl, "hbm_bytes_per_launch": round(rd + wr), "hbm_read_bytes": round(rd),
                   "hbm_write_bytes": round(wr), "algorithmic_bytes": alg, "fetch_correction": round(fetch_corr, 4),
                   "avg_launch_us_profiled": round(st.mean(dur[k]) / 1e3, 1), "source": "profiles/%s_rocprof_summary.md" % tag}
    # rocprofv3's own --stats table of the SAME kernel-trace run, verbatim (never left over from another run)
    stats = glob.glob(os.path.join(src, "kt", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        stats.sort(key=os.path.getmtime)
        import shutil
        shutil.copyfile(stats[-1], os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag))
    with open(os.path.join(ROOT, "profiles", "%s_rocprof_summary.md" % tag), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    jpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    allj = {}
    if os.path.exists(jpath):
        with open(jpath) as fh:
            allj = json.load(fh)
    allj.update(out)
    with open(jpath, "w") as fh:
        json.dump(allj, fh, indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
