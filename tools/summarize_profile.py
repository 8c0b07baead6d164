#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_bench.sh into small files for profiles/:
<tag>_bench_kernel_stats.csv (the --stats table), <tag>_bench_pmc_hbm.csv (mean FETCH_SIZE /
WRITE_SIZE per dispatch, KB), <tag>_bench_pmc_issue.txt (mean SQ counters per dispatch) and
traffic.json (HBM bytes per launch of the slam kernels: 2*FETCH_SIZE + WRITE_SIZE, the gfx950
correction for 16-byte-per-lane streaming reads).

Dispatches are grouped by kernel name WITH its template arguments and by grid size, so that the
launches of one workload are never averaged with another's (the 10 000-hypothesis scoring launch
of the headline step against the 128-hypothesis slices of the device RANSAC, k = 1 against k = 2
matching)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    """Kernel name with template arguments, without namespaces, return type and parameter list."""
    name = name.replace("slam::(anonymous namespace)::", "").replace("void ", "")
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def grid_of(row):
    for key in ("Grid_Size", "Grid_Size_X", "grid_size"):
        if key in row and row[key] not in ("", None):
            return str(row[key])
    return "?"


def counters(src, sub):
    """{(kernel, grid, counter): [values]} of every slam kernel in the counter CSVs under src/sub*."""
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if "slam::" in r["Kernel_Name"]:
                agg[(short(r["Kernel_Name"]), grid_of(r), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def main(src, dst, tag):
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    # per (kernel, grid) durations from the kernel trace: the --stats table merges grids
    traces = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
    if traces:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(traces[0])):
            if "slam::" in r["Kernel_Name"]:
                dur[(short(r["Kernel_Name"]), grid_of(r))].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        with open(os.path.join(dst, f"{tag}_bench_kernel_by_grid.csv"), "w") as f:
            f.write("kernel;grid;dispatches;mean_ns;min_ns;max_ns\n")
            for (k, g), v in sorted(dur.items()):
                f.write("%s;%s;%d;%.0f;%.0f;%.0f\n" % (k, g, len(v), sum(v) / len(v), min(v), max(v)))
    rows, mean, count = [], {}, {}
    for name in ("fetch", "write"):
        for (k, g, c), v in counters(src, f"pmc_{name}").items():
            rows.append((k, g, c, len(v), sum(v) / len(v)))
            mean[(k, g, c)] = sum(v) / len(v)
            count[(k, g, c)] = len(v)
    with open(os.path.join(dst, f"{tag}_bench_pmc_hbm.csv"), "w") as f:
        f.write("kernel;grid;counter;dispatches;mean_value_KB_per_dispatch\n")
        for r in sorted(rows):
            f.write("%s;%s;%s;%d;%.3f\n" % r)
    issue = {}
    for sub in ("pmc_issue1", "pmc_issue2"):
        issue.update(counters(src, sub))
    if issue:
        with open(os.path.join(dst, f"{tag}_bench_pmc_issue.txt"), "w") as f:
            f.write("# rocprofv3 --pmc (two passes of 8 SQ counters each) of bench.py, headline step only "
                    "(SLAM_BENCH_RANSAC=0 SLAM_BENCH_INGEST=0 SLAM_BENCH_FULLSEQ=0), one batch in flight:\n"
                    "# mean counter value per dispatch, by kernel template and grid\n"
                    "# kernel;grid;counter;dispatches;mean\n")
            for (k, g, c), v in sorted(issue.items()):
                f.write("%s;%s;%s;%d;%.4g\n" % (k, g, c, len(v), sum(v) / len(v)))
    traffic = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py, {tag}; "
                         "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of a "
                         "16-byte-per-lane streaming read); mean over all launches of a kernel (its templates and grids: the "
                         "profiled run is the headline step only)"}
    total, calls = collections.defaultdict(float), collections.defaultdict(int)
    for (k, g, c) in mean:
        if c != "FETCH_SIZE" or (k, g, "WRITE_SIZE") not in mean:
            continue
        n = count[(k, g, c)]
        total[k.split("<")[0]] += n * (2 * mean[(k, g, c)] + mean[(k, g, "WRITE_SIZE")]) * 1024
        calls[k.split("<")[0]] += n
    for base in sorted(total):
        traffic[f"{base}_bytes_per_launch"] = total[base] / calls[base]
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(open(os.path.join(dst, "traffic.json")).read())


if __name__ == "__main__":
    main(*sys.argv[1:4])
