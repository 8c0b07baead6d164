#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_bench.sh into small files for profiles/:
<tag>_bench_kernel_stats.csv (the --stats table), <tag>_bench_pmc_hbm.csv (mean FETCH_SIZE /
WRITE_SIZE per dispatch, KB) and traffic.json (HBM bytes per launch of the slam kernels:
2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction for 16-byte-per-lane streaming reads)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    name = name.replace("slam::(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def main(src, dst, tag):
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    rows, mean = [], {}
    for name in ("fetch", "write"):
        files = glob.glob(os.path.join(src, f"pmc_{name}", "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if "slam::" in r["Kernel_Name"]:
                agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            rows.append((k, c, len(v), sum(v) / len(v)))
            mean[(k, c)] = sum(v) / len(v)
    with open(os.path.join(dst, f"{tag}_bench_pmc_hbm.csv"), "w") as f:
        f.write("kernel;counter;dispatches;mean_value_KB_per_dispatch\n")
        for r in sorted(rows):
            f.write("%s;%s;%d;%.3f\n" % r)
    traffic = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py, {tag}; "
                         "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of a "
                         "16-byte-per-lane streaming read)"}
    kernels = sorted({k for k, _ in mean})
    for k in kernels:
        fetch, write = mean.get((k, "FETCH_SIZE")), mean.get((k, "WRITE_SIZE"))
        if fetch is not None and write is not None:
            base = k.split("<")[0]
            traffic[f"{base}_bytes_per_launch"] = (2 * fetch + write) * 1024
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(open(os.path.join(dst, "traffic.json")).read())


if __name__ == "__main__":
    main(*sys.argv[1:4])
