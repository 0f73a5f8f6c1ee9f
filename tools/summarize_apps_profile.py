#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile_apps.sh) into profiles/<tag>_summary.json: for every sweep
kernel its average duration (rocprofv3 --kernel-trace --stats) and the HBM bytes per launch from separate
--pmc FETCH_SIZE / --pmc WRITE_SIZE passes (KiB counters; FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"Sweep<(.*?)>, 1>\(", name) or re.search(r"Sweep<(.*)", name)
    return (m.group(1) if m else name)[:120]


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    stats_file = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")))[-1]
    kernels = {}
    for r in csv.DictReader(open(stats_file)):
        if "sweep_kernel" in r["Name"]:
            kernels[r["Name"]] = {"kernel": short(r["Name"]), "calls": int(r["Calls"]),
                                  "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = sorted(glob.glob(os.path.join(src, kind, "*", "*counter_collection.csv")))
        if not files:
            continue
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(files[-1])):
            if r["Counter_Name"] == counter:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for name, v in vals.items():
            if name in kernels:
                kernels[name][counter + "_KiB_raw"] = sum(v) / len(v)
                if "VGPR_Count" not in kernels[name]:
                    pass
    out = []
    for e in kernels.values():
        if "FETCH_SIZE_KiB_raw" in e and "WRITE_SIZE_KiB_raw" in e:
            e["hbm_read_bytes"] = 2 * e["FETCH_SIZE_KiB_raw"] * 1024
            e["hbm_write_bytes"] = e["WRITE_SIZE_KiB_raw"] * 1024
            e["hbm_GBps"] = (e["hbm_read_bytes"] + e["hbm_write_bytes"]) / (e["avg_us"] * 1e-6) / 1e9
        out.append(e)
    out.sort(key=lambda e: -e["calls"] * e["avg_us"])
    bench = [json.loads(l) for l in open(os.path.join(src, "stats.log")) if l.startswith("{")]
    json.dump({"tag": tag, "command": "tools/profile_apps.sh (STSTHIP_VIRTUAL_STRIPS=1, tools/bench_apps.py)",
               "bench_lines_under_the_profiler": bench, "kernels": out},
              open(os.path.join(ROOT, "profiles", f"{tag}_summary.json"), "w"), indent=1)
    for e in out:
        print(f"{e['kernel'][:90]:90s} calls={e['calls']:5d} avg={e['avg_us']:8.1f} us  HBM={e.get('hbm_GBps', float('nan')):7.0f} GB/s")


if __name__ == "__main__":
    main()
