#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 --kernel-trace --stats, and separate --pmc FETCH_SIZE /
--pmc WRITE_SIZE passes of the same bench command) into tracked files under profiles/.

HBM bytes follow MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for
16-B-per-lane streaming stores."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:] if files else []


def main():
    tag = sys.argv[1]
    kernel_key = sys.argv[2] if len(sys.argv) > 2 else "sweep_kernel"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)

    stats = list(csv.DictReader(open(newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))[0])))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in stats:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
    sweep = [r for r in stats if kernel_key in r["Name"]]

    counters = {}
    resources = {}
    for kind, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = newest(os.path.join(src, kind, "*", "*counter_collection.csv"))
        if not files:
            continue
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == name:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
                if kernel_key in r["Kernel_Name"]:
                    resources[r["Kernel_Name"]] = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                                                     "LDS_Block_Size", "Scratch_Size", "Workgroup_Size",
                                                                     "Grid_Size")}
        for k, v in vals.items():
            if kernel_key in k:
                counters.setdefault(k, {})[name] = (sum(v) / len(v), len(v))

    summary = {"tag": tag, "kernels": []}
    for r in sweep:
        entry = {"name": r["Name"][:200], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                 "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                 "share_of_gpu_time_pct": float(r["Percentage"])}
        for k, c in counters.items():
            if k[:100] == r["Name"][:100]:
                fetch_kib = c.get("FETCH_SIZE", (None, 0))[0]
                write_kib = c.get("WRITE_SIZE", (None, 0))[0]
                entry["FETCH_SIZE_KiB_raw"] = fetch_kib
                entry["WRITE_SIZE_KiB_raw"] = write_kib
                if fetch_kib is not None and write_kib is not None:
                    entry["hbm_read_bytes"] = 2 * fetch_kib * 1024  # gfx950 correction
                    entry["hbm_write_bytes"] = write_kib * 1024
                    entry["hbm_bytes_per_launch"] = entry["hbm_read_bytes"] + entry["hbm_write_bytes"]
                    entry["hbm_GBps"] = entry["hbm_bytes_per_launch"] / (entry["avg_us"] * 1e-6) / 1e9
                entry["resources"] = resources.get(k)
        summary["kernels"].append(entry)
    json.dump(summary, open(os.path.join(out_dir, f"{tag}_summary.json"), "w"), indent=1)
    main_k = max(summary["kernels"], key=lambda e: e["calls"] * e["avg_us"]) if summary["kernels"] else None
    if main_k and "hbm_bytes_per_launch" in main_k:
        json.dump({"kernel": main_k["name"], "hbm_bytes_per_launch": main_k["hbm_bytes_per_launch"],
                   "avg_us": main_k["avg_us"], "source": f"profiles/{tag}_summary.json",
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB units; "
                             "FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section)"},
                  open(os.path.join(out_dir, f"traffic_{tag}.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1)[:3000])


if __name__ == "__main__":
    main()
