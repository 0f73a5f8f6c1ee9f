#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile_bench_r04.sh) into the small files that are committed under profiles/:

  <tag>_kernel_stats.csv   rocprofv3 --stats table of the command the driver times
                           (`python3 bench.py --gpus 1 --steps 20 --warmup 5`)
  <tag>_summary.json       that command's sweep launches per kernel and launch shape (calls, average duration), and the
                           bench line it printed
  <tag>_counters.json      `python3 bench.py --profile-legs` (every kernel leg of the bench line three times): per leg
                           and call the HBM bytes, the VALU / SALU / LDS wave-instructions, the waves, the kernel time
                           and the wave-cycle breakdown, and per launch shape of a leg the same per launch with the
                           kernel's resources.  bench.py reads this file (COUNTER_FILE) and divides by ITS OWN times.

How the launches are told apart: bench.py --profile-legs prints the sequence of legs with the calls and the launches per
call of each; the sweep launches of a pass, in dispatch order, are cut into legs by those counts -- no kernel names are
matched (they are recorded per leg, so a wrong cut would show).

HBM bytes follow MI355X_MICROARCH.md, "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide (16 bytes per lane) coalesced streaming read and WRITE_SIZE is exact for 16-byte-per-lane streaming
stores; "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern" -- the
calibration copies of tools/microbench/fetch_calibration.hip (4, 8, 16 bytes per lane, 1 GiB each way) run under the same
two passes give the factors per width, and every leg is corrected with the factor of ITS access width (`bytes_per_lane`
below)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# bytes per lane and access of the loads and stores of each leg's kernels (cells per lane x element size of a plane)
BYTES_PER_LANE = {"headline": 16, "headline_full_grid": 16, "general_coefficients": 16, "general_coefficients_fma": 16,
                  "hotspot_8192": 8, "fdtd_max_grid": 16}


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


def json_line(path, key):
    try:
        for line in reversed(open(path).read().splitlines()):
            if line.startswith("{") and f'"{key}"' in line:
                return json.loads(line)
    except OSError:
        pass
    return None


def grid_size(row):
    for key in ("Grid_Size_X", "Grid_Size"):
        if key in row and row[key]:
            return int(row[key])
    return 0


def short(name):
    name = name.replace("void stencil::hip::internal::sweep_kernel<stencil::hip::internal::Sweep<", "")
    name = name.replace("stencil::apps::", "").replace("(anonymous namespace)::", "")
    return name[:110]


def dispatches(path, name_key="Kernel_Name"):
    """sweep launches of one pass in dispatch order: [{id, kernel, grid, ns, counters{}, resources{}}]"""
    by_id = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "sweep_kernel" not in r[name_key]:
            continue
        # (legs are separated by blocking calls, so the start time orders them as well as the dispatch id does)
        ident = r.get("Dispatch_Id") or r["Start_Timestamp"]
        d = by_id.get(ident)
        if d is None:
            d = by_id[ident] = {
                "id": int(ident), "kernel": short(r[name_key]), "grid": grid_size(r),
                "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "counters": {},
                "resources": {"vgprs": r.get("VGPR_Count"), "accum_vgprs": r.get("Accum_VGPR_Count"),
                              "sgprs": r.get("SGPR_Count"), "lds_bytes": r.get("LDS_Block_Size"),
                              "scratch_bytes": r.get("Scratch_Size"), "workgroup": r.get("Workgroup_Size")}}
        if "Counter_Name" in r:
            d["counters"][r["Counter_Name"]] = d["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return sorted(by_id.values(), key=lambda d: d["id"])


def cut(launches, sequence):
    """{leg: [calls][launches]} by the printed sequence; None if the counts do not add up"""
    need = sum(sum(s["launches"]) for s in sequence)
    if need != len(launches):
        return None, f"{len(launches)} sweep launches in the pass, the sequence accounts for {need}"
    out, at = {}, 0
    for s in sequence:
        calls = []
        for n in s["launches"]:
            if n == s["launches_per_call"]:  # (a probing first call made other launches: not averaged)
                calls.append(launches[at:at + n])
            at += n
        out[s["leg"]] = calls
    return out, None


def calibration(src):
    """true bytes / reported bytes per access width, from the calibration copies"""
    factors = {"note": "true bytes / (counter x 1024) of tools/microbench/fetch_calibration.hip's 1 GiB copies, "
                       "nontemporal and plain stores; the guide's numbers: FETCH_SIZE 2.0 at 16 B per lane, WRITE_SIZE 1.0"}
    for kind, counter, key in (("cal_fetch", "FETCH_SIZE", "read"), ("cal_write", "WRITE_SIZE", "write")):
        f = newest(os.path.join(src, kind, "*", "*counter_collection.csv"))
        line = json_line(os.path.join(src, kind + ".log"), "calibration_bytes_read_per_launch")
        if not f or not line:
            continue
        true_bytes = line["calibration_bytes_read_per_launch"]
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "calibration_copy" in r["Kernel_Name"]:
                import re

                m = re.search(r"calibration_copy_b(\d+)_(nontemporal|plain)", r["Kernel_Name"])
                if not m:
                    continue
                w, flavour = int(m.group(1)), m.group(2)
                vals[(w, flavour)].append(float(r["Counter_Value"]))
        factors[key] = {f"{w}B_{fl}": true_bytes / (sum(v) / len(v) * 1024) for (w, fl), v in sorted(vals.items()) if sum(v) > 0}
    return factors


def factor(factors, key, width, default):
    for flavour in ("nontemporal", "plain"):
        v = (factors.get(key) or {}).get(f"{width}B_{flavour}")
        if v:
            return v
    return default


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")

    # ---- 1. the driver's command
    bench = json_line(os.path.join(src, "stats.log"), "metric")
    summary = {"tag": tag, "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5",
               "bench_line": bench}
    stats_file = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
    if stats_file:
        rows = list(csv.DictReader(open(stats_file)))
        with open(os.path.join(src, f"{tag}_kernel_stats.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([r["Name"][:200], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
    trace_file = newest(os.path.join(src, "stats", "*", "*kernel_trace.csv"))
    if trace_file:
        shapes = collections.defaultdict(list)
        for r in csv.DictReader(open(trace_file)):
            if "sweep_kernel" in r["Kernel_Name"]:
                shapes[(short(r["Kernel_Name"]), grid_size(r))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        summary["launch_shapes"] = [
            {"kernel": name, "grid_size": grid, "calls": len(d), "avg_us": sum(d) / len(d) / 1e3, "min_us": min(d) / 1e3,
             "max_us": max(d) / 1e3} for (name, grid), d in sorted(shapes.items(), key=lambda kv: -sum(kv[1]))]

    # ---- 2. the legs
    factors = calibration(src)
    counters = {"tag": tag, "program": "python3 bench.py --profile-legs (every kernel leg: one warm-up call and two more)",
                "calibration": factors, "legs": {}, "problems": []}
    per_leg = collections.defaultdict(dict)   # leg -> counter -> per-call mean
    shapes_of = collections.defaultdict(dict)  # leg -> (kernel, grid) -> dict
    for kind in ("legs_trace", "fetch", "write", "sq", "sqwait"):
        seq_line = json_line(os.path.join(src, kind + ".log"), "profile_legs")
        f = newest(os.path.join(src, kind, "*", "*kernel_trace.csv" if kind == "legs_trace" else "*counter_collection.csv"))
        if not seq_line or not f:
            counters["problems"].append(f"pass {kind}: no output")
            continue
        sequence = seq_line["profile_legs"]
        legs, why = cut(dispatches(f), sequence)
        if legs is None:
            counters["problems"].append(f"pass {kind}: {why}")
            continue
        for s in sequence:
            name, calls = s["leg"], legs[s["leg"]]
            entry = counters["legs"].setdefault(name, {
                "calls": len(calls), "launches_per_call": s["launches_per_call"], "cells_per_call": s["cells_per_call"],
                "bytes_per_cell_update_algorithmic": s["bytes_per_cell_update"], "bytes_per_lane": BYTES_PER_LANE.get(name, 16)})
            entry.setdefault("s_per_call_host", {})[kind] = s["s_per_call_min"]
            sums = collections.defaultdict(float)
            for call in calls:
                for d in call:
                    sums["_ns"] += d["ns"]
                    for c, v in d["counters"].items():
                        sums[c] += v
                    sh = shapes_of[name].setdefault((d["kernel"], d["grid"]), {"n": collections.Counter(), "ns": collections.defaultdict(float),
                                                                              "counters": collections.defaultdict(float), "resources": d["resources"]})
                    sh["n"][kind] += 1
                    sh["ns"][kind] += d["ns"]
                    for c, v in d["counters"].items():
                        sh["counters"][c] += v
            for c, v in sums.items():
                per_leg[name][(kind, c)] = v / len(calls)
    for name, entry in counters["legs"].items():
        width = entry["bytes_per_lane"]
        fr, fw = factor(factors, "read", width, 2.0 if width == 16 else 1.0), factor(factors, "write", width, 1.0)
        entry["fetch_size_factor"], entry["write_size_factor"] = fr, fw
        get = lambda kind, c: per_leg[name].get((kind, c))  # noqa: E731
        if get("fetch", "FETCH_SIZE") is not None and get("write", "WRITE_SIZE") is not None:
            entry["FETCH_SIZE_KiB_raw_per_call"], entry["WRITE_SIZE_KiB_raw_per_call"] = get("fetch", "FETCH_SIZE"), get("write", "WRITE_SIZE")
            entry["hbm_read_bytes_per_call"] = fr * get("fetch", "FETCH_SIZE") * 1024
            entry["hbm_write_bytes_per_call"] = fw * get("write", "WRITE_SIZE") * 1024
            entry["hbm_bytes_per_call"] = entry["hbm_read_bytes_per_call"] + entry["hbm_write_bytes_per_call"]
            # compulsory traffic: every cell read once and written once per launch (constant planes: read only)
        for c, out in (("SQ_INSTS_VALU", "valu_per_call"), ("SQ_INSTS_SALU", "salu_per_call"), ("SQ_WAVES", "waves_per_call"),
                       ("SQ_INSTS_LDS", "lds_instructions_per_call")):
            if get("sq", c) is not None:
                entry[out] = get("sq", c)
        if get("legs_trace", "_ns") is not None:
            entry["sum_kernel_ms_per_call"] = get("legs_trace", "_ns") / 1e6
            entry["avg_us_per_launch"] = get("legs_trace", "_ns") / 1e3 / entry["launches_per_call"]
        if get("sq", "_ns") is not None:
            entry["sum_kernel_ms_per_call_under_counters"] = get("sq", "_ns") / 1e6
        wc = get("sqwait", "SQ_WAVE_CYCLES")
        if wc:
            entry["sq"] = {
                "SQ_WAVE_CYCLES": wc, "SQ_BUSY_CYCLES": get("sqwait", "SQ_BUSY_CYCLES"),
                "frac_parked_at_waitcnt_or_barrier": get("sqwait", "SQ_WAIT_ANY") / wc,
                "frac_issue_stalled": get("sqwait", "SQ_WAIT_INST_ANY") / wc,
                "frac_issuing": get("sqwait", "SQ_ACTIVE_INST_ANY") / wc,
                "frac_issuing_valu": (get("sqwait", "SQ_ACTIVE_INST_VALU") or 0) / wc,
                "note": "of the waves' resident cycles (SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES, "
                        "disjoint: MI355X_MICROARCH.md, SQ counters)"}
        shapes = []
        for (kernel, grid), sh in sorted(shapes_of[name].items(), key=lambda kv: -kv[1]["ns"].get("legs_trace", 0)):
            per = lambda kind, c: (sh["counters"].get(c, 0.0) / sh["n"][kind]) if sh["n"].get(kind) else None  # noqa: E731
            one = {"kernel": kernel, "grid_size": grid,
                   "launches_per_call": (sh["n"].get("legs_trace") or sh["n"].get("fetch") or 0) / entry["calls"],
                   "avg_us": sh["ns"]["legs_trace"] / sh["n"]["legs_trace"] / 1e3 if sh["n"].get("legs_trace") else None,
                   "avg_us_under_counters": sh["ns"]["sq"] / sh["n"]["sq"] / 1e3 if sh["n"].get("sq") else None,
                   "vgprs": sh["resources"].get("vgprs"), "lds_bytes": sh["resources"].get("lds_bytes"),
                   "scratch_bytes": sh["resources"].get("scratch_bytes"), "workgroup": sh["resources"].get("workgroup")}
            if per("fetch", "FETCH_SIZE") is not None and per("write", "WRITE_SIZE") is not None:
                one["hbm_bytes_per_launch"] = fr * per("fetch", "FETCH_SIZE") * 1024 + fw * per("write", "WRITE_SIZE") * 1024
            if per("sq", "SQ_INSTS_VALU") is not None:
                one["valu_per_launch"] = per("sq", "SQ_INSTS_VALU")
                one["waves_per_launch"] = per("sq", "SQ_WAVES")
            shapes.append(one)
        entry["shapes"] = shapes
        entry["kernels_seen"] = sorted({k for k, _ in shapes_of[name]})
        entry["source"] = (f"profiles/{tag}_counters.json: rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_INSTS_*; SQ wave "
                           "cycles) over `python3 bench.py --profile-legs`, sums over ALL launches of one call of the leg; KiB "
                           f"units; FETCH_SIZE x {fr:.3f}, WRITE_SIZE x {fw:.3f} (calibration copies at {width} B per lane)")
    json.dump(summary, open(os.path.join(src, f"{tag}_summary.json"), "w"), indent=1)
    json.dump(counters, open(os.path.join(src, f"{tag}_counters.json"), "w"), indent=1)
    brief = {n: {k: e.get(k) for k in ("launches_per_call", "hbm_bytes_per_call", "valu_per_call", "sum_kernel_ms_per_call",
                                       "fetch_size_factor", "write_size_factor")} | {"sq": {k: round(v, 3) for k, v in (e.get("sq") or {}).items() if k.startswith("frac")}}
             for n, e in counters["legs"].items()}
    print(json.dumps({"calibration": factors, "problems": counters["problems"], "legs": brief}, indent=1)[:6000])


if __name__ == "__main__":
    main()
