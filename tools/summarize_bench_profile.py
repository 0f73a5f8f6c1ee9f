#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile_bench_r03.sh) into small files:

  <tag>_kernel_stats.csv   rocprofv3 --stats table of `python3 bench.py --gpus 1 --steps 20 --warmup 5`
  <tag>_summary.json       per sweep kernel and launch shape: calls, average duration; for the timed steps the
                           span on the GPU's own clock (first kernel start to last kernel end), the sum of the
                           kernel durations and the time at least one kernel was running (their union), so that
                           bench.py's ms_per_step, the overlap of the two row strips and the per-kernel averages
                           can be reconciled; the bench's own JSON line
  <tag>_counters.json      PMC passes: HBM bytes and VALU wave-instructions per FULL-GRID launch of the kernel
                           bench.py's roofline leg times, keyed by its `kernel_key`

HBM bytes follow MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for wide streaming stores.
Written next to the raw data (gpurun_out/prof_<tag>/); copy them to profiles/ to commit them."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


def bench_line(path):
    try:
        for line in reversed(open(path).read().splitlines()):
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    except OSError:
        pass
    return None


def grid_size(row):
    for key in ("Grid_Size_X", "Grid_Size"):
        if key in row and row[key]:
            return int(row[key])
    return 0


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    bench = bench_line(os.path.join(src, "stats.log"))
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
        launches = []
        for r in csv.DictReader(open(trace_file)):
            if "sweep_kernel" in r["Kernel_Name"]:
                launches.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:170], grid_size(r)))
        launches.sort()
        shapes = collections.defaultdict(list)
        for a, b, name, grid in launches:
            shapes[(name, grid)].append(b - a)
        summary["launch_shapes"] = [
            {"kernel": name, "grid_size": grid, "calls": len(d), "avg_us": sum(d) / len(d) / 1e3,
             "min_us": min(d) / 1e3, "max_us": max(d) / 1e3}
            for (name, grid), d in sorted(shapes.items(), key=lambda kv: -sum(kv[1]))]
        # the timed region: bench runs `warmup` then `steps` identical steps first; everything after them (the
        # verification run, the roofline leg, the general-coefficient leg) comes later in the trace
        if bench:
            steps, warmup = int(bench["steps"]), int(bench["warmup"])
            first = launches[0][2]
            # launches of one step: count up to the first launch of the verification run is unknown, so take it from
            # the kernel-form names: a step starts with the "first launch" form of the uniform kernel
            starts = [i for i, l in enumerate(launches) if "Jacobi5Uniform<true, false>" in l[2] and
                      (i == 0 or "Jacobi5Uniform<true, false>" not in launches[i - 1][2] or
                       launches[i][0] - launches[i - 1][0] > 5_000_000)]
            summary["step_starts_found"] = len(starts)
            per_step = None
            if len(starts) > warmup + steps:
                per_step = []
                for s in range(warmup, warmup + steps):
                    # (not up to the next start: after the last timed step come the verification's launches)
                    seg = launches[starts[s]:starts[s] + int(bench.get("launches_per_step") or (starts[s + 1] - starts[s]))]
                    span = max(b for a, b, *_ in seg) - seg[0][0]
                    busy, union, cur_a, cur_b = sum(b - a for a, b, *_ in seg), 0, None, None
                    for a, b, *_ in seg:
                        if cur_b is None or a > cur_b:
                            if cur_b is not None:
                                union += cur_b - cur_a
                            cur_a, cur_b = a, b
                        else:
                            cur_b = max(cur_b, b)
                    union += cur_b - cur_a
                    per_step.append({"launches": len(seg), "span_ms": span / 1e6, "sum_of_kernel_ms": busy / 1e6,
                                     "union_busy_ms": union / 1e6})
                n = len(per_step)
                summary["timed_steps"] = {
                    "steps": n,
                    "launches_per_step": per_step[0]["launches"],
                    "span_ms_per_step": sum(p["span_ms"] for p in per_step) / n,
                    "sum_of_kernel_ms_per_step": sum(p["sum_of_kernel_ms"] for p in per_step) / n,
                    "union_busy_ms_per_step": sum(p["union_busy_ms"] for p in per_step) / n,
                    "bench_ms_per_step": bench["ms_per_step"],
                    "note": "sum_of_kernel > span: the two row strips of the pass driver run side by side on two "
                            "streams; span ~ bench ms_per_step (host clock, includes launch latency of the first pass)",
                }

    # ---- counters
    key = None
    for log in ("fetch.log", "write.log", "sq.log", "stats.log"):
        line = bench_line(os.path.join(src, log))
        if line and "roofline" in line:
            key = line["roofline"].get("kernel_key")
            break
    per_counter = {}
    by_shape_resources = {}
    for kind in ("fetch", "write", "sq"):
        f = newest(os.path.join(src, kind, "*", "*counter_collection.csv"))
        if not f:
            continue
        vals = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "sweep_kernel" not in r["Kernel_Name"]:
                continue
            shape = (r["Kernel_Name"][:170], grid_size(r))
            vals[r["Counter_Name"]][shape].append(float(r["Counter_Value"]))
            by_shape_resources[shape] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                                               "LDS_Block_Size", "Scratch_Size", "Workgroup_Size",
                                                               "Grid_Size")}
        depth = key.rsplit(":T", 1)[1] if key else ""
        for counter, by_shape in vals.items():
            # full-grid launches of the roofline leg: the middle-launch kernel of the full depth with the largest grid
            cand = [(s, v) for s, v in by_shape.items()
                    if "Jacobi5Uniform<false, false>, false, " + depth + "," in s[0]] or list(by_shape.items())
            shape, v = max(cand, key=lambda sv: sv[0][1])
            per_counter[counter] = {"mean": sum(v) / len(v), "n": len(v), "kernel": shape[0], "grid_size": shape[1]}
    counters = {"tag": tag, "kernels": {}}
    if key and "FETCH_SIZE" in per_counter and "WRITE_SIZE" in per_counter:
        fetch, write = per_counter["FETCH_SIZE"], per_counter["WRITE_SIZE"]
        entry = {
            "kernel": fetch["kernel"], "grid_size": fetch["grid_size"], "launches_counted": fetch["n"],
            "FETCH_SIZE_KiB_raw": fetch["mean"], "WRITE_SIZE_KiB_raw": write["mean"],
            "hbm_read_bytes": 2 * fetch["mean"] * 1024, "hbm_write_bytes": write["mean"] * 1024,
            "hbm_bytes_per_launch": 2 * fetch["mean"] * 1024 + write["mean"] * 1024,
            "resources": by_shape_resources.get((fetch["kernel"], fetch["grid_size"])),
            "source": f"profiles/{tag}_counters.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU in separate "
                      "passes over `python3 bench.py --gpus 1 --steps 2 --warmup 1`; KiB units; FETCH_SIZE doubled "
                      "(gfx950 wide-read correction, MI355X_MICROARCH.md HBM section); full-grid launches only",
        }
        for name, out in (("SQ_INSTS_VALU", "valu_wave_instructions_per_launch"),
                          ("SQ_INSTS_SALU", "salu_wave_instructions_per_launch"), ("SQ_WAVES", "waves_per_launch")):
            if name in per_counter:
                entry[out] = per_counter[name]["mean"]
        counters["kernels"][key] = entry
    counters["raw"] = per_counter
    # ---- the same counters summed over the launches of ONE TIMED STEP (interiors and bands of the row strips):
    # every pass ran `--warmup 1 --steps 2`; a step is `launches_per_step` consecutive sweep launches in dispatch
    # order (bench.py reports the number), the roofline leg's full-grid launches come after the three steps
    counters["timed_step"] = {}
    line = None
    for log in ("sq.log", "fetch.log", "write.log"):
        line = line or bench_line(os.path.join(src, log))
    n_per_step = (line or {}).get("launches_per_step")
    if key and n_per_step:
        sums, shapes_seen, durations = {}, collections.Counter(), []
        for kind in ("fetch", "write", "sq"):
            f = newest(os.path.join(src, kind, "*", "*counter_collection.csv"))
            if not f:
                continue
            rows = [r for r in csv.DictReader(open(f)) if "sweep_kernel" in r["Kernel_Name"]]
            by_counter = collections.defaultdict(list)
            for r in rows:
                by_counter[r["Counter_Name"]].append(r)
            for counter, rs in by_counter.items():
                rs.sort(key=lambda r: int(r["Dispatch_Id"]))
                steps = [rs[i * n_per_step:(i + 1) * n_per_step] for i in (1, 2)]  # the two timed steps
                if any(len(st) != n_per_step for st in steps):
                    continue
                sums[counter] = sum(float(r["Counter_Value"]) for st in steps for r in st) / len(steps)
                if counter in ("FETCH_SIZE", "SQ_INSTS_VALU"):
                    durations.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for st in steps for r in st) / len(steps))
                if counter == "FETCH_SIZE":
                    for r in steps[0]:
                        shapes_seen[grid_size(r)] += 1
        if "FETCH_SIZE" in sums and "WRITE_SIZE" in sums:
            counters["timed_step"][key] = {
                "launches_per_step": n_per_step,
                "launch_shapes": {str(k): v for k, v in sorted(shapes_seen.items())},
                "hbm_bytes_per_step": 2 * sums["FETCH_SIZE"] * 1024 + sums["WRITE_SIZE"] * 1024,
                "hbm_read_bytes_per_step": 2 * sums["FETCH_SIZE"] * 1024,
                "hbm_write_bytes_per_step": sums["WRITE_SIZE"] * 1024,
                "valu_wave_instructions_per_step": sums.get("SQ_INSTS_VALU"),
                "waves_per_step": sums.get("SQ_WAVES"),
                "sum_kernel_ms_per_step": (sum(durations) / len(durations) / 1e6) if durations else None,
                "source": "the same PMC passes: counter values summed over the sweep launches of one timed step "
                          "(launch_shapes: grid size -> launches per step); FETCH_SIZE doubled as above",
            }
    json.dump(summary, open(os.path.join(src, f"{tag}_summary.json"), "w"), indent=1)
    json.dump(counters, open(os.path.join(src, f"{tag}_counters.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "bench_line"}, indent=1)[:2500])
    print(json.dumps(counters["kernels"], indent=1)[:2000])
    print(json.dumps(counters["timed_step"], indent=1)[:2000])


if __name__ == "__main__":
    main()
