#!/usr/bin/env python3
"""Benchmark harness for the example binaries: the counterpart of the reference's Julia scripts
(scripts/benchmark-common.jl, examples/{jacobi,hotspot,fdtd}/scripts/benchmark.jl), for MI355X.

Same procedure and same definitions as the reference:
  * the binaries are the reference's unchanged example sources built against this backend (examples/Makefile);
  * a run's runtime is the application's own "Walltime: <s> s" line, minimum over n samples after one warm-up
    run (jacobi/scripts/benchmark.jl:52-94);
  * workload = rows * cols * iterations, measured throughput = workload / runtime (sub-iterations are not
    counted), FLOPS = throughput * operations per cell (benchmark-common.jl:97-98,122-123);
  * `max_perf` writes metrics.<name>.json with the keys the reference's CUDA variant writes
    (target, measured, FLOPS); `scaling` writes scaling.<name>.csv with the reference's columns
    (grid_wh, n_iters, runtime, measured_throughput, model_throughput), grid sizes going down by sqrt(2) from
    the largest grid three buffers of which fit the device (benchmark-common.jl:196-209).
What differs is the device model (benchmark-common.jl:148-162 hard-codes an A100 and one pass per sub-iteration):
    memory bound : 0.8 * 8 TB/s / (2 * cell size) cells per second and sub-iteration, divided by the generations
                   one launch covers (temporal blocking: a cell moves through HBM once per launch)
    launch bound : one launch per T generations at LAUNCH_LATENCY each
    model runtime = max(memory time, launch time)
usage: tools/benchmark.py {max_perf|scaling} {jacobi|hotspot|fdtd} [--exe PATH] [--target-runtime S] [--out DIR]
"""
import argparse
import csv
import json
import math
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLES = os.path.join(ROOT, "build", "examples")
WALLTIME = re.compile(r"Walltime: ([0-9]+\.[0-9]+(?:e[-+]?[0-9]+)?) s")

HBM_BYTES_PER_S = 0.8 * 8.0e12   # the reference's model takes 80 % of the device's peak (benchmark-common.jl:149)
HBM_CAPACITY = 288 * 2 ** 30     # MI355X
LAUNCH_LATENCY = 10e-6           # seconds per kernel launch (the reference's figure per pass, benchmark-common.jl:158)

# cell size in bytes, sub-iterations, operations per cell (the reference's constants: jacobi/scripts/benchmark.jl:15,
# kernels.hpp n_operations; hotspot/scripts/benchmark.jl:13-14; fdtd/scripts/benchmark.jl:12-13), generations per
# launch of this backend (SweepTuning of the functor the unchanged source instantiates)
APPS = {
    "jacobi": {"cell_size": 4, "n_subiters": 1, "ops": None, "generations_per_launch": 8,
               "exe": "jacobi_Jacobi5General_hip"},
    "hotspot": {"cell_size": 8, "n_subiters": 1, "ops": 15, "generations_per_launch": 8, "exe": "hotspot_hip"},
    "fdtd": {"cell_size": 32, "n_subiters": 2, "ops": 24, "generations_per_launch": 8, "exe": "fdtd_hip"},
}


def max_grid_wh(cell_size, clip_to_base=None):
    """Largest square grid with three buffers in device memory and 32-bit indexable (benchmark-common.jl:196-209)."""
    max_cells = min(HBM_CAPACITY / 3 / cell_size, 2 ** 31)
    wh = math.sqrt(max_cells)
    if clip_to_base is not None:
        wh = clip_to_base ** math.floor(math.log(wh, clip_to_base) + 1e-9)
    return int(math.floor(wh))


def model_runtime(app, grid_wh, n_iters):
    a = APPS[app]
    cells = float(grid_wh) * grid_wh
    launches = math.ceil(n_iters / a["generations_per_launch"])
    memory = launches * cells * 2 * a["cell_size"] / HBM_BYTES_PER_S
    return max(memory, launches * LAUNCH_LATENCY)


def sample_walltimes(command, n_samples, warmup=True):
    times = []
    for i in range(n_samples + (1 if warmup else 0)):
        out = subprocess.run(command, capture_output=True, text=True, check=True).stdout
        m = WALLTIME.search(out)
        if not m:
            raise SystemExit(f"no 'Walltime:' line in the output of {' '.join(command)}:\n{out[-400:]}")
        if i > 0 or not warmup:
            times.append(float(m.group(1)))
    return min(times)


def jacobi_config(exe):
    """`<exe> show-config` prints the kernel's constants as JSON (jacobi.cpp:90-102)."""
    text = subprocess.run([exe, "show-config"], capture_output=True, text=True, check=True).stdout
    return json.loads(re.search(r"(\{[^{}]+\})\s*$", text).group(1))


def hotspot_inputs(n, directory):
    """examples/hotspot/data/input_gen.jl:3-15: temp = 30, power = 0.5 inside the centred square (1-based n/4..3n/4)."""
    import numpy as np

    temp, power = os.path.join(directory, "temp.bin"), os.path.join(directory, "power.bin")
    np.full((n, n), 30.0, dtype=np.float32).tofile(temp)
    p = np.zeros((n, n), dtype=np.float32)
    p[n // 4 - 1:3 * n // 4, n // 4 - 1:3 * n // 4] = 0.5
    p.tofile(power)
    return temp, power


C0 = 299792458.0


def fdtd_experiment(grid_wh, n_iters, directory):
    """examples/fdtd/scripts/benchmark.jl:138-143: scale dx to the grid size and t_max to the step count."""
    e = json.load(open(os.path.join(ROOT, "tools", "data", "fdtd_max_grid_nosnap.json")))
    e["dx"] = 2.0 * e["cavity_rings"][0]["radius"] / (grid_wh - 2)
    dt = e["dx"] / (C0 * math.sqrt(2.0)) * 0.99
    e["time"]["t_max"] = n_iters * dt / e["tau"]
    path = os.path.join(directory, "experiment.json")
    json.dump(e, open(path, "w"))
    return path


def run_benchmark(app, exe, grid_wh, n_iters, n_samples, workdir, warmup=True):
    if app == "jacobi":
        n_coef = jacobi_config(exe)["n_coefficients"]
        command = [exe, str(grid_wh), str(grid_wh), str(n_iters), "/dev/null"] + [str(1.0 / n_coef)] * n_coef
    elif app == "hotspot":
        temp, power = hotspot_inputs(grid_wh, workdir)
        command = [exe, str(grid_wh), str(grid_wh), str(n_iters), temp, power, "/dev/null"]
    else:
        command = [exe, "-c", fdtd_experiment(grid_wh, n_iters, workdir), "-o", workdir]
    return sample_walltimes(command, n_samples, warmup)


def iterations_for(app, grid_wh, target_runtime):
    per_launch = APPS[app]["generations_per_launch"]
    n_passes = max(1, math.ceil(target_runtime / model_runtime(app, grid_wh, per_launch)))
    return n_passes * per_launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["max_perf", "scaling"])
    ap.add_argument("app", choices=sorted(APPS))
    ap.add_argument("--exe")
    ap.add_argument("--target-runtime", type=float, default=2.0, help="seconds per run by the model (reference: 30)")
    ap.add_argument("--samples", type=int, default=3)
    ap.add_argument("--grid", type=int, default=0, help="max_perf: grid size (default: the BASELINE configuration)")
    ap.add_argument("--min-grid", type=int, default=32)
    ap.add_argument("--max-grid", type=int, default=0, help="scaling: start below the largest grid that fits")
    ap.add_argument("--out", default=".")
    args = ap.parse_args()
    a = APPS[args.app]
    exe = args.exe or os.path.join(EXAMPLES, a["exe"])
    name = os.path.basename(exe)
    ops = a["ops"] if a["ops"] is not None else jacobi_config(exe)["n_operations"]
    os.makedirs(args.out, exist_ok=True)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None

    with tempfile.TemporaryDirectory(dir=shm) as work:
        if args.mode == "max_perf":
            grid = args.grid or {"jacobi": 16384, "hotspot": 8192, "fdtd": 4608}[args.app]
            n_iters = iterations_for(args.app, grid, args.target_runtime)
            runtime = run_benchmark(args.app, exe, grid, n_iters, args.samples, work)
            measured = float(grid) * grid * n_iters / runtime
            metrics = {"target": f"{name}, MI355X (stencil::hip)", "measured": measured, "FLOPS": measured * ops,
                       "grid_wh": grid, "n_iters": n_iters, "runtime": runtime,
                       "model": float(grid) * grid * n_iters / model_runtime(args.app, grid, n_iters)}
            json.dump(metrics, open(os.path.join(args.out, f"metrics.{name}.json"), "w"), indent=1)
            print(json.dumps(metrics))
        else:
            path = os.path.join(args.out, f"scaling.{name}.csv")
            wh = float(max_grid_wh(a["cell_size"], clip_to_base=math.sqrt(2.0)))
            first = True
            with open(path, "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["grid_wh", "n_iters", "runtime", "measured_throughput", "model_throughput"])
                while round(wh) >= args.min_grid:
                    grid = int(round(wh))
                    wh /= math.sqrt(2.0)
                    if args.max_grid and grid > args.max_grid:
                        continue
                    n_iters = iterations_for(args.app, grid, args.target_runtime)
                    runtime = run_benchmark(args.app, exe, grid, n_iters, args.samples, work, warmup=first)
                    workload = float(grid) * grid * n_iters
                    w.writerow([grid, n_iters, runtime, workload / runtime, workload / model_runtime(args.app, grid, n_iters)])
                    f.flush()
                    print(grid, n_iters, runtime, f"{workload / runtime / 1e9:.1f} Gcell/s", flush=True)
                    first = False


if __name__ == "__main__":
    sys.exit(main())
