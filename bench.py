#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json configs[1] -- Jacobi 5-point (Jacobi5General) fp32, 16384 x 16384,
1000 generations, grid resident in HBM.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver launches it with
torch.distributed.run, one rank per GPU; run plainly (`python bench.py --gpus N`, no RANK in the environment) it
starts that launcher itself as a CHILD process -- before anything has touched the GPU -- and exits with its code.
A "step" is ONE StencilUpdate call of `--generations` generations (default 1000, BASELINE's count: 62 launches of 16 + one of 8).  Rank 0 prints
ONE JSON line.

Metric: Gcell-updates/s = H*W*generations / time (sub-iterations not counted), the reference's definition
(scripts/benchmark-common.jl:97-98,122).

N > 1 is BASELINE's series (metric "16384^2, 1/2/4/8 GPU"): STRONG scaling -- the
same 16384^2 grid cut into N row strips, ghost rows exchanged with the two neighbours over RCCL by the native
strip driver of libststhip.so (no collective on the data path).  `--config5` runs the 65536^2 grid of configs[4]
instead (also strong: 65536/N rows per rank; its 1-GPU point fits one MI355X).  `--rows-per-gpu R` fixes the
rows per rank instead (weak scaling).  If the RCCL communicator cannot be created the run FAILS (exit code 3)
unless `--allow-fallback` asks for the torch.distributed driver instead.

What the JSON line carries besides the contract's fields:
  roofline      the dominant kernel measured live with HIP events on the launch stream (full-grid launches of
                the deepest compiled blocking depth).  `achieved` / `frac` are SURVEY 8d's ALGORITHMIC bytes
                (8 B per cell-update x generations per launch) over the launch time -- with temporal blocking
                this exceeds the 8 TB/s peak by design (each cell moves once per T generations) --, `traffic` the
                HBM bytes of such a launch from the PMC passes.  The PHYSICAL fractions lead with `timed_path`:
                HBM bytes and VALU wave-instructions of ONE TIMED STEP (all its launches: two row strips with a
                moving boundary, side by side) from the committed PMC passes of this file's own legs
                (profiles/r04_bench_counters.json, tools/profile_bench_r04.sh) over THIS run's ms_per_step;
                `full_grid_launch` is the same per launch over kernel_ms.  VALU fractions on both bases (the guide's
                2 cycles at 2.4 GHz, this repository's measured 1.09 ns); `bound` is derived from the fractions.
  verified      the result of the LAST TIMED step is compared, bit for bit over the whole grid, with the same
                1000 generations computed by a different code path (general-coefficient kernel, one generation per
                launch, one row strip), and with the CPU oracle on windows on the rim of the square (see verify()).
  legs          random_init (the headline path on random data, with the GPU's shader clock sampled during both),
                general_coefficients / general_coefficients_fma (the kernel any C++ functor compiles to, with
                -ffp-contract=off and in the reference's own GPU compile mode), hotspot_8192, fdtd_max_grid -- each
                with its own `roofline` (algorithmic and physical fractions, binding resource) --, template_api /
                template_api_fma (the reference's unchanged examples/jacobi binary, its own `Walltime:`).
  cpu_baseline  the CPU port timed on this box's host cores, bounded sample.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md, chip-level parameters)
# what a float4 device-to-device copy reaches on MI355X according to the guide (6.29 TB/s); this repository's own
# copy microbenchmark on the pool's boxes reaches 4.9-5.5 TB/s (tools/microbench/hbm_pattern.hip,
# profiles/r02_hbm_pattern.txt) -- both are reported as attainable ceilings next to the 8 TB/s of the specification
HBM_COPY_GUIDE_GBS = 6290.0
HBM_COPY_MEASURED_GBS = 5400.0
BYTES_PER_CELL_UPDATE = 8  # 2 * sizeof(float) * n_subiterations (scripts/benchmark-common.jl:150-151)
COEF = [0.2, 0.2, 0.2, 0.2, 0.2]  # examples/jacobi/scripts/benchmark.jl:44-45
# VALU issue peak, two bases, both reported: (a) the guide's: one wave64 fp32 instruction per 2 cycles and SIMD
# (MI355X_MICROARCH.md, cycle constants: v_fma_f32 wave64 = 2 cycles on SIMD-32) at the 2.4 GHz peak clock = 0.833 ns;
# (b) this repository's own measurement with every SIMD busy, at the clock the chip holds under such a load:
# 1.09 ns (tools/microbench/valu_rate.hip, profiles/r01_microbench_valu_rate.txt).  `bound` is derived on basis (b),
# the one this chip has been seen to reach.
VALU_NS_GUIDE = 2.0 / 2.4
VALU_NS_MEASURED = 1.09
N_SIMDS = 1024
COUNTER_FILE = os.path.join(ROOT, "profiles", "r04_bench_counters.json")
EXAMPLES = os.path.join(ROOT, "build", "examples")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=16384, help="rows and columns of the whole grid")
    ap.add_argument("--config5", action="store_true", help="BASELINE configs[4]: the 65536^2 grid")
    ap.add_argument("--rows-per-gpu", type=int, default=0,
                    help="weak scaling instead: every rank owns this many rows of a (rows*N) x size grid")
    ap.add_argument("--generations", type=int, default=1000, help="generations per step")
    ap.add_argument("--exchange-every", type=int, default=0,
                    help="N > 1: exchange m*g ghost rows every m-th launch instead of g rows every launch "
                         "(0 = the library's default)")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: if the native RCCL communicator cannot be created, run the torch.distributed "
                         "driver (stencilstream_amd/dist.py) instead of failing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the template_api / hotspot / fdtd legs")
    ap.add_argument("--profile-legs", action="store_true",
                    help="run every kernel leg three times and nothing else (the program tools/profile_bench_r04.sh "
                         "profiles); prints the sequence of legs instead of the bench line")
    ap.add_argument("--debug-host-exchange", action="store_true",
                    help="debugging: all ranks on cuda:0, gloo process group, ghost rows through host memory")
    ap.add_argument("--strip-domain", action="store_true",
                    help="debugging: drive the sweep through the multi-GPU strip driver even with one rank")
    ap.add_argument("--single-strip", action="store_true",
                    help="one full-grid launch per pass (no row strips on side streams)")
    ap.add_argument("--cpu-size", type=int, default=8192)
    ap.add_argument("--cpu-generations", type=int, default=192)
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a child process (this process
    has not imported torch or touched HIP) and return its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def init_grid_device(torch, rows, cols, row0, total_rows, device):
    """Centred-square init of examples/jacobi/jacobi.cpp:114-122, generated on the device."""
    r = torch.arange(row0, row0 + rows, device=device, dtype=torch.float64)[:, None]
    c = torch.arange(cols, device=device, dtype=torch.float64)[None, :]
    inside = (r >= total_rows * 0.25) & (r < total_rows * 0.75) & (c >= cols * 0.25) & (c < cols * 0.75)
    return inside.to(torch.float32).contiguous()


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup's CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, cores)


def walltime_of(cmd, env=None, timeout=600):
    """`Walltime: <seconds>` printed by one of the example binaries (the application's own clock)."""
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    m = re.search(r"Walltime:\s*([0-9.eE+-]+)", out.stdout)
    if out.returncode != 0 or not m:
        raise RuntimeError(f"{os.path.basename(cmd[0])}: rc {out.returncode}: {(out.stdout + out.stderr)[-300:]}")
    return float(m.group(1))


def cpu_baseline(size, generations):
    """The CPU port on the host cores, bounded sample.  Preferred: this repository's stencil::cpu template backend
    (OpenMP) running the reference's unchanged examples/jacobi source; else the oracle."""
    cores = usable_cores()
    binary = os.path.join(EXAMPLES, "jacobi_Jacobi5General_cpu")
    fallback_reason = None
    if os.path.exists(binary):
        env = dict(os.environ, OMP_NUM_THREADS=str(cores))
        its = max(8, generations * 3)  # ~10-20 s of CPU work on 16 cores (3-3.5 Gcell-updates/s)
        try:
            dt = walltime_of([binary, str(size), str(size), str(its), "/dev/null"] + [str(c) for c in COEF], env=env)
            return {
                "value": size * size * its / dt / 1e9,
                "unit": "Gcell-updates/s",
                "cores": cores,
                "kind": "port",
                "implementation": "this repository's stencil::cpu backend (include/StencilStream/cpu/StencilUpdate.hpp, "
                                  "OpenMP over rows) compiled with g++ -O2 around the reference's UNCHANGED transition "
                                  "function (examples/jacobi/kernels.hpp); examples/jacobi_cpu.cpp is the reference's "
                                  "jacobi.cpp main() for the cpu backend it has no branch for; the application's own "
                                  "Walltime.  The reference's own cpu backend needs a SYCL compiler (icpx), absent here",
                "sample": f"Jacobi5General {size}x{size}, {its} generations, {dt:.1f} s",
            }
        except Exception as e:  # noqa: BLE001 -- fall through to the oracle, and say why
            fallback_reason = f"{os.path.basename(binary)} failed: {str(e)[:200]}"
    else:
        fallback_reason = f"build/examples/{os.path.basename(binary)} not built (needs /root/reference at build time)"
    from oracle import oracle as O

    grid = O.jacobi_init(size, size)
    O.jacobi("Jacobi5General", COEF, grid[:64, :64].copy(), 1, n_threads=cores)  # warm the pool
    t0 = time.perf_counter()
    O.jacobi("Jacobi5General", COEF, grid, generations, n_threads=cores)
    dt = time.perf_counter() - t0
    return {
        "value": size * size * generations / dt / 1e9,
        "unit": "Gcell-updates/s",
        "cores": cores,
        "kind": "port",
        "implementation": "oracle/stencil_oracle.c: plain-C restatement of the reference's cpu backend "
                          "(cpu/StencilUpdate.hpp:109-142,185-223), transition function through a pointer, rows "
                          "shared among threads with OpenMP",
        "sample": f"Jacobi5General {size}x{size}, {generations} generations, {dt:.1f} s",
        "fallback_reason": fallback_reason,
    }


def with_env(overrides):
    """Context manager: os.environ with `overrides` (the binding re-reads the library's knobs when they change)."""
    class _Env:
        def __enter__(self):
            self.saved = {k: os.environ.get(k) for k in overrides}
            os.environ.update(overrides)

        def __exit__(self, *exc):
            for k, v in self.saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return _Env()


def reference_run(torch, capi, p, halo, dom, src, gens, stream):
    """The same `gens` generations by a different code path: the general-coefficient kernel (no product-carrying
    form), ONE generation per launch (no temporal blocking), one row strip (no bands, no side streams)."""
    ref = torch.empty_like(src)
    with with_env({"STSTHIP_MAX_GENERATIONS": "1", "STSTHIP_VIRTUAL_STRIPS": "1", "STSTHIP_JACOBI_FASTPATH": "0"}):
        capi.app_run("jacobi5general", p, halo, dom, [src.data_ptr()], [ref.data_ptr()], 0, gens, blocking=True,
                     stream=stream.cuda_stream)
    return ref


def oracle_rim_windows(torch, result, src, gens, H, W, n_windows=2, w=64):
    """`w` x `w` windows of `result` (= `gens` generations of `src`) on the rim of the centred square against the CPU
    oracle, bit for bit.  A window's cells depend on the input patch widened by `gens` cells, so the oracle advances
    that patch (clipped to the grid) by the full `gens` generations."""
    import numpy as np

    from oracle import oracle as O

    lo_r, hi_r, lo_c = int(np.ceil(H * 0.25)), int(np.ceil(H * 0.75)), int(np.ceil(W * 0.25))
    spots = [(lo_r - w // 2, lo_c - w // 2), (hi_r - w // 2, W // 2)][:n_windows]  # corner and edge of the square
    ok, report = True, []
    cores = usable_cores()
    for r0, c0 in spots:
        r0, c0 = max(0, min(r0, H - w)), max(0, min(c0, W - w))
        ra, rb, ca, cb = max(0, r0 - gens), min(H, r0 + w + gens), max(0, c0 - gens), min(W, c0 + w + gens)
        t0 = time.perf_counter()
        with O.window(ra, ca, H, W):
            ref = O.jacobi("Jacobi5General", COEF, src[ra:rb, ca:cb].cpu().numpy(), gens, halo=0.0, n_threads=cores)
        want = ref[r0 - ra:r0 - ra + w, c0 - ca:c0 - ca + w]
        got = result[r0:r0 + w, c0:c0 + w].cpu().numpy()
        same = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
        # a window that straddles the rim holds many different values: say so, a constant window would prove little
        report.append({"row": r0, "col": c0, "size": w, "generations": gens, "equal": same,
                       "distinct_values": int(np.unique(got.view(np.uint32)).size),
                       "oracle_s": round(time.perf_counter() - t0, 2)})
        ok = ok and same
    return ok, report


def verify_timed(torch, capi, p, halo, dom, src, dst, gens, stream, H, W, windows=True):
    """Is `dst` -- what the LAST TIMED step left -- `gens` generations of `src`?"""
    ref = reference_run(torch, capi, p, halo, dom, src, gens, stream)
    equal = bool(torch.equal(dst.view(torch.int32), ref.view(torch.int32)))
    changed = int((ref != src).sum().item())  # cells the generations actually changed (the rim of the square)
    report = {"whole_grid_equal_to_T1_general_kernel": equal, "cells_compared": int(dst.numel()),
              "cells_that_differ_from_the_input": changed}
    ok = equal and changed > 0
    if windows:
        ok_w, rep_w = oracle_rim_windows(torch, dst, src, gens, H, W)
        report["oracle_windows"] = rep_w
        ok = ok and ok_w
    del ref
    return ok, report


def load_counters():
    try:
        return json.load(open(COUNTER_FILE))
    except (OSError, ValueError):
        return None


def physical_fractions(hbm_bytes, valu, seconds):
    """Fractions of the two physical roofs for work that moved `hbm_bytes` through HBM and issued `valu` VALU
    wave-instructions in `seconds`: HBM against the 8 TB/s of the specification (and against the two copy rates),
    VALU issue on both bases (VALU_NS_GUIDE, VALU_NS_MEASURED).  `bound` = the larger of the two fractions, VALU on
    the measured basis; `binding` says how close that roof is."""
    gbps = hbm_bytes / seconds / 1e9
    out = {"hbm_GBps": gbps, "physical_hbm": gbps / HBM_PEAK_GBS,
           "physical_hbm_of_copy_rate_guide_6.29TBps": gbps / HBM_COPY_GUIDE_GBS,
           "physical_hbm_of_copy_rate_measured_5.4TBps": gbps / HBM_COPY_MEASURED_GBS,
           "valu_issue": None, "bound": "hbm"}
    if valu:
        guide = valu * VALU_NS_GUIDE * 1e-9 / N_SIMDS / seconds
        measured = valu * VALU_NS_MEASURED * 1e-9 / N_SIMDS / seconds
        out["valu_issue"] = {"guide_2_cycles_at_2.4GHz": guide, "measured_1.09ns": measured}
        out["bound"] = "valu" if measured >= out["physical_hbm_of_copy_rate_guide_6.29TBps"] else "hbm"
        out["bound_rule"] = ("the larger of valu_issue.measured_1.09ns and physical_hbm_of_copy_rate_guide_6.29TBps: "
                             "each roof at the rate this chip has been seen to reach")
    return out


def leg_counters(name, launches_per_call):
    """PMC sums of one call of leg `name` from the committed profile of this file's own `--profile-legs` run
    (tools/profile_bench_r04.sh -> profiles/r04_bench_counters.json), or (None, why).  Attached only when the call
    made the same number of launches as the profiled one: another launch plan moves other bytes."""
    data = load_counters()
    if not data:
        return None, f"{os.path.relpath(COUNTER_FILE, ROOT)} is missing"
    leg = (data.get("legs") or {}).get(name)
    if not leg:
        return None, f"no leg {name} in {os.path.relpath(COUNTER_FILE, ROOT)}"
    if launches_per_call is not None and int(leg.get("launches_per_call", -1)) != int(launches_per_call):
        return None, (f"the profiled call made {leg.get('launches_per_call')} launches, this one {launches_per_call}: "
                      "another launch plan, counters not attached")
    return leg, None


def leg_roofline(name, launches_per_call, seconds_per_call, algorithmic_frac):
    """`roofline` of a leg: SURVEY 8(d)'s algorithmic fraction beside the physical ones of the same call."""
    roof = {"algorithmic_hbm": algorithmic_frac, "traffic": None}
    leg, why = leg_counters(name, launches_per_call)
    if not leg:
        roof["counters"] = why
        return roof
    roof["traffic"] = leg["hbm_bytes_per_call"]
    roof["traffic_unit"] = "HBM bytes per call of the leg (all its launches), PMC"
    roof["valu_wave_instructions"] = leg.get("valu_per_call")
    roof.update(physical_fractions(leg["hbm_bytes_per_call"], leg.get("valu_per_call"), seconds_per_call))
    if leg.get("sq"):
        roof["wave_cycles"] = leg["sq"]
    roof["kernel_shapes"] = [{k: sh.get(k) for k in ("kernel", "grid_size", "launches_per_call", "avg_us", "vgprs",
                                                     "lds_bytes", "scratch_bytes")} for sh in leg.get("shapes", [])[:4]]
    roof["counters"] = leg.get("source")
    return roof


class ClockSampler:
    """Average shader clock of the GPU while a leg runs, sampled in-process from the driver's sysfs files (a thread
    that reads a file every 20 ms; no subprocess, nothing on the GPU).  None where the box does not expose them."""

    def __init__(self, device_index=0):
        import glob

        self.files = []
        cards = sorted(glob.glob("/sys/class/drm/card*/device"))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk")) or
                 glob.glob(os.path.join(c, "hwmon", "hwmon*", "freq1_input"))]
        if device_index < len(cards):
            card = cards[device_index]
            self.files = glob.glob(os.path.join(card, "hwmon", "hwmon*", "freq1_input")) or \
                [os.path.join(card, "pp_dpm_sclk")]
        self.samples = []
        self._stop = None
        self._thread = None

    def _read(self):
        for f in self.files:
            try:
                text = open(f).read()
            except OSError:
                continue
            if f.endswith("freq1_input"):
                return int(text) / 1e6  # Hz -> MHz
            for line in text.splitlines():
                if line.rstrip().endswith("*"):
                    m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
                    if m:
                        return float(m.group(1))
        return None

    def __enter__(self):
        import threading

        self.samples = []
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                v = self._read()
                if v:
                    self.samples.append(v)
                self._stop.wait(0.02)

        if self.files:
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread:
            self._thread.join()

    def report(self):
        if not self.samples:
            return None
        label = None
        if self.files and self.files[0].endswith("freq1_input"):
            try:
                label = open(self.files[0].replace("_input", "_label")).read().strip()
            except OSError:
                pass
        return {"mean_MHz": sum(self.samples) / len(self.samples), "min_MHz": min(self.samples),
                "max_MHz": max(self.samples), "samples": len(self.samples), "source": self.files[0], "label": label}


def hotspot_params(capi, n):
    """examples/hotspot/hotspot.cpp:281-295 in numpy float32/float64 with the same expression types."""
    import numpy as np

    f32, f64 = np.float32, np.float64
    t_chip, chip = f32(0.0005), f32(0.016)
    gh, gw = f32(chip / f32(n)), f32(chip / f32(n))
    cap = f32(f64(0.5) * f64(1.75e6) * f64(t_chip) * f64(gh) * f64(gw))
    rx = f32(f64(gw) / (f64(2.0) * 100 * f64(t_chip) * f64(gh)))
    ry = f32(f64(gh) / (f64(2.0) * 100 * f64(t_chip) * f64(gw)))
    rz = f32(t_chip / f32(f32(f32(100) * gh) * gw))
    max_slope = f32(f64(3.0e6) / (f64(0.5) * f64(t_chip) * f64(1.75e6)))
    step = f32(f64(0.001) / f64(max_slope) / f64(1000.0))
    return capi.HotspotParams(float(f32(1) / rx), float(f32(1) / ry), float(f32(1) / rz), float(step / cap))


def best_time(fn, reps=3):
    """(best seconds, what the last call returned) of `reps` calls after one warm-up call"""
    last = fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        last = fn()
        best = min(best, time.perf_counter() - t0)
    return best, last


def kernel_legs(torch, capi, device, stream, gens, size=16384):
    """The legs that are sweeps on a grid resident in HBM, through the C ABI: name -> setup() -> dict(call, cells,
    bytes_per_cell_update, what).  `call()` runs ONE blocking ststhip_app_run and returns its run info.  Shared by the
    default run (timed, min of 3) and by `--profile-legs` (1 + 2 calls each under rocprofv3)."""
    import numpy as np

    halo0 = np.float32(0.0).tobytes()

    def jacobi(app, coef, init, n=gens, env=None, H=size, W=size):
        def setup():
            p = capi.JacobiParams()
            for i, c in enumerate(coef):
                p.coef[i] = c
            src = init(H, W)
            dst = torch.empty_like(src)
            dom = capi.Domain(H, W, 0, H, W)
            torch.cuda.synchronize()

            def call():
                with with_env(env or {}):
                    return capi.app_run(app, p, halo0, dom, [src.data_ptr()], [dst.data_ptr()], 0, n, blocking=True,
                                        stream=stream.cuda_stream)
            return {"call": call, "cells": H * W * n, "keep": (src, dst, p), "src": src, "dst": dst, "params": p,
                    "dom": dom, "generations": n, "app": app, "shape": (H, W)}
        return setup

    square = lambda H, W: init_grid_device(torch, H, W, 0, H, device)  # noqa: E731
    random = lambda H, W: torch.rand(H, W, device=device,  # noqa: E731
                                     generator=torch.Generator(device=device).manual_seed(0x5EED))
    distinct = [0.2, 0.21, 0.19, 0.22, 0.18]

    def hotspot():
        H = W = 8192
        temp = torch.full((H, W), 30.0, device=device)
        power = torch.zeros(H, W, device=device)
        power[H // 4 - 1:3 * H // 4, W // 4 - 1:3 * W // 4] = 0.5
        out = [torch.empty_like(temp), torch.empty_like(power)]
        dom = capi.Domain(H, W, 0, H, W)
        hp = hotspot_params(capi, H)
        torch.cuda.synchronize()
        n = 1000

        def call():
            return capi.app_run("hotspot", hp, np.zeros(2, np.float32).tobytes(), dom, [temp.data_ptr(), power.data_ptr()],
                                [t.data_ptr() for t in out], 0, n, blocking=True, stream=stream.cuda_stream)
        return {"call": call, "cells": H * W * n, "keep": (temp, power, out, hp), "generations": n}

    def fdtd():
        H = W = 4608
        p = capi.FdtdParams(dt=8.1e-19, t_0=3e-13, tau=1e-13, omega=7.5e14, cutoff_iteration=10 ** 9,
                            detect_iteration=0, source_radius_squared=100.0, source_r=H / 2, source_c=W / 2,
                            source_distance_bound=100.0 - 2 * (H / 2) ** 2, double_center_rc=float(H))
        material = torch.empty(H, W, 4, device=device)
        for i, v in enumerate((1.0, 0.3, 1.0, 0.3)):
            material[..., i] = v
        pa = [torch.rand(H, W, 4, device=device) * 1e-3, material]
        pb = [torch.empty_like(t) for t in pa]
        dom = capi.Domain(H, W, 0, H, W)
        torch.cuda.synchronize()
        n = 1200

        def call():
            return capi.app_run("fdtd_coef_grouped", p, np.zeros(8, np.float32).tobytes(), dom, [t.data_ptr() for t in pa],
                                [t.data_ptr() for t in pb], 0, n, blocking=True, stream=stream.cuda_stream)
        return {"call": call, "cells": H * W * n, "keep": (pa, pb, material, p), "generations": n}

    T = int(capi.app_info("jacobi5uniform").max_generations)
    return {
        # `value`'s own path, and the same kernel as full-grid launches (one row strip, launches of the full depth only:
        # what roofline.kernel_ms times)
        "headline": {"setup": jacobi("jacobi5general", COEF, square), "bytes": 8,
                     "what": f"Jacobi5General {size}^2, coefficients 5 x 0.2, halo 0, centred-square input: the uniform-"
                             "coefficient form, two row strips with a moving boundary (the path `value` times)"},
        "headline_full_grid": {"setup": jacobi("jacobi5general", COEF, square, n=15 * T, env={"STSTHIP_VIRTUAL_STRIPS": "1"}),
                               "bytes": 8, "what": "the same kernel as full-grid launches of the full depth on one stream"},
        "random_init": {"setup": jacobi("jacobi5general", COEF, random), "bytes": 8,
                        "what": f"the headline path on uniform random input in [0, 1) (torch.rand, seed 0x5EED; SURVEY 8(d) "
                                "config 2's random variant): every lane adds different numbers"},
        "general_coefficients": {"setup": jacobi("jacobi5general", distinct, square), "bytes": 8,
                                 "what": "Jacobi5General kernel with coefficients 0.2 0.21 0.19 0.22 0.18 (nine flops per "
                                         "cell: what any C++ functor compiles to) through the C ABI, resident grid"},
        "general_coefficients_fma": {"setup": jacobi("jacobi5general_fma", distinct, square), "bytes": 8,
                                     "what": "the same function compiled with -ffp-contract=fast, the reference's own GPU "
                                             "compile mode (five fused operations per cell; results within the stated fp32 "
                                             "tolerance, tests/test_parity_gpu.py), kernel rate on a resident grid"},
        "hotspot_8192": {"setup": hotspot, "bytes": 16,
                         "what": "HotSpot 2 x fp32, 8192^2, per-field planes, 1000 generations, resident grid (BASELINE "
                                 "configs[2]); roofline 16 B per cell-update at 8 TB/s"},
        "fdtd_max_grid": {"setup": fdtd, "bytes": 128,
                          "what": "FDTD coef resolver, 4608^2 (max_grid.json's grid), 2 sub-iterations, the cell as two planes "
                                  "of 16-byte halves, kernel rate on a resident grid (BASELINE configs[3]); roofline 128 B per "
                                  "cell-update at 8 TB/s"},
    }


def run_kernel_leg(torch, name, leg, reps=3, clock=False):
    """Time one leg (min of `reps` calls after a warm-up) and attach its roofline."""
    state = leg["setup"]()
    sampler = ClockSampler() if clock else None
    if sampler:
        with sampler:
            dt, info = best_time(state["call"], reps)
    else:
        dt, info = best_time(state["call"], reps)
    rate = state["cells"] / dt / 1e9
    out = {"value": rate, "unit": "Gcell-updates/s", "generations": state["generations"], "s": dt,
           "launches_per_call": int(info.n_launches), "what": leg["what"],
           "frac_of_hbm_roofline": rate * leg["bytes"] / HBM_PEAK_GBS,
           "roofline": leg_roofline(name, int(info.n_launches), dt, rate * leg["bytes"] / HBM_PEAK_GBS)}
    if sampler:
        out["sclk"] = sampler.report()
    if "app" in state:
        from stencilstream_amd import capi

        depth = capi.app_tuned_depth(state["app"], *state["shape"])
        if depth:
            out["blocking_depth_by_measurement"] = depth  # the pass driver timed the family's two depths (ststhip.h)
    return out, state


def example_leg(binary_name, args, what, cells):
    """An unchanged example binary by its own `Walltime:` (min of 3 after one warm-up run, as
    examples/jacobi/scripts/benchmark.jl:43-94)."""
    binary = os.path.join(EXAMPLES, binary_name)
    if not os.path.exists(binary):
        return {"value": None, "why": f"build/examples/{binary_name} not built (needs /root/reference at build time)"}
    try:
        cmd = [binary] + args
        walltime_of(cmd)
        dt = min(walltime_of(cmd) for _ in range(3))
        return {"value": cells / dt / 1e9, "unit": "Gcell-updates/s", "walltime_s": dt, "what": what}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "why": str(e)[:200]}


FIRST_USE_NOTE = ("outside Walltime (as the reference excludes queue creation, cuda/StencilUpdate.hpp:124-128): creation "
                  "of the runtime's stream set and the copy engines' first use (a 1 MiB copy each way), which happen "
                  "when the host first asks the runtime for pinned memory, i.e. when the example allocates its grid; "
                  "and the load of the update's code object, which happens when the StencilUpdate object is built")


def profile_legs(torch, capi, device, stream, gens):
    """`--profile-legs`: every kernel leg once as warm-up and twice more, one after the other, nothing else -- the
    program tools/profile_bench_r04.sh runs under rocprofv3 (--kernel-trace, then one --pmc pass per counter group).
    Prints the sequence (leg, calls, launches per call) so that the summariser can cut the dispatch-ordered list of
    sweep launches into legs without looking at kernel names."""
    sequence = []
    # depths the unprofiled run chose for the families whose depth is measured ("leg=depth,..."): the counter passes
    # must plan the same launches, whatever the counters do to the timing of the probes
    pinned = dict(item.split("=") for item in os.environ.get("STSTHIP_BENCH_LEG_DEPTHS", "").split(",") if "=" in item)
    for name, leg in kernel_legs(torch, capi, device, stream, gens).items():
        if name == "random_init":
            continue  # the headline's kernels on other data: same launches, same counters
        state = leg["setup"]()
        launches, seconds = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            with with_env({"STSTHIP_TUNE_DEPTH": pinned[name]} if name in pinned else {}):
                info = state["call"]()
            seconds.append(time.perf_counter() - t0)
            launches.append(int(info.n_launches))
        # (a family whose depth is measured probes on its first call: that call makes other launches than the later ones;
        # the summariser averages over the calls that made as many launches as the last one)
        depth = capi.app_tuned_depth(state["app"], *state["shape"]) if "app" in state else 0
        sequence.append({"leg": name, "launches": launches, "launches_per_call": launches[-1], "cells_per_call": state["cells"],
                         "s_per_call_min": min(seconds), "bytes_per_cell_update": leg["bytes"],
                         "depth_by_measurement": depth or (int(pinned[name]) if name in pinned else None)})
        del state
        torch.cuda.empty_cache()
    print(json.dumps({"profile_legs": sequence}), flush=True)


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.single_strip:
        os.environ["STSTHIP_VIRTUAL_STRIPS"] = "1"
    if args.exchange_every > 0:
        os.environ["STSTHIP_EXCHANGE_EVERY"] = str(args.exchange_every)
    import numpy as np
    import torch

    from stencilstream_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or run `python bench.py --gpus {args.gpus}` plainly")
    if args.debug_host_exchange:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    capi.init(local_rank)
    stream = torch.cuda.Stream(device)  # the stream every sweep is launched on (and timed on)
    if args.profile_legs:
        profile_legs(torch, capi, device, stream, args.generations)
        return

    W, gens = (65536 if args.config5 else args.size), args.generations
    if args.rows_per_gpu:
        total_rows, scaling = args.rows_per_gpu * world, "weak"
    else:
        total_rows, scaling = W, "strong"
    p = capi.JacobiParams()
    for i, c in enumerate(COEF):
        p.coef[i] = c
    halo = np.float32(0.0).tobytes()
    app = "jacobi5general"
    # five equal positive coefficients and a +0 halo: the runtime uses the bit-identical product-carrying
    # form of the kernel (Jacobi5Uniform, 5 instead of 9 flops per cell)
    uniform = os.environ.get("STSTHIP_JACOBI_FASTPATH", "1") != "0"
    info = capi.app_info("jacobi5uniform" if uniform else app)
    ranks_report = None
    H = total_rows
    whole = capi.Domain(H, W, 0, H, W)

    if world == 1 and not args.strip_domain:
        src = init_grid_device(torch, H, W, 0, H, device)
        dst = torch.empty_like(src)
        dom = whole
        torch.cuda.synchronize()  # the grid is resident before anything runs on `stream`
        last = {}

        def step():
            last["info"] = capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens,
                                        blocking=False, profiling=False, stream=stream.cuda_stream)

        barrier = lambda: None
        decomposition = "none"
    else:
        import torch.distributed as dist

        from stencilstream_amd.dist import StripDomain, split_rows

        if world == 1:
            pass  # --strip-domain: no process group, no neighbours
        elif args.debug_host_exchange:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        # Ghost rows: the native strip driver of libststhip.so (ststhip_strip_advance: RCCL send/recv issued from
        # C++, no Python between the launches of a step).
        native = not args.debug_host_exchange and os.environ.get("STSTHIP_BENCH_EXCHANGE", "native") == "native"
        comm, why = None, None
        if native and world > 1:
            # (every rank takes part in the broadcast whatever happened on rank 0: a failure there must not leave
            # the others waiting in a collective)
            uid = [None]
            if rank == 0:
                try:
                    uid = [capi.comm_unique_id()]
                except Exception as e:  # noqa: BLE001
                    why = f"{type(e).__name__}: {e}"
            dist.broadcast_object_list(uid, src=0)
            if uid[0] is not None:
                try:
                    comm = capi.comm_create(uid[0], rank, world)
                except Exception as e:  # noqa: BLE001
                    why = f"{type(e).__name__}: {e}"
            flag = torch.tensor([0 if comm is None else 1], device=device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native = bool(flag.item())
            if not native and not args.allow_fallback:
                # a scaling number from another driver than the one DESIGN.md describes would mislead: fail loudly
                sys.stderr.write(f"bench.py rank {rank}: the native RCCL communicator could not be created"
                                 f"{' (' + why + ')' if why else ' on another rank'}; pass --allow-fallback to run the "
                                 "torch.distributed driver instead\n")
                dist.destroy_process_group()
                sys.exit(3)
        a, b = split_rows(total_rows, world)[rank]
        init = init_grid_device(torch, b - a, W, a, total_rows, device)
        if native or args.debug_host_exchange and os.environ.get("STSTHIP_BENCH_EXCHANGE", "native") == "native":
            exchange_cb = None
            if args.debug_host_exchange and world > 1:
                from stencilstream_amd.dist import host_exchange_callback

                exchange_cb = host_exchange_callback(rank, world)
            strip = capi.Strip(app, p, halo, total_rows, W, rank, world, comm=comm, exchange=exchange_cb)
            assert (strip.row_begin, strip.row_end) == (a, b)
            torch.cuda.synchronize()
            native = True
            exchange = ("RCCL send/recv issued by libststhip.so (ststhip_strip_advance)" if comm is not None else
                        "libststhip.so strip driver, ghost rows staged through host memory over gloo (debugging)"
                        if world > 1 else "libststhip.so strip driver, one strip")

            def load_init():
                strip.upload_from_device(0, init.data_ptr(), init.numel() * init.element_size())
                strip.synchronize()

            def owned_rows():
                ptr, row_bytes = strip.plane(0)
                strip.synchronize()
                out = torch.empty(b - a, W, device=device, dtype=torch.float32)
                capi.check(capi.load().ststhip_memcpy_d2d(capi.C.c_void_p(out.data_ptr()), capi.C.c_void_p(ptr),
                                                          out.numel() * 4, capi.C.c_void_p(strip.stream)), "d2d")
                strip.synchronize()
                return out

            def step():
                strip.advance(0, gens)
        else:
            strip = StripDomain(app, p, halo, total_rows, W, rank, world, device, np.dtype("<f4"),
                                exchange_via_host=args.debug_host_exchange)
            exchange = "torch.distributed batch_isend_irecv (stencilstream_amd/dist.py)" + (f"; native: {why}" if why else "")

            def load_init():
                strip.load_owned(init)

            def owned_rows():
                return strip.owned(0, torch.float32).clone()

            def step():
                strip.advance(0, gens)
        load_init()
        if world > 1:
            if native:
                strip.warm_up()  # RCCL creates its p2p channels on first use: outside the timed region
            else:
                strip.warm_up_exchange()
            # what the driver can check the launch against: ranks, devices, RCCL
            mine = {"rank": rank, "device": torch.cuda.get_device_name(device), "local_rank": local_rank,
                    "rows": [a, b]}
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            ranks_report = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                            "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version())
                            if not args.debug_host_exchange else None,
                            "ghost_exchange": exchange, "exchange_every": capi.options().exchange_every or 1,
                            "ranks": gathered}

        barrier = dist.barrier if world > 1 else (lambda: None)
        decomposition = f"{world} row strips of {total_rows // world} rows; {exchange}"

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    strips_verified, strips_report = None, None
    if world > 1 or args.strip_domain:
        import torch.distributed as dist

        cpu_side = args.debug_host_exchange
        if world > 1:
            t = torch.tensor([elapsed], device="cpu" if cpu_side else device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if not args.no_verify:
            # The strips' state has advanced through warm-up and timed steps, so the check runs ONE more step from
            # the initial grid with the very same driver (outside the timed region) and compares every rank's rows,
            # bit for bit, with the same generations of the WHOLE grid computed on this rank's own GPU by the
            # single-GPU pass driver -- rim of the square, strip boundaries and all.  Rank 0 additionally holds that
            # single-GPU result against the T = 1 general kernel and the CPU oracle (verify_timed).
            load_init()
            step()
            mine = owned_rows()
            src_all = init_grid_device(torch, H, W, 0, H, device)
            ref_all = torch.empty_like(src_all)
            torch.cuda.synchronize()
            capi.app_run(app, p, halo, whole, [src_all.data_ptr()], [ref_all.data_ptr()], 0, gens, blocking=True,
                         stream=stream.cuda_stream)
            ok = bool(torch.equal(mine.view(torch.int32), ref_all[a:b].view(torch.int32)))
            changed = int((ref_all[a:b] != src_all[a:b]).sum().item())
            report0 = None
            if rank == 0:
                ok0, report0 = verify_timed(torch, capi, p, halo, whole, src_all, ref_all, gens, stream, H, W)
                ok = ok and ok0
            if world > 1:
                flag = torch.tensor([1 if ok else 0], device="cpu" if cpu_side else device, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = bool(flag.item())
                total_changed = torch.tensor([changed], device="cpu" if cpu_side else device, dtype=torch.int64)
                dist.all_reduce(total_changed, op=dist.ReduceOp.SUM)
                changed = int(total_changed.item())
            strips_verified = ok and changed > 0
            strips_report = {"what": "every rank: its rows after one more step of the same driver from the initial grid, "
                                     "bit for bit against the same generations of the whole grid on its own GPU (single-GPU "
                                     "pass driver); rank 0: that single-GPU result against the T = 1 general kernel (whole "
                                     "grid) and the CPU oracle (rim windows, all generations)",
                             "cells_that_differ_from_the_input": changed, "single_gpu_reference": report0}
            del src_all, ref_all, mine

    if rank == 0:
        cells = total_rows * W * gens * args.steps
        value = cells / elapsed / 1e9
        ms_per_step = elapsed / args.steps * 1e3
        form = "jacobi5uniform" if uniform else app
        out = {
            "metric": f"Gcell-updates/s, Jacobi-5pt fp32 {total_rows}x{W}, {world} GPU",
            "value": value,
            "unit": "Gcell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"Jacobi5General fp32 {total_rows}x{W}, {gens} generations per step, halo 0, coef 5x0.2 "
                            f"(BASELINE configs[{4 if args.config5 else 1}])",
                "kernel_form": "Jacobi5Uniform (bit-identical product-carrying form of Jacobi5General)" if uniform
                               else "Jacobi5General",
                "generations_total": gens * args.steps,
                "temporal_blocking": int(info.max_generations),
                "stages": int(info.stages),
                "decomposition": decomposition,
            },
            "hbm_roofline_gcells": HBM_PEAK_GBS / BYTES_PER_CELL_UPDATE,
            "frac_of_hbm_roofline": value * BYTES_PER_CELL_UPDATE / HBM_PEAK_GBS / world,
        }
        if ranks_report:
            out["ranks"] = ranks_report
        if strips_verified is not None:
            out["verified"] = strips_verified
            out["verification"] = strips_report
        if world == 1 and not args.strip_domain:
            launches_per_step = int(last["info"].n_launches)
            out["launches_per_step"] = launches_per_step
            if not args.no_verify:
                ok, report = verify_timed(torch, capi, p, halo, dom, src, dst, gens, stream, total_rows, W)
                out["verified"] = bool(ok)
                out["verification"] = report
            # The dominant (only) kernel, measured live: full-grid launches (one row strip) of T
            # generations, HIP events around every launch on the launch stream.
            depth = int(info.max_generations)
            gens_timed = max(depth, min(gens, 240) // depth * depth)  # launches of the full depth only
            with with_env({"STSTHIP_VIRTUAL_STRIPS": "1"}):
                prof = capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens_timed,
                                    blocking=True, profiling=True, stream=stream.cuda_stream)
            kernel_ms = prof.kernel_time_s * 1e3 / max(prof.n_launches, 1)
            T = gens_timed / max(prof.n_launches, 1)
            alg_bytes = total_rows * W * BYTES_PER_CELL_UPDATE * T
            achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
            default_size = (total_rows, W) == (16384, 16384) and gens == 1000
            # physical fractions, from the committed PMC passes of this file's own legs (profiles/r04_bench_counters.json)
            # over THIS run's times.  The lead is the timed path -- what `value` is --, the full-grid launch the footnote.
            timed, why_t = leg_counters("headline", launches_per_step) if default_size else (None, "not the profiled grid")
            full, why_f = leg_counters("headline_full_grid", None) if default_size else (None, "not the profiled grid")
            roof = {
                "bound": "valu",  # replaced below by the fractions' verdict when the counters are there
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "kernel": f"sweep_kernel<Sweep<{form}, T={int(info.max_generations)}, "
                          f"K={int(info.cells_per_lane)}, P={int(info.prefetch_rows)}, stages={int(info.stages)}>>",
                "kernel_ms": kernel_ms,
                "launches_timed": int(prof.n_launches),
                "generations_per_launch": T,
                "algorithmic_bytes_per_launch": alg_bytes,
                "attainable_copy_rate_GBps": {"guide_float4_copy": HBM_COPY_GUIDE_GBS,
                                              "this_repo_copy_microbenchmark": HBM_COPY_MEASURED_GBS},
                "valu_ns_per_wave_instruction": {"guide_2_cycles_at_2.4GHz": VALU_NS_GUIDE, "measured": VALU_NS_MEASURED},
                "note": "achieved / frac: SURVEY 8(d)'s algorithmic bytes (8 B x cell-updates of one launch) over the live "
                        "launch time of a FULL-GRID launch (kernel_ms, HIP events); above 1 because a launch advances T "
                        "generations while a cell crosses HBM once (temporal blocking).  The physical fractions: "
                        "`timed_path` = HBM bytes and VALU wave-instructions of ONE TIMED STEP (all its launches: two row "
                        "strips with a moving boundary, side by side; PMC) over this run's ms_per_step -- what "
                        "`value` is; `full_grid_launch` = the same counters per full-grid launch over kernel_ms.  `bound` is "
                        "the timed path's.",
            }
            if timed:
                roof["timed_path"] = {"launches_per_step": launches_per_step, "hbm_bytes_per_step": timed["hbm_bytes_per_call"],
                                      "valu_per_step": timed.get("valu_per_call"),
                                      "sum_kernel_ms_per_step": timed.get("sum_kernel_ms_per_call"),
                                      **physical_fractions(timed["hbm_bytes_per_call"], timed.get("valu_per_call"),
                                                           ms_per_step * 1e-3),
                                      "wave_cycles": timed.get("sq"), "launch_shapes": timed.get("shapes"),
                                      "counters": timed.get("source")}
                roof["bound"] = roof["timed_path"]["bound"]
            else:
                roof["timed_path"] = {"counters": why_t}
            if full:
                n = max(int(full["launches_per_call"]), 1)
                roof["traffic"] = full["hbm_bytes_per_call"] / n
                roof["full_grid_launch"] = {"hbm_bytes_per_launch": full["hbm_bytes_per_call"] / n,
                                            "valu_per_launch": (full.get("valu_per_call") or 0) / n or None,
                                            **physical_fractions(full["hbm_bytes_per_call"] / n,
                                                                 (full.get("valu_per_call") or 0) / n or None,
                                                                 kernel_ms * 1e-3),
                                            "wave_cycles": full.get("sq"), "profiled_avg_us": full.get("avg_us_per_launch"),
                                            "counters": full.get("source")}
            else:
                roof["full_grid_launch"] = {"counters": why_f}
            out["roofline"] = roof
            extras = os.environ.get("STSTHIP_BENCH_MINIMAL", "0") == "0"  # profiling runs skip the extra legs
            if extras and not args.no_legs:
                del src, dst
                torch.cuda.empty_cache()
                legs = {}
                catalogue = kernel_legs(torch, capi, device, stream, gens, size=W)
                for name in ("random_init", "general_coefficients", "general_coefficients_fma", "hotspot_8192",
                             "fdtd_max_grid"):
                    if name.startswith("general") and not uniform:
                        continue
                    try:
                        legs[name], state = run_kernel_leg(torch, name, catalogue[name], reps=2 if "general" in name else 3,
                                                           clock=(name == "random_init"))
                        if name == "random_init":
                            # the headline again on the same box and clock sampler, so that the two are comparable
                            head, hstate = run_kernel_leg(torch, "headline", catalogue["headline"], reps=3, clock=True)
                            legs[name]["centred_square_same_protocol"] = {k: head[k] for k in ("value", "s", "sclk")}
                            del hstate
                            if not args.no_verify:
                                ok, report = verify_timed(torch, capi, state["params"], halo, state["dom"], state["src"],
                                                          state["dst"], gens, stream, total_rows, W, windows=False)
                                legs[name]["verified"] = bool(ok)
                                legs[name]["verification"] = report
                        del state
                    except Exception as e:  # noqa: BLE001
                        legs[name] = {"value": None, "why": f"{type(e).__name__}: {str(e)[:200]}"}
                    torch.cuda.empty_cache()
                # configs[1] through the template API: the reference's unchanged examples/jacobi/jacobi.cpp compiled against
                # this backend (a C++ functor is opaque: the nine-flop general kernel); the application's own `Walltime:`
                # (allocation of the swap grids, upload of the 1 GiB grid over PCIe, sweeps -- cuda/StencilUpdate.hpp:123-144)
                args_j = ["16384", "16384", str(gens), "/dev/null"] + [str(c) for c in COEF]
                legs["template_api"] = example_leg(
                    "jacobi_Jacobi5General_hip", args_j,
                    f"build/examples/jacobi_Jacobi5General_hip 16384 16384 {gens} /dev/null 0.2 x5: the reference's unchanged "
                    "source on stencil::hip, bit-identical to the cpu backend (-ffp-contract=off); the application's own "
                    "Walltime, which includes the upload of the 1 GiB grid -- in row blocks that the pass driver follows "
                    "(ABI 6, DESIGN.md 3.7; STSTHIP_STREAM_UPLOAD=0: one copy in front of the first pass, 0.083 s); min of 3 "
                    "after a warm-up; " + FIRST_USE_NOTE,
                    16384 * 16384 * gens)
                legs["template_api_fma"] = example_leg(
                    "jacobi_Jacobi5General_hip_fma", args_j,
                    "the same unchanged source compiled as the reference compiles its GPU builds (no -ffp-contract flag in "
                    "its CMakeLists.txt:46-51: multiply-adds fused, five operations per cell; results within abs 1e-5 of "
                    "the cpu backend after 1000 generations, tests/test_examples.py); its own Walltime, upload included; "
                    "min of 3 after a warm-up; " + FIRST_USE_NOTE, 16384 * 16384 * gens)
                out["legs"] = legs
            if not args.no_cpu_baseline and extras:
                out["cpu_baseline"] = cpu_baseline(args.cpu_size, args.cpu_generations)
        print(json.dumps(out), flush=True)

    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()
    if strips_verified is False:
        sys.exit(4)


if __name__ == "__main__":
    main()
