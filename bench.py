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
                this exceeds the 8 TB/s peak by design (each cell moves once per T generations), so the line
                also carries the physical fractions from the PMC passes of this very command
                (profiles/r03_bench_counters.json): HBM bytes and VALU wave-instructions per launch, and
                `timed_path`: the same two sums over the launches of ONE TIMED STEP (interiors and bands of the
                row strips), so that bytes per step / ms_per_step can be held against 8 TB/s.
  verified      the result of the LAST TIMED step is compared, bit for bit over the whole grid, with the same
                1000 generations computed by a different code path (general-coefficient kernel, one generation per
                launch, one row strip), and with the CPU oracle on windows on the rim of the square (see verify()).
  legs          template_api (the reference's unchanged examples/jacobi binary, its own `Walltime:`), hotspot_8192,
                fdtd_max_grid, general_coefficients: the other BASELINE configurations and the API path.
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
# VALU issue peak: one wave64 fp32 instruction per 2 cycles and SIMD (MI355X_MICROARCH.md, cycle constants) at the
# ~2.06 GHz the chip holds under this load = 0.97 ns; measured with every SIMD busy: 0.97-1.09 ns
# (tools/microbench/valu_rate.hip, profiles/r01_microbench_valu_rate.txt).  The fractions use the measured 1.09.
VALU_NS_PER_WAVE_INSTRUCTION = 1.09
N_SIMDS = 1024
COUNTER_FILE = os.path.join(ROOT, "profiles", "r03_bench_counters.json")
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
        except Exception:  # noqa: BLE001 -- fall through to the oracle
            pass
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


def attach_counters(roofline, kernel_key, kernel_ms, ms_per_step, launches_per_step):
    """HBM bytes and VALU instructions per launch (dominant kernel) and per timed step from the committed PMC
    passes of this command (tools/profile_bench_r03.sh -> profiles/r03_bench_counters.json)."""
    roofline["traffic"] = None
    roofline["fractions"] = {"algorithmic_hbm": roofline["frac"], "physical_hbm": None, "valu_issue": None}
    roofline["timed_path"] = None
    data = load_counters()
    counters = (data or {}).get("kernels", {}).get(kernel_key)
    if not counters:
        roofline["counters"] = f"no PMC profile of {kernel_key} in {os.path.relpath(COUNTER_FILE, ROOT)}"
        return
    traffic = counters["hbm_bytes_per_launch"]
    roofline["traffic"] = traffic
    f = roofline["fractions"]
    f["physical_hbm"] = traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    f["physical_hbm_of_copy_rate_guide_6.29TBps"] = traffic / (kernel_ms * 1e-3) / 1e9 / HBM_COPY_GUIDE_GBS
    f["physical_hbm_of_copy_rate_measured_5.4TBps"] = traffic / (kernel_ms * 1e-3) / 1e9 / HBM_COPY_MEASURED_GBS
    valu = counters.get("valu_wave_instructions_per_launch")
    if valu:
        f["valu_issue"] = valu * VALU_NS_PER_WAVE_INSTRUCTION * 1e-6 / N_SIMDS / kernel_ms
        roofline["valu_wave_instructions_per_launch"] = valu
        roofline["bound"] = "valu" if f["valu_issue"] >= f["physical_hbm"] else "hbm"
    roofline["counters"] = counters.get("source")
    step = (data or {}).get("timed_step", {}).get(kernel_key)
    if step and step.get("launches_per_step") == launches_per_step:
        hbm, valu_step = step["hbm_bytes_per_step"], step.get("valu_wave_instructions_per_step")
        roofline["timed_path"] = {
            "launches_per_step": launches_per_step,
            "launch_shapes": step.get("launch_shapes"),
            "hbm_bytes_per_step": hbm,
            "valu_per_step": valu_step,
            "physical_hbm_frac": hbm / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "valu_frac": (valu_step * VALU_NS_PER_WAVE_INSTRUCTION * 1e-6 / N_SIMDS / ms_per_step) if valu_step else None,
            "sum_kernel_ms_per_step": step.get("sum_kernel_ms_per_step"),
            "note": "sums over the launches of one timed step (interiors and bands of the row strips, from the same "
                    "PMC passes) over THIS run's ms_per_step",
        }


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


def best_rate(fn, cells, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return cells / best / 1e9, best


def extra_legs(torch, capi, device, stream, gens):
    """The other BASELINE configurations and the API path, each with its grid resident in HBM unless it says
    otherwise; none of them is `value`."""
    import numpy as np

    legs = {}
    # configs[1] through the template API: the reference's unchanged examples/jacobi/jacobi.cpp, compiled against
    # this backend; the application's own `Walltime:` (allocation of the swap grids, upload, sweeps --
    # cuda/StencilUpdate.hpp:123-144), min of 3 after one warm-up run as examples/jacobi/scripts/benchmark.jl:43-94
    binary = os.path.join(EXAMPLES, "jacobi_Jacobi5General_hip")
    if os.path.exists(binary):
        try:
            cmd = [binary, "16384", "16384", str(gens), "/dev/null"] + [str(c) for c in COEF]
            walltime_of(cmd)
            dt = min(walltime_of(cmd) for _ in range(3))
            legs["template_api"] = {
                "value": 16384 * 16384 * gens / dt / 1e9, "unit": "Gcell-updates/s", "walltime_s": dt,
                "what": f"build/examples/jacobi_Jacobi5General_hip 16384 16384 {gens} /dev/null 0.2 x5: the reference's "
                        "unchanged source on stencil::hip (a C++ functor is opaque: the 9-flop general kernel); the "
                        "application's own Walltime, which includes the upload of the 1 GiB grid; min of 3 after a warm-up"}
        except Exception as e:  # noqa: BLE001
            legs["template_api"] = {"value": None, "why": str(e)[:200]}
    else:
        legs["template_api"] = {"value": None, "why": "build/examples/jacobi_Jacobi5General_hip not built (needs /root/reference at build time)"}
    # configs[2]: HotSpot fp32 8192^2 on per-field planes (the reference is fp32)
    try:
        H = W = 8192
        temp = torch.full((H, W), 30.0, device=device)
        power = torch.zeros(H, W, device=device)
        power[H // 4 - 1:3 * H // 4, W // 4 - 1:3 * W // 4] = 0.5
        out = [torch.empty_like(temp), torch.empty_like(power)]
        dom = capi.Domain(H, W, 0, H, W)
        hp = hotspot_params(capi, H)
        torch.cuda.synchronize()
        n = 1000
        rate, dt = best_rate(lambda: capi.app_run("hotspot", hp, np.zeros(2, np.float32).tobytes(), dom,
                                                  [temp.data_ptr(), power.data_ptr()], [t.data_ptr() for t in out],
                                                  0, n, blocking=True, stream=stream.cuda_stream), H * W * n)
        legs["hotspot_8192"] = {"value": rate, "unit": "Gcell-updates/s", "generations": n, "s": dt,
                                "frac_of_hbm_roofline": rate * 16 / HBM_PEAK_GBS,
                                "what": "HotSpot 2 x fp32, 8192^2, per-field planes, 1000 generations, resident grid "
                                        "(BASELINE configs[2]); roofline 16 B per cell-update at 8 TB/s"}
        del temp, power, out
    except Exception as e:  # noqa: BLE001
        legs["hotspot_8192"] = {"value": None, "why": str(e)[:200]}
    # configs[3]: FDTD coefficient resolver on the 4608^2 grid of max_grid.json, kernel rate
    try:
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
        rate, dt = best_rate(lambda: capi.app_run("fdtd_coef_grouped", p, np.zeros(8, np.float32).tobytes(), dom,
                                                  [t.data_ptr() for t in pa], [t.data_ptr() for t in pb], 0, n,
                                                  blocking=True, stream=stream.cuda_stream), H * W * n)
        legs["fdtd_max_grid"] = {"value": rate, "unit": "Gcell-updates/s", "generations": n, "s": dt,
                                 "frac_of_hbm_roofline": rate * 128 / HBM_PEAK_GBS,
                                 "what": "FDTD coef resolver, 4608^2 (max_grid.json's grid), 2 sub-iterations, the cell as "
                                         "two planes of 16-byte halves, kernel rate on a resident grid (BASELINE "
                                         "configs[3]); roofline 128 B per cell-update at 8 TB/s"}
        del pa, pb, material
    except Exception as e:  # noqa: BLE001
        legs["fdtd_max_grid"] = {"value": None, "why": str(e)[:200]}
    torch.cuda.empty_cache()
    return legs


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
            kernel_key = f"{form}:{total_rows}x{W}:T{int(T)}"
            out["roofline"] = {
                "bound": "valu",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "kernel": f"sweep_kernel<Sweep<{form}, T={int(info.max_generations)}, "
                          f"K={int(info.cells_per_lane)}, P={int(info.prefetch_rows)}, stages={int(info.stages)}>>",
                "kernel_key": kernel_key,
                "kernel_ms": kernel_ms,
                "launches_timed": int(prof.n_launches),
                "generations_per_launch": T,
                "algorithmic_bytes_per_launch": alg_bytes,
                "attainable_copy_rate_GBps": {"guide_float4_copy": HBM_COPY_GUIDE_GBS,
                                              "this_repo_copy_microbenchmark": HBM_COPY_MEASURED_GBS},
                "note": "achieved/frac: SURVEY 8d algorithmic bytes (8 B x cell-updates) over the launch time; it "
                        "exceeds 1 because one launch advances T generations while a cell moves through HBM once "
                        "(temporal blocking).  fractions.physical_hbm = PMC bytes / launch time / 8 TB/s; "
                        "fractions.valu_issue = VALU wave-instructions x 1.09 ns / 1024 SIMDs / launch time.  "
                        "kernel_ms is a full-grid launch; the timed steps run two row strips side by side with their "
                        "boundary bands on streams of their own: timed_path has their sums",
            }
            attach_counters(out["roofline"], kernel_key, kernel_ms, ms_per_step, launches_per_step)
            extras = os.environ.get("STSTHIP_BENCH_MINIMAL", "0") == "0"  # profiling runs skip the extra legs
            if extras and not args.no_legs:
                legs = {}
                if uniform:
                    # the general-coefficient kernel (nine flops per cell), same grid, coefficients that differ
                    q = capi.JacobiParams()
                    for i, c in enumerate([0.2, 0.21, 0.19, 0.22, 0.18]):
                        q.coef[i] = c
                    rate, _ = best_rate(lambda: capi.app_run(app, q, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0,
                                                             gens, blocking=True, stream=stream.cuda_stream),
                                        total_rows * W * gens, reps=2)
                    legs["general_coefficients"] = {"value": rate, "unit": "Gcell-updates/s",
                                                    "what": "Jacobi5General kernel with coefficients 0.2 0.21 0.19 0.22 "
                                                            "0.18 through the C ABI, resident grid"}
                del src, dst
                torch.cuda.empty_cache()
                legs.update(extra_legs(torch, capi, device, stream, gens))
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
