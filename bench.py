#!/usr/bin/env python3
"""Headline benchmark: Jacobi 5-point (Jacobi5General) fp32 on a synthetic 16384 x 16384 grid.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  A "step" is one StencilUpdate call of
`--generations` generations (default 252 = 21 launches of 12 generations; four steps are 1008 generations,
BASELINE.json's 1000 rounded up to whole launches, and the default 8 steps run that twice).  Rank 0 prints ONE JSON line.

Metric: Gcell-updates/s = H*W*generations / time (sub-iterations not counted), the reference's
definition (scripts/benchmark-common.jl:97-98,122).  The grid is resident in HBM before the timed
region starts.  N > 1 is weak scaling over row strips: every rank owns a 16384-row strip of a
(16384*N) x 16384 grid and exchanges ghost rows with its neighbours over RCCL once per launch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md, chip-level parameters)
BYTES_PER_CELL_UPDATE = 8  # 2 * sizeof(float) * n_subiterations (scripts/benchmark-common.jl:150-151)
COEF = [0.2, 0.2, 0.2, 0.2, 0.2]  # examples/jacobi/scripts/benchmark.jl:44-45


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=16384, help="grid columns, and rows per GPU unless --rows-per-gpu")
    ap.add_argument("--rows-per-gpu", type=int, default=0,
                    help="rows of every rank's strip (default: --size); BASELINE config 5 = --size 65536 "
                         "--rows-per-gpu 8192 on 8 GPUs")
    ap.add_argument("--generations", type=int, default=252, help="generations per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug-host-exchange", action="store_true",
                    help="debugging: all ranks on cuda:0, gloo process group, ghost rows through host memory")
    ap.add_argument("--strip-domain", action="store_true",
                    help="debugging: drive the sweep through the multi-GPU strip driver even with one rank")
    ap.add_argument("--single-strip", action="store_true",
                    help="one full-grid launch per pass (no row strips on side streams); used for profiling")
    ap.add_argument("--cpu-size", type=int, default=8192)
    ap.add_argument("--cpu-generations", type=int, default=192)
    return ap.parse_args()


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


def cpu_baseline(size, generations):
    """The oracle (a port of the reference's cpu backend) on the host cores, bounded sample."""
    import numpy as np

    from oracle import oracle as O

    cores = usable_cores()
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
        "sample": f"Jacobi5General {size}x{size}, {generations} generations, OpenMP over rows, {dt:.1f} s",
    }


def main():
    args = parse_args()
    if args.single_strip:
        os.environ["STSTHIP_VIRTUAL_STRIPS"] = "1"
    import numpy as np
    import torch

    from stencilstream_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if args.debug_host_exchange:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    capi.init(local_rank)
    stream = torch.cuda.Stream(device)  # the stream every sweep is launched on (and timed on)

    H, W, gens = (args.rows_per_gpu or args.size), args.size, args.generations
    p = capi.JacobiParams()
    for i, c in enumerate(COEF):
        p.coef[i] = c
    halo = np.float32(0.0).tobytes()
    app = os.environ.get("STSTHIP_BENCH_APP", "jacobi5general")  # tuning experiments only
    # five equal positive coefficients and a +0 halo: the runtime uses the bit-identical product-carrying
    # form of the kernel (Jacobi5Uniform, 5 instead of 9 flops per cell, 12 generations per launch)
    uniform = app == "jacobi5general" and len(set(COEF)) == 1 and COEF[0] > 0 and \
        os.environ.get("STSTHIP_JACOBI_FASTPATH", "1") != "0"
    info = capi.app_info("jacobi5uniform" if uniform else app)

    if world == 1 and not args.strip_domain:
        src = init_grid_device(torch, H, W, 0, H, device)
        dst = torch.empty_like(src)
        dom = capi.Domain(H, W, 0, H, W)
        torch.cuda.synchronize()  # the grid is resident before anything runs on `stream`

        def step():
            return capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens,
                                blocking=False, profiling=False, stream=stream.cuda_stream)

        barrier = lambda: None
        total_rows = H
    else:
        import torch.distributed as dist

        from stencilstream_amd.dist import StripDomain

        if world == 1:
            pass  # --strip-domain: no process group, no neighbours
        elif args.debug_host_exchange:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        total_rows = H * world
        strip = StripDomain(app, p, halo, total_rows, W, rank, world, device, np.dtype("<f4"),
                            exchange_via_host=args.debug_host_exchange)
        strip.load_owned(init_grid_device(torch, H, W, rank * H, total_rows, device))

        def step():
            strip.advance(0, gens)

        barrier = dist.barrier if world > 1 else (lambda: None)

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

    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], device="cpu" if args.debug_host_exchange else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        cells = total_rows * W * gens * args.steps
        value = cells / elapsed / 1e9
        out = {
            "metric": "Gcell-updates/s, Jacobi-5pt fp32 16384^2",
            "value": value,
            "unit": "Gcell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"Jacobi5General fp32 {total_rows}x{W}, {gens} generations per step, halo 0, coef 5x0.2",
                "kernel_form": "Jacobi5Uniform (bit-identical product-carrying form)" if uniform else "Jacobi5General",
                "generations_total": gens * args.steps,
                "temporal_blocking": int(info.max_generations),
                "decomposition": "none" if world == 1 else f"{world} row strips of {H} rows, RCCL ghost rows",
            },
        }
        if world == 1 and not args.strip_domain:
            # The dominant (only) kernel, measured live: full-grid launches (one row strip) of T
            # generations, HIP events around every launch on the launch stream.
            saved = os.environ.get("STSTHIP_VIRTUAL_STRIPS")
            os.environ["STSTHIP_VIRTUAL_STRIPS"] = "1"
            depth = int(info.max_generations)
            gens_timed = max(depth, gens // depth * depth)  # launches of the full depth only
            try:
                prof = capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens_timed,
                                    blocking=True, profiling=True, stream=stream.cuda_stream)
            finally:
                if saved is None:
                    del os.environ["STSTHIP_VIRTUAL_STRIPS"]
                else:
                    os.environ["STSTHIP_VIRTUAL_STRIPS"] = saved
            kernel_ms = prof.kernel_time_s * 1e3 / max(prof.n_launches, 1)
            T = gens_timed / max(prof.n_launches, 1)
            alg_bytes = H * W * BYTES_PER_CELL_UPDATE * T
            achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
            out["roofline"] = {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "kernel": ("sweep_kernel<Sweep<Jacobi5Uniform, AoS, T=12, K=3, P=4>>" if uniform
                           else "sweep_kernel<Sweep<Jacobi<General5>, AoS, T=8, K=4, P=4>>"),
                "kernel_ms": kernel_ms,
                "launches_timed": int(prof.n_launches),
                "generations_per_launch": T,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "full-grid launches; `value` additionally overlaps two row strips on two streams",
            }
            traffic_file = os.path.join(ROOT, "profiles", "traffic_r01.json")
            if os.path.exists(traffic_file):
                try:
                    out["roofline"]["traffic"] = json.load(open(traffic_file)).get("hbm_bytes_per_launch")
                except Exception:
                    pass
            # the general-coefficient kernel (nine flops per cell), same grid, coefficients that differ
            extras = os.environ.get("STSTHIP_BENCH_MINIMAL", "0") == "0"  # profiling runs skip the extra legs
            if uniform and extras:
                q = capi.JacobiParams()
                for i, c in enumerate([0.2, 0.21, 0.19, 0.22, 0.18]):
                    q.coef[i] = c
                capi.app_run(app, q, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                             stream=stream.cuda_stream)
                t1 = time.perf_counter()
                capi.app_run(app, q, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                             stream=stream.cuda_stream)
                out["general_coefficients"] = {"value": H * W * gens / (time.perf_counter() - t1) / 1e9,
                                               "unit": "Gcell-updates/s",
                                               "note": "Jacobi5General kernel, coefficients 0.2 0.21 0.19 0.22 0.18"}
            # the same kernel with fused multiply-adds (not bit-identical to the reference's cpu backend;
            # reported for information only, never as `value`)
            try:
                if not extras:
                    raise capi.StsthipError(0, "skipped", "")
                capi.app_info(app + "_fma")
                capi.app_run(app + "_fma", p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                             stream=stream.cuda_stream)
                t1 = time.perf_counter()
                capi.app_run(app + "_fma", p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                             stream=stream.cuda_stream)
                dt = time.perf_counter() - t1
                out["fma_flavour"] = {"value": H * W * gens / dt / 1e9, "unit": "Gcell-updates/s",
                                      "note": "-ffp-contract=fast build of the same kernel; tolerance parity only"}
            except capi.StsthipError:
                pass
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.cpu_size, args.cpu_generations)
        print(json.dumps(out), flush=True)

    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
