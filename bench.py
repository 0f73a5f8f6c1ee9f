#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json configs[1] -- Jacobi 5-point (Jacobi5General) fp32, 16384 x 16384,
1000 generations, grid resident in HBM.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  A "step" is ONE StencilUpdate call of `--generations`
generations (default 1000, BASELINE's count: 83 launches of 12 generations + one of 3 + one of 1).
Rank 0 prints ONE JSON line.

Metric: Gcell-updates/s = H*W*generations / time (sub-iterations not counted), the reference's definition
(scripts/benchmark-common.jl:97-98,122).

N > 1 is BASELINE's series (metric "16384^2, 1/2/4/8 GPU"): STRONG scaling -- the
same 16384^2 grid cut into N row strips, ghost rows exchanged with the two neighbours over RCCL (no collective
on the data path).  `--config5` runs the 65536^2 grid of configs[4] instead (also strong: 65536/N rows per rank;
its 1-GPU point fits one MI355X).  `--rows-per-gpu R` fixes the rows per rank instead (weak scaling).

What the JSON line carries besides the contract's fields:
  roofline      the dominant kernel measured live with HIP events on the launch stream (full-grid launches of
                the deepest compiled blocking depth).  `achieved` / `frac` are SURVEY 8d's ALGORITHMIC bytes
                (8 B per cell-update x generations per launch) over the launch time -- with temporal blocking
                this exceeds the 8 TB/s peak by design (each cell moves once per 12 generations), so the line
                also carries the two physical fractions: HBM bytes from the PMC counters over the launch time
                against 8 TB/s, and VALU wave-instructions against the measured issue peak.  `bound` names the
                larger of the two.  Counter values come from profiles/r02_bench_counters.json (rocprofv3
                --pmc passes over this very command, tools/profile_bench.sh) and are only attached when the
                kernel, grid and depth of this run match the profiled ones; null otherwise.
  verified      after the timed loop the result of the last step is checked (see verify()).
  cpu_baseline  the CPU port timed on this box's host cores, bounded sample.
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
# what a plain device-to-device copy of a 1 GiB buffer reaches on this hardware (read + write bytes over time;
# tools/microbench/hbm_pattern.hip, profiles/r02_hbm_pattern.txt: 4.9-5.5 TB/s): the attainable ceiling
HBM_COPY_GBS = 5400.0
BYTES_PER_CELL_UPDATE = 8  # 2 * sizeof(float) * n_subiterations (scripts/benchmark-common.jl:150-151)
COEF = [0.2, 0.2, 0.2, 0.2, 0.2]  # examples/jacobi/scripts/benchmark.jl:44-45
# VALU issue peak: plain fp32 wave-instructions per second and SIMD, measured on MI355X with every SIMD busy
# (tools/microbench/valu_rate.hip, profiles/r01_microbench_valu_rate.txt: 1.09 ns per instruction and SIMD)
VALU_NS_PER_WAVE_INSTRUCTION = 1.09
N_SIMDS = 1024
COUNTER_FILE = os.path.join(ROOT, "profiles", "r02_bench_counters.json")


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--debug-host-exchange", action="store_true",
                    help="debugging: all ranks on cuda:0, gloo process group, ghost rows through host memory")
    ap.add_argument("--strip-domain", action="store_true",
                    help="debugging: drive the sweep through the multi-GPU strip driver even with one rank")
    ap.add_argument("--single-strip", action="store_true",
                    help="one full-grid launch per pass (no row strips on side streams)")
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
        "implementation": "oracle/stencil_oracle.c: plain-C restatement of the reference's cpu backend "
                          "(cpu/StencilUpdate.hpp:109-142,185-223: double buffer, halo test per neighbour, "
                          "transition function through a pointer), rows shared among threads with OpenMP.  The "
                          "reference's own cpu backend needs a SYCL compiler (icpx), absent on this box; this "
                          "repository's stencil::cpu template backend is a C++ binary per application and is "
                          "timed by tools/benchmark.py instead (profiles/r02_harness)",
        "sample": f"Jacobi5General {size}x{size}, {generations} generations, {dt:.1f} s",
    }


def uniform_field_value(n):
    """Value after n generations of a cell whose whole dependency cone holds 1.0f (five equal coefficients):
    the scalar recurrence of examples/jacobi/kernels.hpp:267-271, every operation rounded to fp32."""
    import numpy as np

    c, v = np.float32(COEF[0]), np.float32(1.0)
    for _ in range(n):
        p = c * v
        v = np.float32(np.float32(np.float32(np.float32(p + p) + p) + p) + p)
    return v


def verify(torch, capi, app, p, halo, dom, src, dst, gens, stream, total_rows, W):
    """Checks of what the timed loop produced (dst = `gens` generations of src, centred-square input):
    (1) bulk, exact: every cell further than `gens` cells inside the square equals the scalar recurrence, every
        cell further than `gens` cells outside it (and its halo of zeros) is +0 -- the bit patterns are compared
        on the device over those two regions (most of the grid at 16384^2, 1000 generations);
    (2) windows vs the oracle: a second, short run of the same call (36 generations = first / middle / last launch
        forms, same row strips and streams) compared bit for bit with the CPU oracle on 10 windows of 64 x 64 cells
        (grid corners, edges, rim of the square, the boundary of the two row strips), each computed from the input
        patch widened by 36 cells."""
    import numpy as np

    from oracle import oracle as O

    H = total_rows
    report = {"bulk_cells_checked": 0, "windows": 0}
    ok = True
    lo_r, hi_r, lo_c, hi_c = int(np.ceil(H * 0.25)), int(np.ceil(H * 0.75)), int(np.ceil(W * 0.25)), int(np.ceil(W * 0.75))
    if hi_r - lo_r > 2 * gens + 2 and hi_c - lo_c > 2 * gens + 2:
        inner = dst[lo_r + gens:hi_r - gens, lo_c + gens:hi_c - gens]
        want = float(uniform_field_value(gens))
        ok = ok and bool((inner == want).all().item())
        report["bulk_cells_checked"] += inner.numel()
        report["bulk_value_inside_square"] = want
    if lo_r - gens > 0:
        for outer in (dst[:lo_r - gens], dst[hi_r + gens:], dst[:, :lo_c - gens], dst[:, hi_c + gens:]):
            ok = ok and bool((outer.view(torch.int32) == 0).all().item())
            report["bulk_cells_checked"] += outer.numel()
    n, w = 36, 64
    short = torch.empty_like(dst)
    capi.app_run(app, p, halo, dom, [src.data_ptr()], [short.data_ptr()], 0, n, blocking=True,
                 stream=stream.cuda_stream)
    spots = [(0, 0), (0, W - w), (H - w, 0), (H - w, W - w), (lo_r - 30, lo_c - 30), (hi_r - 30, hi_c - 30),
             (H // 2, 0), (0, W // 3), (H * 2 // 5 - 30, W // 2), (lo_r - 30, W // 2)]
    for r0, c0 in spots:
        r0, c0 = max(0, min(r0, H - w)), max(0, min(c0, W - w))
        ra, rb, ca, cb = max(0, r0 - n), min(H, r0 + w + n), max(0, c0 - n), min(W, c0 + w + n)
        ref = O.jacobi("Jacobi5General", COEF, src[ra:rb, ca:cb].cpu().numpy(), n, halo=0.0)
        got = short[r0:r0 + w, c0:c0 + w].cpu().numpy()
        want = ref[r0 - ra:r0 - ra + w, c0 - ca:c0 - ca + w]
        ok = ok and np.array_equal(got.view(np.uint32), want.view(np.uint32))
        report["windows"] += 1
    report["window_generations"] = n
    return ok, report


def attach_counters(roofline, kernel_key, kernel_ms):
    """HBM bytes and VALU instructions per launch from the committed PMC passes of this command."""
    roofline["traffic"] = None
    roofline["fractions"] = {"algorithmic_hbm": roofline["frac"], "physical_hbm": None, "valu_issue": None}
    try:
        counters = json.load(open(COUNTER_FILE))["kernels"].get(kernel_key)
    except (OSError, ValueError, KeyError):
        counters = None
    if not counters:
        roofline["counters"] = f"no PMC profile of {kernel_key} in profiles/r02_bench_counters.json"
        return
    traffic = counters["hbm_bytes_per_launch"]
    roofline["traffic"] = traffic
    roofline["fractions"]["physical_hbm"] = traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    roofline["fractions"]["physical_hbm_of_measured_copy_rate"] = traffic / (kernel_ms * 1e-3) / 1e9 / HBM_COPY_GBS
    roofline["measured_copy_rate_GBps"] = HBM_COPY_GBS
    valu = counters.get("valu_wave_instructions_per_launch")
    if valu:
        busy_ms = valu * VALU_NS_PER_WAVE_INSTRUCTION * 1e-6 / N_SIMDS
        roofline["fractions"]["valu_issue"] = busy_ms / kernel_ms
        roofline["valu_wave_instructions_per_launch"] = valu
    roofline["counters"] = counters.get("source")
    f = roofline["fractions"]
    if f["valu_issue"] is not None:
        roofline["bound"] = "valu" if f["valu_issue"] >= f["physical_hbm"] else "hbm"


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

    W, gens = (65536 if args.config5 else args.size), args.generations
    if args.rows_per_gpu:
        total_rows, scaling = args.rows_per_gpu * world, "weak"
    else:
        total_rows, scaling = W, "strong"
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
    ranks_report = None

    if world == 1 and not args.strip_domain:
        H = total_rows
        src = init_grid_device(torch, H, W, 0, H, device)
        dst = torch.empty_like(src)
        dom = capi.Domain(H, W, 0, H, W)
        torch.cuda.synchronize()  # the grid is resident before anything runs on `stream`

        def step():
            return capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens,
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
        # C++, no Python between the launches of a step).  If its communicator cannot be created on this node the
        # run falls back -- on every rank alike -- to the same algorithm driven from Python over torch.distributed's
        # RCCL point-to-point operations (stencilstream_amd/dist.py); the JSON line says which one ran.
        native = not args.debug_host_exchange and os.environ.get("STSTHIP_BENCH_EXCHANGE", "native") == "native"
        comm, why = None, None
        if native and world > 1:
            try:
                uid = [capi.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                comm = capi.comm_create(uid[0], rank, world)
            except Exception as e:  # noqa: BLE001
                why = f"{type(e).__name__}: {e}"
            flag = torch.tensor([0 if comm is None else 1], device=device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native = bool(flag.item())
        a, b = split_rows(total_rows, world)[rank]
        init = init_grid_device(torch, b - a, W, a, total_rows, device)
        if native:
            strip = capi.Strip(app, p, halo, total_rows, W, rank, world, comm=comm)
            assert (strip.row_begin, strip.row_end) == (a, b)
            torch.cuda.synchronize()
            strip.upload_from_device(0, init.data_ptr(), init.numel() * init.element_size())
            strip.synchronize()
            exchange = "RCCL send/recv issued by libststhip.so (ststhip_strip_advance)"

            def step():
                strip.advance(0, gens)
        else:
            strip = StripDomain(app, p, halo, total_rows, W, rank, world, device, np.dtype("<f4"),
                                exchange_via_host=args.debug_host_exchange)
            strip.load_owned(init)
            exchange = "torch.distributed batch_isend_irecv (stencilstream_amd/dist.py)" + (f"; native: {why}" if why else "")

            def step():
                strip.advance(0, gens)
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
                            "ghost_exchange": exchange, "ranks": gathered}

        barrier = dist.barrier if world > 1 else (lambda: None)
        decomposition = f"{world} row strips of {total_rows // world} rows, RCCL ghost rows once per launch; {exchange}"

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

    strips_verified = None
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], device="cpu" if args.debug_host_exchange else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if not args.no_verify:
            # every rank checks the bulk of its own rows of the last step's result: exact scalar recurrence further
            # than `gens` cells inside the square, +0 further than `gens` cells outside it (verify() has the argument)
            # (the strips' state has advanced through warm-up and timed steps; the check runs one more step from
            # the initial grid, outside the timed region)
            if native:
                strip.upload_from_device(0, init.data_ptr(), init.numel() * init.element_size())
                strip.advance(0, gens, blocking=True)
                rows = torch.from_numpy(strip.download(0, np.float32))
            else:
                strip.load_owned(init)
                strip.advance(0, gens)
                rows = strip.owned(0, torch.float32).cpu()
            r = torch.arange(a, b)[:, None]
            c = torch.arange(W)[None, :]
            lo_r, hi_r, lo_c, hi_c = total_rows // 4, 3 * total_rows // 4, W // 4, 3 * W // 4
            inside = (r >= lo_r + gens) & (r < hi_r - gens) & (c >= lo_c + gens) & (c < hi_c - gens)
            outside = (r < lo_r - gens) | (r >= hi_r + gens) | (c < lo_c - gens) | (c >= hi_c + gens)
            want = float(uniform_field_value(gens))
            ok = bool((rows[inside.expand_as(rows)] == want).all()) and \
                bool((rows.view(torch.int32)[outside.expand_as(rows)] == 0).all())
            flag = torch.tensor([1 if ok else 0], device="cpu" if args.debug_host_exchange else device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            strips_verified = bool(flag.item())

    if rank == 0:
        cells = total_rows * W * gens * args.steps
        value = cells / elapsed / 1e9
        form = "jacobi5uniform" if uniform else app
        out = {
            "metric": f"Gcell-updates/s, Jacobi-5pt fp32 {total_rows}x{W}, {world} GPU",
            "value": value,
            "unit": "Gcell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
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
                "decomposition": decomposition,
            },
            "hbm_roofline_gcells": HBM_PEAK_GBS / BYTES_PER_CELL_UPDATE,
            "frac_of_hbm_roofline": value * BYTES_PER_CELL_UPDATE / HBM_PEAK_GBS / world,
        }
        if ranks_report:
            out["ranks"] = ranks_report
        if strips_verified is not None:
            out["verified"] = strips_verified
            out["verification"] = {"what": "every rank: the bulk of its rows of the last step against the exact scalar "
                                           "recurrence inside the square and +0 outside it (all cells further than the "
                                           "generations per step from the rim of the square)"}
        if world == 1 and not args.strip_domain:
            if not args.no_verify:
                ok, report = verify(torch, capi, app, p, halo, dom, src, dst, gens, stream, total_rows, W)
                out["verified"] = bool(ok)
                out["verification"] = report
            # The dominant (only) kernel, measured live: full-grid launches (one row strip) of T
            # generations, HIP events around every launch on the launch stream.
            saved = os.environ.get("STSTHIP_VIRTUAL_STRIPS")
            os.environ["STSTHIP_VIRTUAL_STRIPS"] = "1"
            depth = int(info.max_generations)
            gens_timed = max(depth, min(gens, 240) // depth * depth)  # launches of the full depth only
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
                          f"K={int(info.cells_per_lane)}, P={int(info.prefetch_rows)}>>",
                "kernel_key": kernel_key,
                "kernel_ms": kernel_ms,
                "launches_timed": int(prof.n_launches),
                "generations_per_launch": T,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "achieved/frac: SURVEY 8d algorithmic bytes (8 B x cell-updates) over the launch time; it "
                        "exceeds 1 because one launch advances T generations while a cell moves through HBM once "
                        "(temporal blocking).  fractions.physical_hbm = PMC bytes / launch time / 8 TB/s; "
                        "fractions.valu_issue = VALU wave-instructions x 1.09 ns / 1024 SIMDs / launch time.  "
                        "Full-grid launches; `value` additionally overlaps two row strips on two streams",
            }
            attach_counters(out["roofline"], kernel_key, kernel_ms)
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
                out["general_coefficients"] = {"value": total_rows * W * gens / (time.perf_counter() - t1) / 1e9,
                                               "unit": "Gcell-updates/s",
                                               "note": "Jacobi5General kernel (what a C++ user functor gets), "
                                                       "coefficients 0.2 0.21 0.19 0.22 0.18"}
            if not args.no_cpu_baseline and extras:
                out["cpu_baseline"] = cpu_baseline(args.cpu_size, args.cpu_generations)
        print(json.dumps(out), flush=True)

    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
