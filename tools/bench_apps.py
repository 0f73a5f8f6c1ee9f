#!/usr/bin/env python3
"""Throughput of every BASELINE.json GPU configuration on one MI355X, through the C ABI:

  jacobi   Jacobi5General fp32 16384^2               (configs[1], also bench.py)
  hotspot  HotSpot 2 x fp32 8192^2, per-field planes  (configs[2]; the reference is fp32)
  fdtd     FDTD coef resolver 4608^2, 2 sub-iterations (configs[3] grid of max_grid.json)
  conway   Game of Life 16384^2, 1 byte per cell

Prints one JSON line per app: Gcell-updates/s (sub-iterations not counted), ms per launch, and the
algorithmic HBM rate 2*sizeof(Cell)*n_sub bytes per cell-update against the 8 TB/s roofline.
Inputs are synthetic and resident in HBM; planes are used directly (no scatter/gather in the loop)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stencilstream_amd import capi


def hotspot_params(n):
    # examples/hotspot/hotspot.cpp:281-295 in numpy float32/float64 with the same expression types
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


def dims(rows, cols):
    """Grid of a configuration; BENCH_APPS_ROWS / BENCH_APPS_COLS override it (strip-rule sweeps)."""
    return int(os.environ.get("BENCH_APPS_ROWS", rows)), int(os.environ.get("BENCH_APPS_COLS", cols))


def run(app, params, halo, planes_a, planes_b, H, W, gens, stream, reps=3):
    dom = capi.Domain(H, W, 0, H, W)
    a = [t.data_ptr() for t in planes_a]
    b = [t.data_ptr() for t in planes_b]
    capi.app_run(app, params, halo, dom, a, b, 0, gens, blocking=True, stream=stream.cuda_stream)
    best, launches = 1e9, 1
    for _ in range(reps):
        t0 = time.perf_counter()
        info = capi.app_run(app, params, halo, dom, a, b, 0, gens, blocking=True, stream=stream.cuda_stream)
        best = min(best, time.perf_counter() - t0)
        launches = info.n_launches
    return best, launches


def main():
    which = sys.argv[1:] or ["jacobi", "hotspot", "hotspot_aos", "hotspot_f64", "hotspot_f64_aos", "fdtd", "fdtd_aos",
                             "fdtd_grouped", "conway"]
    if which == ["experiments"]:
        which = sorted(a for a in capi.list_apps() if a.startswith(("x_hs_", "x_fd_", "x_cw_")))
        if os.environ.get("AB_ONLY"):
            which = [a for a in which if any(a.startswith(p) for p in os.environ["AB_ONLY"].split(","))]
    capi.init(0)
    dev = "cuda"
    stream = torch.cuda.Stream()
    out = []
    for name in which:
        if name == "jacobi":
            app, (H, W), gens = "jacobi5general", dims(16384, 16384), 240
            p = capi.JacobiParams()
            for i in range(5):
                p.coef[i] = 0.2
            halo = np.float32(0).tobytes()
            pa, pb = [torch.rand(H, W, device=dev)], [torch.empty(H, W, device=dev)]
        elif name in ("jacobi_general", "jacobi_general_independent"):
            app, (H, W), gens = ("jacobi5general" if name == "jacobi_general" else "jacobi5general_independent"), dims(16384, 16384), 240
            p = capi.JacobiParams()
            for i, c in enumerate([0.2, 0.21, 0.19, 0.22, 0.18]):
                p.coef[i] = c
            halo = np.float32(0).tobytes()
            pa, pb = [torch.rand(H, W, device=dev)], [torch.empty(H, W, device=dev)]
        elif name in ("hotspot", "hotspot_aos"):
            app, (H, W), gens = name, dims(8192, 8192), 200
            p = hotspot_params(H)
            halo = np.zeros(2, np.float32).tobytes()
            if name == "hotspot":
                temp = torch.full((H, W), 30.0, device=dev)
                power = torch.zeros(H, W, device=dev)
                power[H // 4 - 1:3 * H // 4, W // 4 - 1:3 * W // 4] = 0.5
                pa, pb = [temp, power], [torch.empty_like(temp), torch.empty_like(power)]
            else:
                cells = torch.zeros(H, W, 2, device=dev)
                cells[..., 0] = 30.0
                pa, pb = [cells], [torch.empty_like(cells)]
        elif name in ("hotspot_f64", "hotspot_f64_aos") or name.startswith("x_h64_"):
            app, (H, W), gens = name, dims(8192, 8192), (240 if name.startswith("x_h64_") else 200)
            p32 = hotspot_params(H)
            p = capi.HotspotParamsF64(p32.Rx_1, p32.Ry_1, p32.Rz_1, p32.Cap_1)
            halo = np.zeros(2, np.float64).tobytes()
            if name == "hotspot_f64" or "_soa_" in name:
                temp = torch.full((H, W), 30.0, device=dev, dtype=torch.float64)
                power = torch.zeros(H, W, device=dev, dtype=torch.float64)
                power[H // 4 - 1:3 * H // 4, W // 4 - 1:3 * W // 4] = 0.5
                pa, pb = [temp, power], [torch.empty_like(temp), torch.empty_like(power)]
            else:
                cells = torch.zeros(H, W, 2, device=dev, dtype=torch.float64)
                cells[..., 0] = 30.0
                pa, pb = [cells], [torch.empty_like(cells)]
        elif name in ("fdtd", "fdtd_aos", "fdtd_grouped"):
            app, (H, W), gens = {"fdtd": "fdtd_coef", "fdtd_aos": "fdtd_coef_aos", "fdtd_grouped": "fdtd_coef_grouped"}[name], dims(4608, 4608), 120
            p = capi.FdtdParams(dt=8.1e-19, t_0=3e-13, tau=1e-13, omega=7.5e14, cutoff_iteration=10 ** 9,
                                detect_iteration=0, source_radius_squared=100.0, source_r=H / 2, source_c=W / 2,
                                source_distance_bound=100.0 - 2 * (H / 2) ** 2, double_center_rc=float(H))
            halo = np.zeros(8, np.float32).tobytes()
            if name == "fdtd":
                pa = [torch.rand(H, W, device=dev) * 1e-3 for _ in range(4)] + [torch.full((H, W), v, device=dev) for v in (1.0, 0.3, 1.0, 0.3)]
                pb = [torch.empty(H, W, device=dev) for _ in range(8)]
            elif name == "fdtd_grouped":
                material = torch.empty(H, W, 4, device=dev)
                for i, v in enumerate((1.0, 0.3, 1.0, 0.3)):
                    material[..., i] = v
                pa = [torch.rand(H, W, 4, device=dev) * 1e-3, material]
                pb = [torch.empty_like(t) for t in pa]
            else:
                cells = torch.rand(H, W, 8, device=dev) * 1e-3
                pa, pb = [cells], [torch.empty_like(cells)]
        elif name == "conway":
            app, (H, W), gens = "conway", dims(16384, 16384), 200
            p = capi.NoParams()
            halo = b"\0"
            pa = [(torch.rand(H, W, device=dev) < 0.35).to(torch.uint8)]
            pb = [torch.empty_like(pa[0])]
        elif name.startswith(("x_hs_", "x_fd_", "x_cw_")):
            # experiments: buffers are built from the registry's own description of the app, so a
            # name can never be paired with the wrong layout
            app = name
            meta = capi.app_info(app)
            if name.startswith("x_hs_"):
                H, W, gens, p, fill = 8192, 8192, 240, hotspot_params(8192), [30.0, 0.25]
            elif name.startswith("x_fd_"):
                H, W, gens, fill = 4608, 4608, 840, [1e-4, 2e-4, 3e-4, 0.0, 1.0, 0.3, 1.0, 0.3]
                p = capi.FdtdParams(dt=8.1e-19, t_0=3e-13, tau=1e-13, omega=7.5e14, cutoff_iteration=10 ** 9,
                                    detect_iteration=0, source_radius_squared=100.0, source_r=H / 2, source_c=W / 2,
                                    source_distance_bound=100.0 - 2 * (H / 2) ** 2, double_center_rc=float(H))
            else:
                # packed Game of Life: a 16384^2 grid of cells = 16384 x 4096 words of four cells
                H, W, gens, p, fill = 16384, 4096, 192, capi.NoParams(), None
            halo = bytes(meta.cell_size)
            # seeded random fields (constants where the application has constants): every cell evolves differently,
            # so the checksum below tells two shapes' results apart
            gen = torch.Generator(device=dev).manual_seed(1234)

            def field(i, v):
                if name.startswith("x_hs_"):
                    return (320.0 + 10.0 * torch.rand(H, W, device=dev, generator=gen)) if i == 0 else \
                        0.01 * torch.rand(H, W, device=dev, generator=gen)
                if i < 4:
                    return 1e-3 * torch.rand(H, W, device=dev, generator=gen)
                return torch.full((H, W), v, device=dev)

            if fill is None:
                pa = [(torch.rand(H, 4 * W, device=dev, generator=gen) < 0.35).to(torch.uint8).view(torch.int32)]
            elif "_grp_" in name or (meta.n_planes == 2 and meta.plane_elem_size[0] == 16):  # FdtdGrouped: two planes of 16-byte halves {ex, ey, hz, hz_sum} / {ca, cb, da, db}
                pa = [torch.stack([field(i, fill[i]) for i in range(4 * h, 4 * h + 4)], dim=-1).contiguous() for h in range(2)]
            elif meta.n_planes == 1:
                pa = [torch.stack([field(i, v) for i, v in enumerate(fill)], dim=-1).contiguous()]
                assert pa[0].element_size() * len(fill) == meta.cell_size
            else:
                assert meta.n_planes == len(fill) and all(meta.plane_elem_size[i] == 4 for i in range(len(fill)))
                pa = [field(i, v) for i, v in enumerate(fill)]
            pb = [torch.empty_like(t) for t in pa]
        else:
            raise SystemExit(f"unknown app {name}")
        torch.cuda.synchronize()
        info = capi.app_info(app)
        best, launches = run(app, p, halo, pa, pb, H, W, gens, stream)
        bytes_per_update = 2 * info.cell_size * info.n_subiterations
        if name.startswith("x_cw_"):
            bytes_per_update = 2  # one byte per cell of the game, four cells per word
        cells_per_elem = 4 if name.startswith("x_cw_") else 1
        gcells = H * W * cells_per_elem * gens / best / 1e9
        # results of the same function, layout and generation count must agree bit for bit whatever the shape
        checksum = sum(int(t.view(torch.int32).to(torch.int64).sum().item()) for t in pb) & 0xFFFFFFFFFFFF
        line = {"app": app, "grid": [H, W], "generations": gens, "Gcell_updates_per_s": round(gcells, 1),
                "stages": int(info.stages), "K": int(info.cells_per_lane), "T": int(info.max_generations), "checksum": checksum,
                "ms_per_launch": round(best / launches * 1e3, 4), "generations_per_launch": gens / launches,
                "algorithmic_bytes_per_cell_update": bytes_per_update,
                "algorithmic_GBps": round(gcells * bytes_per_update, 1),
                "frac_of_8TBps": round(gcells * bytes_per_update / 8000.0, 3)}
        print(json.dumps(line), flush=True)
        out.append(line)
        del pa, pb
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
