#!/usr/bin/env python3
"""A/B of sweep shapes on one MI355X: every candidate app is run against a baseline app with the same
transition function on the same random grid -- results must be bit-identical (the baselines are the kernels the
parity tests check against the oracle) -- and both are timed.  Needs the EXPERIMENTS=1 library for x_* names.

usage: tools/ab_coop.py [family ...]     families: fdtd hotspot hotspot64 jacobi uniform"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from stencilstream_amd import capi

sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_apps import hotspot_params

FAMILIES = {
    # family: (grid, params, value ranges per field, baseline, candidates)
    "fdtd": (4608, "fdtd", ["fdtd_coef_aos", "fdtd_coef", "fdtd_coef_grouped"],
             ["x_fd_aos_k1t6_coop", "x_fd_soa_k1t6_coop", "x_fd_aos_k1t4_coop", "x_fd_aos_k1t8_coop",
              "x_fd_aos_k1t6_coop_nobarrier", "x_fd_aos_k1t6_coop_nolds", "x_fd_aos_k1t6_coop_neither",
              "x_fd_aos_k1t6p4", "x_fd_aos_k1t6p6", "x_fd_aos_k1t6p8", "x_fd_aos_k1t4p8", "x_fdg_k1t6_persist"]),
    "hotspot": (8192, "hotspot", ["hotspot", "hotspot_aos"],
                ["x_hs_soa_k1t8_coop", "x_hs_soa_k1t16_coop", "x_hs_soa_k2t8_coop", "x_hs_aos_k2t8_coop",
                 "x_hs_aos_k1t16_coop", "x_hs_soa_k1t8_persist", "x_hs_soa_k1t12p4", "x_hs_soa_k1t16p4"]),
    "hotspot64": (8192, "hotspot64", ["hotspot_f64", "hotspot_f64_aos"],
                  ["x_h64_soa_k1t8_coop", "x_h64_soa_k1t12_coop", "x_h64_soa_k1t16_coop", "x_h64_aos_k1t8_coop",
                   "x_h64_soa_k1t12p4", "x_h64_soa_k1t12p2", "x_h64_soa_k1t16p2", "x_h64_soa_k1t6p4",
                   "x_h64_soa_k2t8p2", "x_h64_soa_k2t8p4", "x_h64_soa_k2t6p2", "x_h64_soa_k2t4p4"]),
    "jacobi": (16384, "jacobi", ["jacobi5general"], ["x_j5_k4t8_coop", "x_j5_k2t8_coop", "x_j5_k2t16_coop",
                                                      "x_j5_k4t8_persist"]),
    "uniform": (16384, "uniform", ["x_ju_k3t12"], ["x_ju_k3t12_coop", "x_ju_k2t16_coop", "x_ju_k3t12_persist"]),
}


def make_params(kind, N):
    if kind == "fdtd":
        return capi.FdtdParams(dt=8.1e-19, t_0=3e-13, tau=1e-13, omega=7.5e14, cutoff_iteration=10 ** 9,
                               detect_iteration=0, source_radius_squared=100.0, source_r=N / 2, source_c=N / 2,
                               source_distance_bound=100.0 - 2 * (N / 2) ** 2, double_center_rc=float(N))
    if kind == "hotspot":
        return hotspot_params(N)
    if kind == "hotspot64":
        p = hotspot_params(N)
        return capi.HotspotParamsF64(p.Rx_1, p.Ry_1, p.Rz_1, p.Cap_1)
    if kind == "jacobi":
        p = capi.JacobiParams()
        for i, c in enumerate([0.2, 0.21, 0.19, 0.22, 0.18]):
            p.coef[i] = c
        return p
    return capi.JacobiUniformParams(0.2)


def fields(kind, N, dev, gen):
    """One tensor per field of the cell, seeded."""
    if kind == "fdtd":
        f = [(torch.rand(N, N, device=dev, generator=gen) - 0.5) * 1e-3 for _ in range(4)]
        return f + [torch.full((N, N), v, device=dev) for v in (1.0, 0.3, 1.0, 0.29)]
    if kind == "hotspot":
        return [30 + 8 * torch.rand(N, N, device=dev, generator=gen), 0.5 * torch.rand(N, N, device=dev, generator=gen)]
    if kind == "hotspot64":
        return [30 + 8 * torch.rand(N, N, device=dev, generator=gen, dtype=torch.float64),
                0.5 * torch.rand(N, N, device=dev, generator=gen, dtype=torch.float64)]
    return [torch.rand(N, N, device=dev, generator=gen)]


def buffers(app, field_list):
    info = capi.app_info(app)
    if info.n_planes == 1 and len(field_list) > 1:
        src = [torch.stack(field_list, dim=-1).contiguous()]
    elif info.n_planes != len(field_list):  # planes of several fields each (equal group sizes)
        per = len(field_list) // info.n_planes
        src = [torch.stack(field_list[g * per:(g + 1) * per], dim=-1).contiguous() for g in range(info.n_planes)]
    else:
        src = [f.clone() for f in field_list]
    return info, src, [torch.empty_like(t) for t in src]


def as_fields(info, planes, n_fields):
    if info.n_planes == 1 and n_fields > 1:
        return [planes[0][..., i] for i in range(n_fields)]
    if info.n_planes != n_fields:
        per = n_fields // info.n_planes
        return [planes[g][..., i] for g in range(info.n_planes) for i in range(per)]
    return planes


def main():
    which = sys.argv[1:] or list(FAMILIES)
    capi.init(0)
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream()
    available = set(capi.list_apps())
    for family in which:
        N, kind, baselines, candidates = FAMILIES[family]
        p = make_params(kind, N)
        gen = torch.Generator(device=dev).manual_seed(1234)
        field_list = fields(kind, N, dev, gen)
        cell_bytes = sum(f.element_size() for f in field_list)
        halo = bytes(cell_bytes)
        dom = capi.Domain(N, N, 0, N, N)
        torch.cuda.synchronize()
        reference = None
        only = [a for a in os.environ.get("AB_ONLY", "").split(",") if a]
        for app in baselines + candidates:
            if only and app not in only:
                continue
            if app not in available:
                print(json.dumps({"app": app, "skipped": "not in this library"}), flush=True)
                continue
            info, src, dst = buffers(app, field_list)
            torch.cuda.synchronize()  # the buffers were filled on torch's stream, the sweeps run on `stream`
            a, b = [t.data_ptr() for t in src], [t.data_ptr() for t in dst]
            n_check = 48
            capi.app_run(app, p, halo, dom, a, b, 0, n_check, blocking=True, stream=stream.cuda_stream)
            got = [t.clone() for t in as_fields(info, dst, len(field_list))]
            same = None
            if reference is None:
                reference = got
            else:
                same = all(torch.equal(x, y) for x, y in zip(got, reference))
            gens = 20 * int(info.max_generations)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                run = capi.app_run(app, p, halo, dom, a, b, 0, gens, blocking=True, stream=stream.cuda_stream)
                best = min(best, time.perf_counter() - t0)
            line = {"family": family, "app": app, "grid": N, "K": int(info.cells_per_lane), "T": int(info.max_generations),
                    "coop": int(info.cooperative), "bit_identical_to_baseline": same,
                    "Gcell_updates_per_s": round(N * N * gens / best / 1e9, 1),
                    "ms_per_launch": round(best / run.n_launches * 1e3, 4), "launches": int(run.n_launches)}
            print(json.dumps(line), flush=True)
            del src, dst, got
            torch.cuda.empty_cache()
        del reference, field_list
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
