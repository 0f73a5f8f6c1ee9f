#!/usr/bin/env python3
"""Staged sweeps (EXPERIMENTS=1 build, tools/experiments/app_experiments12/13.hip) against the independent-wave
shapes: Gcell-updates/s on the full grid and on thin row strips, and bit-equality of every shape's result with the
baseline's (the same generations of the same function, so the bits must agree whatever the shape).

usage: tools/ab_staged.py <family> [rows x cols ...]    family: uniform | general
       STSTHIP_VIRTUAL_STRIPS=1 for single launches per pass"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stencilstream_amd import capi


def main():
    family = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]] or [(16384, 16384), (4096, 16384), (2048, 16384)]
    capi.init(0)
    p = capi.JacobiParams()
    coef = [0.2] * 5 if family == "uniform" else [0.2, 0.21, 0.19, 0.22, 0.18]
    for i, c in enumerate(coef):
        p.coef[i] = c
    prefix, base = ("x_ju_", "x_ju_k3t12s1") if family == "uniform" else ("x_j5_", "jacobi5general")
    names = [base] + [a for a in capi.list_apps() if a.startswith(prefix) and a != base and "s" in a.split("_")[-1]]
    if os.environ.get("AB_ONLY"):  # profiling runs: these apps only, no baseline comparison
        names = os.environ["AB_ONLY"].split(",")
    if family == "uniform":
        # Jacobi5Uniform's parameter block is the single coefficient
        class Block(capi.C.Structure):
            _fields_ = [("c", capi.C.c_float)]
        q = Block(0.2)
    halo = np.float32(0).tobytes()
    stream = torch.cuda.Stream()
    gens = int(os.environ.get("AB_GENS", "240"))  # 960 with AB_STEPS=5: bench.py's regime (long calls queued back to back)
    steps = int(os.environ.get("AB_STEPS", "1"))
    for H, W in shapes:
        src = torch.rand(H, W, device="cuda")
        dom = capi.Domain(H, W, 0, H, W)
        ref = None
        for app in names:
            dst = torch.zeros_like(src)
            params = q if (family == "uniform") else p
            if app == "jacobi5general":
                os.environ["STSTHIP_JACOBI_FASTPATH"] = "0"
                capi.reload_options()
            try:
                capi.app_run(app, params, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                             stream=stream.cuda_stream)
            except Exception as e:  # noqa: BLE001
                print(f"{H}x{W} {app:20s} FAILED {e}", flush=True)
                continue
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _step in range(steps):
                    info = capi.app_run(app, params, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens,
                                        blocking=(_step == steps - 1), stream=stream.cuda_stream)
                best = min(best, (time.perf_counter() - t0) / steps)
            torch.cuda.synchronize()
            if ref is None:
                ref = dst.clone()
                same = "baseline"
            else:
                same = "bit-identical" if torch.equal(ref.view(torch.int32), dst.view(torch.int32)) else "DIFFERS"
            i = capi.app_info(app)
            print(f"{H}x{W} {app:20s} K={i.cells_per_lane} T={i.max_generations} P={i.prefetch_rows} stages={i.stages} "
                  f"{H * W * gens / best / 1e9:8.1f} Gcell/s  launches={info.n_launches}  {same}", flush=True)
        del src, ref


if __name__ == "__main__":
    main()
