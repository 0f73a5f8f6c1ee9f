#!/usr/bin/env python3
"""Shapes of the dense stencils on 16384^2: Jacobi9General (3 x 3) and Jacobi25 (5 x 5, radius 2).  Every shape
must equal the registered default bit for bit; Gcell-updates/s of 20 launches, best of 3.  EXPERIMENTS=1 library."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stencilstream_amd import capi


def main():
    capi.init(0)
    N = int(os.environ.get("AB_SIZE", "16384"))
    have = capi.list_apps()
    p9 = capi.JacobiParams()
    for i in range(9):
        p9.coef[i] = (1.0 + 0.01 * i) / 9
    p25 = capi.Jacobi25Params()
    for i in range(25):
        p25.coef[i] = (1.0 + 0.01 * (i % 5)) / 25
    halo = np.float32(0).tobytes()
    stream = torch.cuda.Stream()
    src = torch.rand(N, N, device="cuda")
    dst = torch.empty_like(src)
    dom = capi.Domain(N, N, 0, N, N)
    torch.cuda.synchronize()
    for base, prefix, p in (("jacobi9general", "x_j9_", p9), ("jacobi25general", "x_j25_", p25)):
        ref = None
        for app in [base] + [a for a in have if a.startswith(prefix)]:
            info = capi.app_info(app)
            capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, 24, blocking=True, stream=stream.cuda_stream)
            same = None
            if ref is None:
                ref = dst.clone()
            else:
                same = bool(torch.equal(ref, dst))
            gens = 20 * int(info.max_generations)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                run = capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                                   stream=stream.cuda_stream)
                best = min(best, time.perf_counter() - t0)
            print(json.dumps({"app": app, "grid": N, "K": int(info.cells_per_lane), "T": int(info.max_generations),
                              "bit_identical_to_default": same, "Gcell_updates_per_s": round(N * N * gens / best / 1e9, 1),
                              "launches": int(run.n_launches)}), flush=True)


if __name__ == "__main__":
    main()
