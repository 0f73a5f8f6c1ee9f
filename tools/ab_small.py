#!/usr/bin/env python3
"""Small grids: Jacobi5General shapes (cells per lane, generations per launch) against the default on grids from
128^2 to 4096^2; Gcell-updates/s of 960 generations, best of 5.  Needs the EXPERIMENTS=1 library for x_* names."""
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
    apps = sys.argv[1:] or ["jacobi5general", "x_j5_k1t2", "x_j5_k1t4", "x_j5_k2t4", "x_j5_k1t8", "x_j5_k2t8"]
    capi.init(0)
    p = capi.JacobiParams()
    for i, c in enumerate([0.2, 0.21, 0.19, 0.22, 0.18]):
        p.coef[i] = c
    halo = np.float32(0).tobytes()
    stream = torch.cuda.Stream()
    have = set(capi.list_apps())
    for n in (128, 256, 512, 1024, 1536, 2048, 3072, 4096):
        src = torch.rand(n, n, device="cuda")
        dst = torch.empty_like(src)
        dom = capi.Domain(n, n, 0, n, n)
        torch.cuda.synchronize()
        row = {"grid": n}
        ref = None
        for app in apps:
            if app not in have:
                continue
            gens = 960
            capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, 48, blocking=True, stream=stream.cuda_stream)
            if ref is None:
                ref = dst.clone()
            else:
                assert torch.equal(ref, dst), (app, n)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                             stream=stream.cuda_stream)
                best = min(best, time.perf_counter() - t0)
            row[app] = round(n * n * gens / best / 1e9, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
