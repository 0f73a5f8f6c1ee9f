#!/usr/bin/env python3
"""Time the registered pipeline-shape experiments (app_experiments.hip) side by side."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stencilstream_amd import capi


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    gens = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    prefix = sys.argv[3] if len(sys.argv) > 3 else "x_j"
    capi.init(0)
    p = capi.JacobiParams()
    for i in range(5):
        p.coef[i] = 0.2
    src = torch.rand(size, size, device="cuda")
    dst = torch.empty_like(src)
    dom = capi.Domain(size, size, 0, size, size)
    halo = np.float32(0).tobytes()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    names = ["jacobi5general"] + [a for a in capi.list_apps() if a.startswith(prefix)]
    for wpc in (500,):
        os.environ["STSTHIP_TAIL_PERMILLE"] = str(wpc)
        for app in names:
            capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True, stream=side.cuda_stream)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                             stream=side.cuda_stream)
                best = min(best, time.perf_counter() - t0)
            print(f"tail permille={wpc:3d} {app:18s} {size * size * gens / best / 1e9:8.1f} Gcell/s", flush=True)


if __name__ == "__main__":
    main()
