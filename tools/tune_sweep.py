#!/usr/bin/env python3
"""Sweep the launch-shape knobs of the Jacobi sweep on one GPU and print Gcell-updates/s.
Knobs are environment variables read by the runtime at every launch:
  STSTHIP_MAX_GENERATIONS (temporal blocking depth cap), STSTHIP_TAIL_PERMILLE, STSTHIP_CHUNK_ROWS."""
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stencilstream_amd import capi


def main():
    app = sys.argv[1] if len(sys.argv) > 1 else "jacobi5general"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    gens = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    capi.init(0)
    info = capi.app_info(app)
    p = capi.JacobiParams()
    for i in range(9):
        p.coef[i] = 0.2 if i < 5 else 0.1
    src = torch.rand(size, size, device="cuda")
    dst = torch.empty_like(src)
    dom = capi.Domain(size, size, 0, size, size)
    halo = np.float32(0).tobytes()
    side = torch.cuda.Stream()
    stream = side.cuda_stream
    torch.cuda.synchronize()
    depths = [t for t in (1, 2, 4, 8) if t <= info.max_generations]
    for T, wpc in itertools.product(depths, (125, 250, 500, 1000, 2000)):
        os.environ["STSTHIP_MAX_GENERATIONS"] = str(T)
        os.environ["STSTHIP_TAIL_PERMILLE"] = str(wpc)
        capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True, stream=stream)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True, stream=stream)
            best = min(best, time.perf_counter() - t0)
        print(f"{app} {size}^2 T={T} tail permille={wpc}: {size * size * gens / best / 1e9:8.1f} Gcell/s "
              f"({best / (gens / T) * 1e3:.3f} ms/launch)", flush=True)


if __name__ == "__main__":
    main()
