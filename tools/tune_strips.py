#!/usr/bin/env python3
"""Effect of STSTHIP_VIRTUAL_STRIPS (row strips advancing on separate streams inside one GPU)."""
import os
import statistics
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
    gens = int(sys.argv[3]) if len(sys.argv) > 3 else 252
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
    options = [(1, 500), (2, 500), (2, 400), (2, 333), (2, 250), (2, 600), (2, 700), (3, 0), (4, 0)]
    times = {v: [] for v in options}
    for rnd in range(5):
        for v in options:
            os.environ["STSTHIP_VIRTUAL_STRIPS"] = str(v[0])
            os.environ["STSTHIP_STRIP_SKEW_PERMILLE"] = str(v[1])
            t0 = time.perf_counter()
            capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True, stream=side.cuda_stream)
            if rnd:
                times[v].append(time.perf_counter() - t0)
    for v in options:
        print(f"{app} {size}^2 virtual strips={v}: median {size * size * gens / statistics.median(times[v]) / 1e9:8.1f} "
              f"best {size * size * gens / min(times[v]) / 1e9:8.1f} Gcell/s", flush=True)


if __name__ == "__main__":
    main()
