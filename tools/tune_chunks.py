#!/usr/bin/env python3
"""Sweep STSTHIP_CHUNK_ROWS for one app on one GPU (interleaved rounds, median and best)."""
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
    gens = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    rows_list = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [64, 96, 112, 128, 139, 150, 160, 176, 192, 224, 256, 320, 384, 512]
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
    times = {r: [] for r in rows_list}
    for rnd in range(6):
        for r in rows_list:
            os.environ["STSTHIP_CHUNK_ROWS"] = str(r)
            t0 = time.perf_counter()
            capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True, stream=side.cuda_stream)
            if rnd > 0:
                times[r].append(time.perf_counter() - t0)
    for r in rows_list:
        med, best = statistics.median(times[r]), min(times[r])
        print(f"{app} chunk_rows={r:4d}: median {size * size * gens / med / 1e9:8.1f}  best {size * size * gens / best / 1e9:8.1f} Gcell/s", flush=True)


if __name__ == "__main__":
    main()
