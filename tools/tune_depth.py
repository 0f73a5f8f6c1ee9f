#!/usr/bin/env python3
"""Gcell-updates/s of one app for each temporal-blocking depth cap and several grid sizes."""
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
    sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4096, 8192, 16384, 32768]
    capi.init(0)
    info = capi.app_info(app)
    p = capi.JacobiParams()
    for i in range(5):
        p.coef[i] = 0.2
    halo = np.float32(0).tobytes()
    side = torch.cuda.Stream()
    for size in sizes:
        src = torch.rand(size, size, device="cuda")
        dst = torch.empty_like(src)
        dom = capi.Domain(size, size, 0, size, size)
        torch.cuda.synchronize()
        gens = max(64, int(64 * (16384 / size) ** 2))
        gens -= gens % 8
        for T in (1, 2, 4, 8):
            if T > info.max_generations:
                continue
            os.environ["STSTHIP_MAX_GENERATIONS"] = str(T)
            capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True, stream=side.cuda_stream)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                             stream=side.cuda_stream)
                best = min(best, time.perf_counter() - t0)
            print(f"{app} {size}^2 T<={T}: {size * size * gens / best / 1e9:8.1f} Gcell/s", flush=True)
        del src, dst
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
