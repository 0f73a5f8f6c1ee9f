#!/usr/bin/env python3
"""Rows that do not start on a 128-byte boundary: the same grid with the tight pitch (= width, what the example
binaries' grids have) and with the pitch rounded up to 128 bytes.  usage: tools/ab_pitch.py [width ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stencilstream_amd import capi
from tools.bench_apps import hotspot_params


def rate(app, p, halo, H, W, pitch, planes, gens, stream):
    a = [torch.rand(H, pitch, device="cuda", dtype=dt) for dt in planes]
    b = [torch.empty_like(t) for t in a]
    dom = capi.Domain(H, W, 0, H, pitch)
    torch.cuda.synchronize()
    best = 1e9
    for i in range(4):
        t0 = time.perf_counter()
        capi.app_run(app, p, halo, dom, [t.data_ptr() for t in a], [t.data_ptr() for t in b], 0, gens, blocking=True,
                     stream=stream.cuda_stream)
        if i:
            best = min(best, time.perf_counter() - t0)
    return H * W * gens / best / 1e9


def main():
    widths = [int(x) for x in sys.argv[1:]] or [11585, 5792, 2896, 8191, 8192]
    capi.init(0)
    stream = torch.cuda.Stream()
    jp = capi.JacobiParams()
    for i, c in enumerate([0.2, 0.21, 0.19, 0.22, 0.18]):
        jp.coef[i] = c
    for W in widths:
        H = W
        for name, app, p, halo, planes, elem in (
                ("jacobi general", "jacobi5general", jp, np.float32(0).tobytes(), [torch.float32], 4),
                ("hotspot planes", "hotspot", hotspot_params(H), np.zeros(2, np.float32).tobytes(), [torch.float32] * 2, 4),
                ("hotspot fp64 planes", "hotspot_f64", None, np.zeros(2, np.float64).tobytes(), [torch.float64] * 2, 8)):
            if p is None:
                p32 = hotspot_params(H)
                p = capi.HotspotParamsF64(p32.Rx_1, p32.Ry_1, p32.Rz_1, p32.Cap_1)
            per_line = 128 // elem
            padded = (W + per_line - 1) // per_line * per_line
            tight = rate(app, p, halo, H, W, W, planes, 96, stream)
            wide = rate(app, p, halo, H, W, padded, planes, 96, stream)
            print(f"{name} {H} x {W}: pitch {W}: {tight:8.1f}   pitch {padded}: {wide:8.1f} Gcell-updates/s  ({(wide / tight - 1) * 100:+.1f} %)",
                  flush=True)


if __name__ == "__main__":
    main()
