#!/usr/bin/env python3
"""Race screen of the passes behind an arriving source: 16384^2 random data uploaded in eight row blocks from pinned memory
and followed by the pass driver, again and again (a fresh device buffer filled with NaNs every time, so that a tile that
ran ahead of its block would show), every result compared on the device with the run on a resident grid.
usage: tools/debug/stress_streamed_upload.py [repetitions]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stencilstream_amd import capi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
capi.init(0)
halo = np.float32(0).tobytes()
n = 16384
host = torch.rand(n, n).pin_memory()
resident = host.cuda()
dom = capi.Domain(n, n, 0, n, n)
bad = 0
for name, coef, gens in (("uniform", [0.2] * 5, 1000), ("general", [0.2, 0.21, 0.19, 0.22, 0.18], 500), ("general", [0.2, 0.21, 0.19, 0.22, 0.18], 90)):
    p = capi.JacobiParams()
    for i, c in enumerate(coef):
        p.coef[i] = c
    want = torch.empty_like(resident)
    torch.cuda.synchronize()
    capi.app_run("jacobi5general", p, halo, dom, [resident.data_ptr()], [want.data_ptr()], 0, gens, blocking=True)
    streamed = []
    for r in range(reps):
        dev = torch.full((n, n), float("nan"), device="cuda")
        got = torch.empty_like(dev)
        torch.cuda.synchronize()
        events = capi.upload_in_blocks([(host.data_ptr(), dev.data_ptr(), n * 4)], n, n_blocks=8)
        info = capi.app_run("jacobi5general", p, halo, dom, [dev.data_ptr()], [got.data_ptr()], 0, gens, blocking=True)
        capi.events_destroy(events)
        streamed.append(int(info.n_streamed_passes))
        if not torch.equal(got.view(torch.int32), want.view(torch.int32)):
            bad += 1
            print(f"{name} {gens} generations, repetition {r}: DIFFERS", flush=True)
    print(f"{name}, {gens} generations: {reps} uploads in row blocks followed by the driver (passes as tiles: {min(streamed)}-{max(streamed)})", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
