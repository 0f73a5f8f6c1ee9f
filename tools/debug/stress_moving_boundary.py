#!/usr/bin/env python3
"""Race screen of the pass driver's strips with moving boundaries: the same 1000 (and 333) generations of random data at
16384^2, again and again, every result compared on the device with the one of the first run with fixed strips and
boundary bands (STSTHIP_SKEWED_STRIPS=0).  Uniform form (T = 16) and general coefficients (T = 8; depth pinned).
usage: tools/debug/stress_moving_boundary.py [repetitions]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stencilstream_amd import capi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
capi.init(0)
stream = torch.cuda.Stream()
halo = np.float32(0).tobytes()
n = 16384
src = torch.rand(n, n, device="cuda")
dom = capi.Domain(n, n, 0, n, n)
bad = 0
for name, coef in (("uniform", [0.2] * 5), ("general", [0.2, 0.21, 0.19, 0.22, 0.18])):
    p = capi.JacobiParams()
    for i, c in enumerate(coef):
        p.coef[i] = c
    os.environ["STSTHIP_TUNE_DEPTH"] = "8" if name == "general" else "1"
    for gens in (1000, 333):
        os.environ["STSTHIP_SKEWED_STRIPS"] = "0"
        capi.reload_options()
        want = torch.empty_like(src)
        capi.app_run("jacobi5general", p, halo, dom, [src.data_ptr()], [want.data_ptr()], 0, gens, blocking=True, stream=stream.cuda_stream)
        os.environ["STSTHIP_SKEWED_STRIPS"] = "1"
        capi.reload_options()
        got = torch.empty_like(src)
        for r in range(reps):
            got.zero_()
            torch.cuda.synchronize()
            info = capi.app_run("jacobi5general", p, halo, dom, [src.data_ptr()], [got.data_ptr()], 0, gens, blocking=True, stream=stream.cuda_stream)
            if not torch.equal(got.view(torch.int32), want.view(torch.int32)):
                bad += 1
                print(f"{name} {gens} generations, repetition {r}: DIFFERS in {(got != want).sum().item()} cells", flush=True)
        print(f"{name}, {gens} generations: {reps} repetitions with moving boundaries ({info.n_launches} launches each) compared with fixed strips", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
