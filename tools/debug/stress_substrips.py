#!/usr/bin/env python3
"""Race screen of the strip driver's sub-strips with a moving boundary: one rank's 2048 x 16384 strip (the middle one of
three) over the RCCL loopback -- real sends and receives on one GPU, the rank is its own neighbour on both sides --, 1000
generations again and again, every result compared on the device with the same strip swept as one launch per pass.
usage: tools/debug/stress_substrips.py [repetitions]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from stencilstream_amd import capi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
capi.init(0)
comm = capi.comm_create(capi.comm_unique_id(), 0, 1)
capi.comm_set_neighbours(comm, 0, 0)
R, W = 2048, 16384
init = torch.rand(R, W, device="cuda")
torch.cuda.synchronize()
bad = 0


def run(p, halo, sub):
    os.environ["STSTHIP_STRIP_SUBSTRIPS"] = sub
    capi.reload_options()
    strip = capi.Strip("jacobi5general", p, halo, 3 * R, W, 1, 3, comm=comm)
    strip.upload_from_device(0, init.data_ptr(), init.numel() * 4)
    strip.warm_up()
    strip.advance(0, 600)
    strip.advance(600, 400, blocking=True)
    ptr, row_bytes = strip.plane(0)
    out = torch.empty_like(init)
    capi.check(capi.load().ststhip_memcpy_d2d(out.data_ptr(), ptr, R * row_bytes, None), "d2d")
    capi.check(capi.load().ststhip_stream_synchronize(None), "sync")
    launches = strip.counters()[0]
    strip.close()
    return out, launches


for name, coef in (("uniform", [0.2] * 5), ("general", [0.2, 0.21, 0.19, 0.22, 0.18])):
    p = capi.JacobiParams()
    for i, c in enumerate(coef):
        p.coef[i] = c
    halo = np.float32(0).tobytes()
    want, one = run(p, halo, "1")
    for r in range(reps):
        got, two = run(p, halo, "2")
        if not torch.equal(got.view(torch.int32), want.view(torch.int32)):
            bad += 1
            print(f"{name}, repetition {r}: DIFFERS", flush=True)
    print(f"{name}: {reps} repetitions as two sub-strips ({two} launches) against one launch per pass ({one})", flush=True)
capi.comm_destroy(comm)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
