#!/usr/bin/env python3
"""A/B of environment settings of the sweep launcher in one process (same box, alternating):
    tools/ab_env.py NAME=a,b [NAME2=c,d ...] [-- app size generations]
Every combination is timed `reps` times round-robin; prints the median Gcell-updates/s per combination."""
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
    args = sys.argv[1:]
    rest = []
    if "--" in args:
        rest = args[args.index("--") + 1:]
        args = args[:args.index("--")]
    knobs = [(a.split("=", 1)[0], a.split("=", 1)[1].split(",")) for a in args]
    app = rest[0] if rest else "jacobi5general"
    size = int(rest[1]) if len(rest) > 1 else 16384
    gens = int(rest[2]) if len(rest) > 2 else 252
    capi.init(0)
    p = capi.JacobiParams()
    for i in range(5):
        p.coef[i] = 0.2
    src = torch.rand(size, size, device="cuda")
    dst = torch.empty_like(src)
    dom = capi.Domain(size, size, 0, size, size)
    halo = np.float32(0).tobytes()
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    combos = list(itertools.product(*[v for _, v in knobs]))
    times = {c: [] for c in combos}
    for rep in range(7):
        for c in combos:
            for (name, _), value in zip(knobs, c):
                if value == "unset":
                    os.environ.pop(name, None)
                else:
                    os.environ[name] = value
            if rep == 0:
                capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                             stream=s.cuda_stream)
            t0 = time.perf_counter()
            capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                         stream=s.cuda_stream)
            times[c].append(time.perf_counter() - t0)
    for c in combos:
        t = sorted(times[c])[len(times[c]) // 2]
        label = " ".join(f"{n}={v}" for (n, _), v in zip(knobs, c))
        print(f"{app} {size}^2 x {gens}: {label:60s} {size * size * gens / t / 1e9:8.1f} Gcell/s", flush=True)


if __name__ == "__main__":
    main()
