#!/usr/bin/env python3
"""Tapered chunk lengths: STSTHIP_TAPER="permille:split[,...]" cuts the chunks of the last `permille` of a
launch's rows into `split` shorter ones, so that the ragged end of the launch is short.  Sweeps a few
settings for the headline configuration, with one launch per pass and with the two overlapped row strips."""
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
    gens = int(sys.argv[3]) if len(sys.argv) > 3 else 240
    capi.init(0)
    info = capi.app_info(app)
    if app.startswith("jacobi"):
        p = capi.JacobiParams()
        for i in range(5):
            p.coef[i] = 0.2
    else:
        raise SystemExit("only the Jacobi apps are wired up here")
    src = torch.rand(size, size, device="cuda")
    dst = torch.empty_like(src)
    dom = capi.Domain(size, size, 0, size, size)
    halo = np.float32(0).tobytes()
    s = torch.cuda.Stream()
    torch.cuda.synchronize()

    def rate():
        capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True, stream=s.cuda_stream)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            capi.app_run(app, p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, gens, blocking=True,
                         stream=s.cuda_stream)
            ts.append(time.perf_counter() - t0)
        return size * size * gens / sorted(ts)[len(ts) // 2] / 1e9

    specs = sys.argv[4].split() if len(sys.argv) > 4 else [
        "", "150:2", "150:4", "200:3", "120:4", "250:2,100:4", "300:2,120:4", "300:2,150:4,60:8", "400:2,200:4,80:8",
        "200:2,80:6", "150:3,50:9"]
    tails = [int(t) for t in os.environ.get("TUNE_TAIL_PERMILLE", "500").split(",")]
    for strips in os.environ.get("TUNE_STRIPS", "1,0").split(","):
        os.environ["STSTHIP_VIRTUAL_STRIPS"] = strips
        for tail in tails:
            os.environ["STSTHIP_TAIL_PERMILLE"] = str(tail)
            for spec in specs:
                os.environ["STSTHIP_TAPER"] = "" if spec == "none" else spec
                print(f"{app} {size}^2 strips={'one' if strips == '1' else 'auto'} tail weight={tail:3d} permille "
                      f"taper={spec or 'none':16s}: {rate():8.1f} Gcell/s", flush=True)

if __name__ == "__main__":
    main()
