#!/bin/bash
# Row strips of the pass driver with moving boundaries: 1 / 2 / 3 strips at several grid sizes (the rule that picks
# between one and two was fitted to strips with boundary bands).  Uniform Jacobi (T = 16) and general coefficients (T = 8),
# 960 generations per call, five calls queued back to back; Gcell-updates/s.
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from stencilstream_amd import capi
capi.init(0)
stream = torch.cuda.Stream()
halo = np.float32(0).tobytes()
def params(coef):
    p = capi.JacobiParams()
    for i, c in enumerate(coef): p.coef[i] = c
    return p
for name, coef in (("uniform", [0.2]*5), ("general", [0.2, 0.21, 0.19, 0.22, 0.18])):
    for n in (4096, 6144, 8192, 12288, 16384):
        src = torch.rand(n, n, device="cuda"); dst = torch.zeros_like(src)
        dom = capi.Domain(n, n, 0, n, n)
        row = []
        for strips in ("0", "1", "2", "3"):
            os.environ["STSTHIP_VIRTUAL_STRIPS"] = strips
            os.environ["STSTHIP_TUNE_DEPTH"] = "0" if name == "general" else "1"
            capi.reload_options()
            p = params(coef)
            capi.app_run("jacobi5general", p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, 960, blocking=True, stream=stream.cuda_stream)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for s in range(5):
                    info = capi.app_run("jacobi5general", p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 0, 960, blocking=(s == 4), stream=stream.cuda_stream)
                best = min(best, (time.perf_counter() - t0) / 5)
            row.append(f"{'rule' if strips == '0' else strips}: {n*n*960/best/1e9:7.1f} ({info.n_launches})")
        print(f"{name} {n}^2  " + "   ".join(row), flush=True)
PY
