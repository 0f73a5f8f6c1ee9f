"""Debug aid: where does the pass driver with several row strips differ from one strip?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stencilstream_amd import capi
capi.init(0)
N = int(os.environ.get("N", "16384")); gens = int(os.environ.get("GENS", "40"))
gen = torch.Generator(device="cuda").manual_seed(3)
src = torch.rand(N, N, device="cuda", generator=gen)
dom = capi.Domain(N, N, 0, N, N)
s = torch.cuda.Stream(); torch.cuda.synchronize()
for coef in ([0.2] * 5, [0.2, 0.21, 0.19, 0.22, 0.18]):
    p = capi.JacobiParams()
    for i, c in enumerate(coef): p.coef[i] = c
    os.environ["STSTHIP_VIRTUAL_STRIPS"] = "1"
    want = torch.empty_like(src)
    capi.app_run("jacobi5general", p, np.float32(0).tobytes(), dom, [src.data_ptr()], [want.data_ptr()], 0, gens, blocking=True, stream=s.cuda_stream)
    for strips in ("2", "3", "4"):
        os.environ["STSTHIP_VIRTUAL_STRIPS"] = strips
        for rep in range(3):
            got = torch.zeros_like(src)
            torch.cuda.synchronize()  # filled on torch's stream, swept on `s`
            capi.app_run("jacobi5general", p, np.float32(0).tobytes(), dom, [src.data_ptr()], [got.data_ptr()], 0, gens, blocking=True, stream=s.cuda_stream)
            d = (got != want)
            n = int(d.sum())
            msg = f"coef0={coef[1]} strips={strips} rep={rep} diffs={n}"
            if n:
                rows = d.any(dim=1).nonzero().flatten()
                cols = d.any(dim=0).nonzero().flatten()
                msg += f" rows {int(rows.min())}..{int(rows.max())} ({rows.numel()} rows) cols {int(cols.min())}..{int(cols.max())}"
            print(msg, flush=True)
