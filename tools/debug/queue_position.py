"""How the position of the caller's stream in the process's stream-creation order changes the sweep's speed
(HIP deals streams onto hardware queues in creation order).  One fresh process per position."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from stencilstream_amd import capi
k = int(sys.argv[1])
capi.init(0)
dummies = [torch.cuda.Stream() for _ in range(k)]
s = torch.cuda.Stream()
N = 16384
src = torch.rand(N, N, device="cuda"); dst = torch.empty_like(src)
p = capi.JacobiParams()
for i in range(5): p.coef[i] = 0.2
dom = capi.Domain(N, N, 0, N, N)
torch.cuda.synchronize()
out = []
for strips in ("1", "0"):
    os.environ["STSTHIP_VIRTUAL_STRIPS"] = strips
    capi.app_run("jacobi5general", p, np.float32(0).tobytes(), dom, [src.data_ptr()], [dst.data_ptr()], 0, 1000, blocking=True, stream=s.cuda_stream)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        capi.app_run("jacobi5general", p, np.float32(0).tobytes(), dom, [src.data_ptr()], [dst.data_ptr()], 0, 1000, blocking=True, stream=s.cuda_stream)
        best = min(best, time.perf_counter() - t0)
    out.append(round(N * N * 1000 / best / 1e9, 1))
print("dummy streams before the caller's:", k, "prepare:", os.environ.get("STSTHIP_PREPARE_STREAMS", "default"), "single launches:", out[0], "two strips:", out[1], flush=True)
''' % ROOT
for prep in ("0", "1"):
    for k in range(0, 9):
        env = dict(os.environ, STSTHIP_PREPARE_STREAMS=prep)
        r = subprocess.run([sys.executable, "-c", CHILD, str(k)], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-300:], flush=True)
