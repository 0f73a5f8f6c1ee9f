"""Race screen for the staged sweeps: many repetitions of long runs against the independent-wave result."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from stencilstream_amd import capi
import bench_apps
capi.init(0)
s = torch.cuda.Stream()
H = W = 8192
gen = torch.Generator(device="cuda").manual_seed(5)
temp = 320 + 10 * torch.rand(H, W, device="cuda", generator=gen)
power = 0.01 * torch.rand(H, W, device="cuda", generator=gen)
dom = capi.Domain(H, W, 0, H, W)
p = bench_apps.hotspot_params(H)
os.environ["STSTHIP_VIRTUAL_STRIPS"] = sys.argv[2] if len(sys.argv) > 2 else "1"
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
def run(app, n):
    out = [torch.zeros_like(temp), torch.zeros_like(power)]
    torch.cuda.synchronize()
    capi.app_run(app, p, bytes(8), dom, [temp.data_ptr(), power.data_ptr()], [t.data_ptr() for t in out], 0, n, blocking=True, stream=s.cuda_stream)
    return out
n = 240
ref = run("hotspot", n)
for app in ("x_hs_soa_k1t8s4", "x_hs_soa_k1t12s4", "x_hs_soa_k2t12s4"):
    bad_runs = 0
    for trial in range(trials):
        out = run(app, n)
        bad = (ref[0].view(torch.int32) != out[0].view(torch.int32))
        if bad.any():
            bad_runs += 1
            idx = bad.nonzero()
            print(f"{app} trial {trial}: {idx.shape[0]} cells rows {idx[:,0].min().item()}..{idx[:,0].max().item()} cols {idx[:,1].min().item()}..{idx[:,1].max().item()}", flush=True)
    print(f"{app}: {bad_runs} wrong runs of {trials} ({trials * n // capi.app_info(app).max_generations} launches)", flush=True)
