import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
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
os.environ["STSTHIP_VIRTUAL_STRIPS"] = "1"
def run(app, n):
    out = [torch.zeros_like(temp), torch.zeros_like(power)]
    torch.cuda.synchronize()
    capi.app_run(app, p, bytes(8), dom, [temp.data_ptr(), power.data_ptr()], [t.data_ptr() for t in out], 0, n, blocking=True, stream=s.cuda_stream)
    return out
for skip in ("1", "0"):
    os.environ["STSTHIP_SKIP_CONSTANT_STORES"] = skip
    for n in (8, 240):
        ref = run("x_hs_soa_k1t8s1", n)
        for app in ("x_hs_soa_k1t8s4", "x_hs_soa_k2t8s4", "x_hs_soa_k1t12s4", "x_hs_soa_k1t8s1"):
            for trial in range(6):
                out = run(app, n)
                bad = (ref[0].view(torch.int32) != out[0].view(torch.int32))
                if bad.any():
                    idx = bad.nonzero()
                    print(f"skip={skip} n={n} {app} trial {trial}: {idx.shape[0]} cells rows {idx[:,0].min().item()}..{idx[:,0].max().item()} cols {idx[:,1].min().item()}..{idx[:,1].max().item()}", flush=True)
        print("skip", skip, "n", n, "done", flush=True)
