import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from stencilstream_amd import capi
capi.init(0)
s = torch.cuda.Stream()
def hp(n):
    sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
    import bench_apps
    return bench_apps.hotspot_params(n)
for (H, W) in [(300, 257), (700, 1100), (1500, 2300), (8192, 8192)]:
    gen = torch.Generator(device="cuda").manual_seed(5)
    temp = 320 + 10 * torch.rand(H, W, device="cuda", generator=gen)
    power = 0.01 * torch.rand(H, W, device="cuda", generator=gen)
    dom = capi.Domain(H, W, 0, H, W)
    p = hp(H)
    for strips in ("1", "0"):
        os.environ["STSTHIP_VIRTUAL_STRIPS"] = strips
        for n in (8, 16, 24, 40, 240):
            ref = None
            for app in ("x_hs_soa_k1t8s1", "x_hs_soa_k1t8s4", "x_hs_soa_k2t8s4", "x_hs_soa_k1t12s4"):
                out = [torch.zeros_like(temp), torch.zeros_like(power)]
                torch.cuda.synchronize()
                capi.app_run(app, p, bytes(8), dom, [temp.data_ptr(), power.data_ptr()], [t.data_ptr() for t in out], 0, n, blocking=True, stream=s.cuda_stream)
                if ref is None:
                    ref = out
                else:
                    bad_t = (ref[0].view(torch.int32) != out[0].view(torch.int32))
                    bad_p = (ref[1].view(torch.int32) != out[1].view(torch.int32))
                    if bad_t.any() or bad_p.any():
                        idx = bad_t.nonzero()
                        idp = bad_p.nonzero()
                        print(f"{H}x{W} strips={strips} n={n} {app}: temp differs at {idx.shape[0]} cells rows {idx[:,0].min().item() if len(idx) else None}..{idx[:,0].max().item() if len(idx) else None} cols {idx[:,1].min().item() if len(idx) else None}..{idx[:,1].max().item() if len(idx) else None}; power differs at {idp.shape[0]} rows {idp[:,0].min().item() if len(idp) else None}..{idp[:,0].max().item() if len(idp) else None}", flush=True)
    print(H, W, "done", flush=True)
