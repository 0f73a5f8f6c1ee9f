"""Debug aid: where do the AoS and planes HotSpot sweeps differ at 8192^2?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stencilstream_amd import capi
from oracle import oracle
sys.path.insert(0, "tests")
from test_full_size_gpu import hotspot_inputs
gpu = torch.device("cuda:0")
capi.init(0)
N = 8192
for n in (8, 16, 24, 27):
    p32 = oracle.hotspot_params(N, N)
    vals = [float(p32.Rx_1), float(p32.Ry_1), float(p32.Rz_1), float(p32.Cap_1)]
    pc = capi.HotspotParams(*vals)
    temp, power = hotspot_inputs(torch, N, torch.float32, gpu)
    dom = capi.Domain(N, N, 0, N, N)
    halo = np.zeros(2, np.float32).tobytes()
    s = torch.cuda.Stream(); torch.cuda.synchronize()
    out_t, out_p = torch.empty_like(temp), torch.empty_like(power)
    capi.app_run("hotspot", pc, halo, dom, [temp.data_ptr(), power.data_ptr()], [out_t.data_ptr(), out_p.data_ptr()], 0, n, blocking=True, stream=s.cuda_stream)
    aos = torch.stack([temp, power], dim=-1).contiguous()
    out = torch.empty_like(aos)
    torch.cuda.synchronize()  # filled on torch's stream, swept on `s`
    capi.app_run("hotspot_aos", pc, halo, dom, [aos.data_ptr()], [out.data_ptr()], 0, n, blocking=True, stream=s.cuda_stream)
    dt = (out[..., 0] != out_t); dp = (out[..., 1] != out_p)
    print("n", n, "temp diffs", int(dt.sum()), "power diffs", int(dp.sum()))
    for name, d in (("temp", dt), ("power", dp)):
        if d.any():
            idx = d.nonzero()
            print(name, "rows", int(idx[:, 0].min()), int(idx[:, 0].max()), "cols", int(idx[:, 1].min()), int(idx[:, 1].max()))
            print(idx[:10].tolist())
            r, c = idx[0].tolist()
            print("aos", out[r, c].tolist(), "planes", out_t[r, c].item(), out_p[r, c].item(), "nan?", bool(torch.isnan(out[..., 0]).any()))
