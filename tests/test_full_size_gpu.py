"""BASELINE configs 3 and 4 at their full sizes on a real MI355X, against the CPU oracle.

The oracle cannot sweep 8192^2 / 4608^2 grids for dozens of generations in seconds, so it runs in window
mode (oracle/stencil_oracle.c): a window of the grid widened by the dependency cone of the run, with the
GLOBAL cell coordinates and grid range handed to the transition function -- HotSpot reflects at the true grid
edges (examples/hotspot/hotspot.cpp:77-87), FDTD places its source by the global index
(examples/fdtd/src/Kernel.hpp:89-92).  Cells of the window further than the cone from its inner borders are
exact and are compared bit for bit.  Windows sit at the grid corners and edges, inside, on the boundary of the
two row strips of the pass driver, and -- FDTD -- on the source cell.

Both layouts (per-field planes and AoS) go through ststhip_app_run with >= 3 launches per run, so the planes
kernels that leave out the stores of constant fields (hotspot `power`, FDTD `ca..db`) are the ones checked."""
import numpy as np
import pytest



def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def window_spots(H, W, w):
    """(row, col) of the windows' top-left corners."""
    return [(0, 0), (0, W - w), (H - w, 0), (H - w, W - w), (H // 2 - w // 2, W // 2 - w // 2), (H // 3, 0),
            (0, W // 3), (H - w, W // 2), (H * 2 // 5 - w // 2, W // 4), (H // 4 - w // 2, W // 4 - w // 2)]


def check_windows(torch, run_oracle, cells_host_window, result_window, H, W, w, margin, spots, what):
    """run_oracle(window_cells, row0, col0) -> oracle result of the widened window; compares the w x w core."""
    for r0, c0 in spots:
        r0, c0 = max(0, min(r0, H - w)), max(0, min(c0, W - w))
        ra, rb, ca, cb = max(0, r0 - margin), min(H, r0 + w + margin), max(0, c0 - margin), min(W, c0 + w + margin)
        ref = run_oracle(cells_host_window(ra, rb, ca, cb), ra, ca)
        want = ref[r0 - ra:r0 - ra + w, c0 - ca:c0 - ca + w]
        got = result_window(r0, r0 + w, c0, c0 + w)
        assert np.array_equal(bits(got), bits(want)), f"{what}: window at ({r0}, {c0})"


# ------------------------------------------------------------------ config 3: HotSpot 8192^2
def hotspot_inputs(torch, N, dtype, device):
    """examples/hotspot/data/input_gen.jl:3-15 (temp 30, power 0.5 on the centre block, 1-based n/4..3n/4),
    plus a seeded perturbation so that every window has structure."""
    gen = torch.Generator(device=device).manual_seed(8192)
    temp = torch.full((N, N), 30.0, device=device, dtype=dtype)
    temp += torch.rand(N, N, device=device, generator=gen).to(dtype) * 8
    power = torch.zeros(N, N, device=device, dtype=dtype)
    power[N // 4 - 1:3 * N // 4, N // 4 - 1:3 * N // 4] = 0.5
    power += torch.rand(N, N, device=device, generator=gen).to(dtype) * 0.01
    return temp, power


@pytest.mark.gpu
@pytest.mark.parametrize("real", ["f32", "f64"])
def test_hotspot_full_size_windows(gpu, oracle, real):
    import torch

    from stencilstream_amd import capi

    N, n, w = 8192, 27, 72  # 27 generations = launches of 8 + 8 + 8 + 2 + 1
    dtype, np_real = (torch.float32, np.float32) if real == "f32" else (torch.float64, np.float64)
    cell_dtype = oracle.HOTSPOT_CELL if real == "f32" else oracle.HOTSPOT_CELL_F64
    p32 = oracle.hotspot_params(N, N)
    vals = [float(p32.Rx_1), float(p32.Ry_1), float(p32.Rz_1), float(p32.Cap_1)]
    if real == "f32":
        po, pc, names = p32, capi.HotspotParams(*vals), ("hotspot", "hotspot_aos")
        run = oracle.hotspot
    else:
        po, pc, names = oracle.HotspotParamsF64(*vals), capi.HotspotParamsF64(*vals), ("hotspot_f64", "hotspot_f64_aos")
        run = oracle.hotspot_f64
    temp, power = hotspot_inputs(torch, N, dtype, gpu)
    dom = capi.Domain(N, N, 0, N, N)
    halo = np.zeros(2, np_real).tobytes()
    s = torch.cuda.Stream()
    torch.cuda.synchronize()

    def host_window(ra, rb, ca, cb):
        cells = np.zeros((rb - ra, cb - ca), dtype=cell_dtype)
        cells["temp"] = temp[ra:rb, ca:cb].cpu().numpy()
        cells["power"] = power[ra:rb, ca:cb].cpu().numpy()
        return cells

    def run_oracle(cells, row0, col0):
        with oracle.window(row0, col0, N, N):
            return run(po, cells, n, n_threads=4)

    # per-field planes
    out_t, out_p = torch.empty_like(temp), torch.empty_like(power)
    info = capi.app_run(names[0], pc, halo, dom, [temp.data_ptr(), power.data_ptr()], [out_t.data_ptr(), out_p.data_ptr()],
                        0, n, blocking=True, stream=s.cuda_stream)
    assert info.n_launches >= 5
    assert torch.equal(out_p, power), "power is carried through unchanged (hotspot.cpp:96)"

    def planes_window(ra, rb, ca, cb):
        cells = np.zeros((rb - ra, cb - ca), dtype=cell_dtype)
        cells["temp"] = out_t[ra:rb, ca:cb].cpu().numpy()
        cells["power"] = out_p[ra:rb, ca:cb].cpu().numpy()
        return cells

    spots = window_spots(N, N, w)
    check_windows(torch, run_oracle, host_window, planes_window, N, N, w, n, spots, names[0])

    # AoS cells: the same result, cell for cell
    aos = torch.stack([temp, power], dim=-1).contiguous()
    out = torch.empty_like(aos)
    torch.cuda.synchronize()  # filled on torch's stream, swept on `s`
    capi.app_run(names[1], pc, halo, dom, [aos.data_ptr()], [out.data_ptr()], 0, n, blocking=True, stream=s.cuda_stream)
    assert torch.equal(out[..., 0], out_t) and torch.equal(out[..., 1], out_p), "AoS and planes sweeps differ"


# ------------------------------------------------------------------ config 4: FDTD max_grid.json
# The experiment of examples/fdtd/experiments/max_grid.json (data, not code): the largest grid of the example
MAX_GRID = {
    "tau": 100e-15, "dx": 3.474e-10,
    "time": {"t_cutoff": 7.0, "t_detect": 14.0, "t_max": 1.5, "t_snap": 0.1},
    "source": {"frequency": 120e12, "phase": 3.0, "x": 0, "y": 0, "radius": 0},
    "cavity_rings": [{"radius": 800e-9, "mu_r": 11.56, "eps_r": 1.0, "sigma": 0.0}],
}


def test_fdtd_max_grid_constants():
    """The set-up arithmetic of the example for max_grid.json (4608^2, 184 911 steps,
    snapshot every 12 328)."""
    from fdtd_harness import Experiment

    ex = Experiment(MAX_GRID)
    assert ex.grid_width() == 4608 and ex.n_timesteps() == 184911 and ex.n_snap_timesteps() == 12328
    k = ex.kernel_constants()
    assert abs(float(k["dt"]) - 8.11203e-19) < 1e-24
    assert k["cutoff_iteration"] > ex.n_timesteps() and k["detect_iteration"] > k["cutoff_iteration"]


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["start", "cutoff", "detect"])
def test_fdtd_max_grid_full_size_windows(gpu, oracle, where):
    """4608^2 with the constants the example derives from max_grid.json; 30 generations (= 60 sweeps; launches
    of 6 x 5) starting at iteration 0, across the source cut-off and across the start of the detection window."""
    import torch

    from fdtd_harness import Experiment
    from stencilstream_amd import capi

    ex = Experiment(MAX_GRID)
    N = ex.grid_width()
    k = ex.kernel_constants()
    po, pc = oracle.FdtdParams(), capi.FdtdParams()
    for name, v in k.items():
        setattr(po, name, v if isinstance(v, int) else float(v))
        setattr(pc, name, v if isinstance(v, int) else float(v))
    n, w = 30, 64
    margin = 2 * n  # two sub-iterations per generation
    offset = {"start": 0, "cutoff": k["cutoff_iteration"] - 13, "detect": k["detect_iteration"] - 11}[where]
    host = ex.initial_grid(oracle.FDTD_CELL)  # ring coefficients per cell, fields zero (fdtd.cpp:188-212)
    rng = np.random.default_rng(4608)
    if where != "start":  # mid-run: fields everywhere
        for f in ("ex", "ey", "hz", "hz_sum"):
            host[f] = (rng.random((N, N), dtype=np.float32) - 0.5) * 1e-3
    cells = torch.from_numpy(host.view(np.float32).reshape(N, N, 8)).to(gpu)
    dom = capi.Domain(N, N, 0, N, N)
    halo = np.zeros(8, np.float32).tobytes()
    s = torch.cuda.Stream()
    torch.cuda.synchronize()

    def run_oracle(window, row0, col0):
        with oracle.window(row0, col0, N, N):
            return oracle.fdtd(po, window, n, iteration_offset=offset, n_threads=4)

    def host_window(ra, rb, ca, cb):
        return host[ra:rb, ca:cb].copy()

    sr, sc = ex.source_r(), ex.source_c()
    spots = window_spots(N, N, w) + [(sr - w // 2, sc - w // 2), (sr - 3, sc - w + 5)]

    # AoS sweep
    out = torch.empty_like(cells)
    info = capi.app_run("fdtd_coef_aos", pc, halo, dom, [cells.data_ptr()], [out.data_ptr()], offset, n, blocking=True,
                        stream=s.cuda_stream)
    assert info.n_launches >= 5

    def aos_window(ra, rb, ca, cb):
        return out[ra:rb, ca:cb].cpu().numpy().view(oracle.FDTD_CELL).reshape(rb - ra, cb - ca)

    check_windows(torch, run_oracle, host_window, aos_window, N, N, w, margin, spots, f"fdtd_coef_aos@{where}")
    got = out.cpu().numpy().view(oracle.FDTD_CELL).reshape(N, N)
    assert np.abs(got["hz"]).max() > 0
    if where == "start":
        # the source acts on the one cell at (source_r, source_c) (radius 0): after 30 steps the field has spread
        # at most 60 cells, everything else is still exactly zero
        far = np.ones((N, N), dtype=bool)
        far[sr - 61:sr + 62, sc - 61:sc + 62] = False
        assert not got["hz"][far].any() and not got["ex"][far].any() and got["hz"][sr, sc] != 0
    if where == "detect":
        assert not np.array_equal(got["hz_sum"], host["hz_sum"])  # the detection window opened inside the run
    for f in ("ca", "cb", "da", "db"):
        assert np.array_equal(bits(got[f]), bits(host[f])), "material coefficients are carried through"

    # per-field planes (stores of ca..db left out from the third launch on): the same cells
    planes = [cells[..., f].contiguous() for f in range(8)]
    outs = [torch.empty_like(p) for p in planes]
    torch.cuda.synchronize()  # filled on torch's stream, swept on `s`
    capi.app_run("fdtd_coef", pc, halo, dom, [p.data_ptr() for p in planes], [p.data_ptr() for p in outs], offset, n,
                 blocking=True, stream=s.cuda_stream)
    for f in range(8):
        assert torch.equal(outs[f], out[..., f]), f"planes and AoS sweeps differ in field {f}"

    # two planes of 16-byte halves (fields / material coefficients; the coefficient plane is not stored again)
    halves = [cells[..., :4].contiguous(), cells[..., 4:].contiguous()]
    outs = [torch.empty_like(h) for h in halves]
    torch.cuda.synchronize()
    capi.app_run("fdtd_coef_grouped", pc, halo, dom, [h.data_ptr() for h in halves], [h.data_ptr() for h in outs], offset,
                 n, blocking=True, stream=s.cuda_stream)
    assert torch.equal(outs[0], out[..., :4]) and torch.equal(outs[1], out[..., 4:]), "grouped planes and AoS sweeps differ"
