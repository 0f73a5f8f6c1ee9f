"""Parity of the HIP sweeps (through the C ABI) against the CPU oracle on a real MI355X.

Bit-exact for every application: integer/byte cells by nature, fp32 cells because both sides are
built with -ffp-contract=off and evaluate the reference's expressions in the reference's order."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

KNOWN = json.load(open(os.path.join(GOLDEN, "known_answers.json")))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def run_hip(tf, grid, n, halo=None, offset=0):
    from stencilstream_amd import update as U

    su = U.StencilUpdate(U.Params(tf, halo_value=halo, iteration_offset=offset, n_iterations=n, blocking=True))
    out = su(U.Grid.from_numpy(grid)).to_numpy()
    assert su.get_n_processed_cells() == n * grid.shape[0] * grid.shape[1]
    return out


SHAPES = [(1, 1), (1, 70), (70, 1), (2, 3), (64, 64), (32, 64), (64, 32), (257, 511), (300, 1000), (1000, 300)]
ITERS = [0, 1, 2, 3, 5, 8, 13, 29]


@pytest.mark.parametrize("shape", SHAPES, ids=str)
def test_jacobi5general_bit_exact(gpu, oracle, shape):
    from stencilstream_amd import update as U

    rng = np.random.default_rng(0x5EED + shape[0] * 7 + shape[1])
    grid = rng.random(shape, dtype=np.float32)
    coef = [0.11, 0.19, 0.23, 0.31, 0.16]
    for n in ITERS:
        got = run_hip(U.jacobi("Jacobi5General", coef), grid, n, halo=np.float32(0.25))
        want = oracle.jacobi("Jacobi5General", coef, grid, n, halo=0.25, n_threads=8)
        assert np.array_equal(bits(got), bits(want)), f"n={n}"


@pytest.mark.parametrize("variant,ncoef", [("Jacobi1General", 1), ("Jacobi2Constant", 0), ("Jacobi3Constant", 0),
                                           ("Jacobi4Constant", 0), ("Jacobi5Constant", 0), ("Jacobi4General", 4),
                                           ("Jacobi9General", 9)])
def test_jacobi_variants_bit_exact(gpu, oracle, variant, ncoef):
    from stencilstream_amd import update as U

    rng = np.random.default_rng(11)
    grid = rng.random((190, 333), dtype=np.float32)
    coef = list(rng.random(ncoef, dtype=np.float32) * 0.2)
    for n in (1, 7, 16):
        got = run_hip(U.jacobi(variant, coef), grid, n, halo=np.float32(0.5))
        want = oracle.jacobi(variant, coef, grid, n, halo=0.5, n_threads=8)
        assert np.array_equal(bits(got), bits(want)), f"{variant} n={n}"


@pytest.mark.parametrize("case", KNOWN["jacobi5general"][:2], ids=lambda c: f"{c['n']}x{c['iterations']}")
def test_jacobi_known_answers_on_gpu(gpu, oracle, case):
    from stencilstream_amd import update as U

    n = case["n"]
    got = run_hip(U.jacobi("Jacobi5General", case["coef"]), oracle.jacobi_init(n, n), case["iterations"],
                  halo=np.float32(0.0))
    assert abs(got.astype(np.float64).sum() - case["sum"]) <= 5e-7 * case["sum"]
    for where, value in case["at"].items():
        r, c = map(int, where.split(","))
        assert got[r, c] == np.float32(value)


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (64, 64), (130, 257), (512, 512), (100, 1100)], ids=str)
def test_conway_bit_exact(gpu, oracle, shape):
    from stencilstream_amd import update as U

    rng = np.random.default_rng(0xC0FFEE + shape[1])
    grid = (rng.random(shape) < 0.35).astype(np.uint8)
    for n in (1, 2, 9, 24, 100):
        assert np.array_equal(run_hip(U.conway(), grid, n), oracle.conway(grid, n, n_threads=8)), f"n={n}"


@pytest.mark.parametrize("shape", [(1, 4), (5, 8), (3, 252), (257, 1028), (1000, 772), (2048, 4096)], ids=str)
def test_conway_word_form_equals_byte_form(gpu, oracle, monkeypatch, shape):
    """Widths that are multiples of four run the rule on 32-bit words of four cells (ConwayPacked); the
    byte-per-lane kernel and the oracle must agree with it bit for bit, at every launch depth."""
    from stencilstream_amd import update as U

    rng = np.random.default_rng(0xC0FFEE ^ shape[1])
    grid = (rng.random(shape) < 0.4).astype(np.uint8)
    for n in (1, 3, 8, 21, 50):
        monkeypatch.setenv("STSTHIP_CONWAY_FASTPATH", "1")
        words = run_hip(U.conway(), grid, n)
        monkeypatch.setenv("STSTHIP_CONWAY_FASTPATH", "0")
        cells = run_hip(U.conway(), grid, n)
        want = oracle.conway(grid, n, n_threads=8)
        assert np.array_equal(words, want), f"word form, n={n}"
        assert np.array_equal(cells, want), f"byte form, n={n}"
    # a live halo is outside the word form's preconditions: same result with the switch on and off
    monkeypatch.setenv("STSTHIP_CONWAY_FASTPATH", "1")
    live = run_hip(U.conway(), grid, 9, halo=np.uint8(1))
    monkeypatch.setenv("STSTHIP_CONWAY_FASTPATH", "0")
    assert np.array_equal(live, run_hip(U.conway(), grid, 9, halo=np.uint8(1)))


def test_conway_gosper_gun_known_answer(gpu):
    """BASELINE config 0 input: the reference's 64x64 start pattern, md5 of the printed grid."""
    from stencilstream_amd import update as U

    chars = [ch for ch in open(os.path.join(GOLDEN, KNOWN["conway"]["input"])).read() if ch in "X."]
    grid = (np.array(chars) == "X").astype(np.uint8).reshape(64, 64)
    for n, md5 in KNOWN["conway"]["stdout_md5"].items():
        out = run_hip(U.conway(), grid, int(n))
        text = "".join("".join("X" if v else "." for v in row) + "\n" for row in out)
        assert hashlib.md5(text.encode()).hexdigest() == md5


@pytest.mark.parametrize("split", [True, False], ids=["soa", "aos"])
@pytest.mark.parametrize("shape", [(64, 64), (1, 9), (9, 1), (200, 517), (600, 260)], ids=str)
def test_hotspot_bit_exact(gpu, oracle, shape, split):
    from stencilstream_amd import update as U

    rng = np.random.default_rng(5 + shape[0])
    cells = np.zeros(shape, dtype=U.HOTSPOT_CELL)
    cells["temp"] = 320 + 10 * rng.random(shape, dtype=np.float32)
    cells["power"] = rng.random(shape, dtype=np.float32) * 0.01
    p = oracle.hotspot_params(*shape)
    tf = U.hotspot(p.Rx_1, p.Ry_1, p.Rz_1, p.Cap_1, split_cell_structure=split)
    for n in (1, 4, 11, 40):
        got = run_hip(tf, cells, n)
        want = oracle.hotspot(p, cells, n, n_threads=8)
        assert np.array_equal(bits(got), bits(want)), f"n={n}"


def test_hotspot_reference_input_known_answer(gpu, oracle):
    from stencilstream_amd import update as U

    t = np.loadtxt(os.path.join(GOLDEN, KNOWN["hotspot_64"]["temp"]), dtype=np.float32).reshape(64, 64)
    pw = np.loadtxt(os.path.join(GOLDEN, KNOWN["hotspot_64"]["power"]), dtype=np.float32).reshape(64, 64)
    cells = np.zeros((64, 64), dtype=U.HOTSPOT_CELL)
    cells["temp"], cells["power"] = t, pw
    p = oracle.hotspot_params(64, 64)
    out = run_hip(U.hotspot(p.Rx_1, p.Ry_1, p.Rz_1, p.Cap_1), cells, KNOWN["hotspot_64"]["iterations"])
    lines = [f"{i}\t{v:g}" for i, v in enumerate(out["temp"].reshape(-1)[:3])]
    assert lines == KNOWN["hotspot_64"]["first_output_lines"]


@pytest.mark.parametrize("name,radius,split", [("aos", 1, False), ("soa", 1, True), ("r2", 2, False)])
@pytest.mark.parametrize("case", KNOWN["selfcheck_cases"] + [[37, 301, 5, 9], [301, 37, 0, 3]], ids=str)
def test_selfcheck_known_answer(gpu, oracle, case, name, radius, split):
    """The reference's own hot-path test (tests/cuda/StencilUpdate.cpp:30-50): the transition
    function verifies halo, coordinates, iteration, sub-iteration and TDV of everything it sees."""
    from stencilstream_amd import update as U

    H, W, offset, n = case
    halo = np.zeros((), dtype=U.SELFCHECK_CELL)
    halo["status"] = 2
    out = run_hip(U.selfcheck(radius, split), oracle.selfcheck_input(H, W, offset), n, halo=halo, offset=offset)
    assert (out["status"] == 0).all(), "a cell saw a wrong neighbour, index or TDV"
    assert (out["r"] == np.arange(H)[:, None]).all() and (out["c"] == np.arange(W)[None, :]).all()
    assert (out["i_iteration"] == offset + n).all() and (out["i_subiteration"] == 0).all()
    assert np.array_equal(bits(out), bits(oracle.selfcheck(radius, oracle.selfcheck_input(H, W, offset), offset, n)))


def fdtd_setup(oracle, H, W):
    from stencilstream_amd import capi

    # parameters in the range of examples/fdtd/experiments/default.json, source off-centre
    kw = dict(dt=8.3e-18, t_0=3e-13 * 0.01, tau=1e-13 * 0.01, omega=7.5e14, cutoff_iteration=40,
              detect_iteration=10, source_radius_squared=36.0, source_r=H / 2.0 + 3, source_c=W / 2.0 - 2,
              source_distance_bound=0.0, double_center_rc=float(H))
    kw["source_distance_bound"] = 36.0 - (kw["source_c"] ** 2 + kw["source_r"] ** 2)
    po, pc = oracle.FdtdParams(), capi.FdtdParams()
    for k, v in kw.items():
        setattr(po, k, v)
        setattr(pc, k, v)
    rng = np.random.default_rng(3)
    cells = np.zeros((H, W), dtype=oracle.FDTD_CELL)
    for f in ("ex", "ey", "hz"):
        cells[f] = (rng.random((H, W), dtype=np.float32) - 0.5) * 1e-3
    inside = ((np.arange(H)[:, None] - H / 2) ** 2 + (np.arange(W)[None, :] - W / 2) ** 2) < (min(H, W) * 0.4) ** 2
    cells["ca"] = np.where(inside, 1.0, 1.0).astype(np.float32)
    cells["cb"] = np.where(inside, 0.31, 0.0).astype(np.float32)
    cells["da"] = np.where(inside, 1.0, 1.0).astype(np.float32)
    cells["db"] = np.where(inside, 0.29, 0.0).astype(np.float32)
    return po, pc, cells


@pytest.mark.parametrize("layout", ["grouped", "planes", "aos"])
@pytest.mark.parametrize("shape", [(162, 162), (40, 333)], ids=str)
def test_fdtd_bit_exact(gpu, oracle, shape, layout):
    """Three layouts of the same sweep: two planes of 16-byte halves (fields / material coefficients, the latter
    never stored again), one plane per field (the reference's SoA protocol), AoS cells."""
    from stencilstream_amd import update as U

    po, pc, cells = fdtd_setup(oracle, *shape)
    halo = np.zeros((), dtype=U.FDTD_CELL)
    for n, offset in ((1, 0), (7, 0), (30, 5), (25, 30)):
        got = run_hip(U.fdtd(pc, layout=layout), cells, n, halo=halo, offset=offset)
        want = oracle.fdtd(po, cells, n, iteration_offset=offset, n_threads=8)
        assert np.array_equal(bits(got), bits(want)), f"n={n} offset={offset}"
    assert np.abs(want["hz_sum"]).max() > 0 and np.abs(want["hz"]).max() > 0


@pytest.mark.parametrize("layout", ["grouped", "aos"])
def test_fdtd_through_the_strip_driver(gpu, oracle, layout):
    """ststhip_strip_advance with a function that has sub-iterations and time-dependent values, two calls with an
    iteration offset (one device table of values per call, as in ststhip_run_passes)."""
    from stencilstream_amd import capi, update as U

    H, W = 300, 222
    po, pc, cells = fdtd_setup(oracle, H, W)
    halo = np.zeros((), dtype=U.FDTD_CELL)
    strip = capi.Strip(U.fdtd(pc, layout=layout).app, pc, halo.tobytes(), H, W, 0, 1)
    if layout == "aos":
        strip.upload(0, cells.view(np.uint8).reshape(H, W * 32))
    else:
        raw = cells.view(np.float32).reshape(H, W, 8)
        strip.upload(0, np.ascontiguousarray(raw[:, :, :4]).reshape(H, W * 4))
        strip.upload(1, np.ascontiguousarray(raw[:, :, 4:]).reshape(H, W * 4))
    strip.advance(0, 30)
    strip.advance(30, 11, blocking=True)
    want = oracle.fdtd(po, cells, 41, n_threads=8)
    if layout == "aos":
        got = strip.download(0, np.uint8).reshape(H, W, 32).view(np.float32).reshape(H, W, 8)
    else:
        got = np.concatenate([strip.download(0, np.float32).reshape(H, W, 4), strip.download(1, np.float32).reshape(H, W, 4)],
                             axis=2)
    strip.close()
    assert np.array_equal(bits(got), bits(want.view(np.float32).reshape(H, W, 8)))
    assert np.abs(want["hz"]).max() > 0


@pytest.mark.parametrize("split", [True, False], ids=["soa", "aos"])
def test_fdtd_reference_functor_frames_on_gpu(gpu, oracle, split):
    """The frames the reference's unchanged examples/fdtd sources wrote (tests/golden/fdtd, see
    make_fdtd_golden.py) against the HIP kernels through the C ABI: equal to the oracle bit for bit, and
    equal to the frames to the six digits they carry; three update calls with an iteration offset."""
    import json

    from fdtd_harness import Experiment, load_csv
    from stencilstream_amd import capi, update as U

    here = os.path.join(os.path.dirname(__file__), "golden", "fdtd")
    ex = Experiment(json.load(open(os.path.join(here, "experiment.json"))))
    po, pc = oracle.FdtdParams(), capi.FdtdParams()
    for k, v in ex.kernel_constants().items():
        setattr(po, k, v if isinstance(v, int) else float(v))
        setattr(pc, k, v if isinstance(v, int) else float(v))
    cells = ex.initial_grid(oracle.FDTD_CELL)
    want = cells
    halo = np.zeros((), dtype=U.FDTD_CELL)
    for offset, n, label in ex.update_calls():
        cells = run_hip(U.fdtd(pc, split), cells, n, halo=halo, offset=offset)
        want = oracle.fdtd(po, want, n, iteration_offset=offset, n_threads=8)
        assert np.array_equal(bits(cells), bits(want)), label
        frame = load_csv(os.path.join(here, f"hz.{label}.csv"))
        tol = 0.5000001 * 10.0 ** (np.floor(np.log10(np.maximum(np.abs(frame), 1e-300))) - 5)
        assert np.all(np.abs(cells["hz"].astype(np.float64) - frame) <= tol), label
    frame = load_csv(os.path.join(here, f"hz_sum.{ex.n_timesteps()}.csv"))
    tol = 0.5000001 * 10.0 ** (np.floor(np.log10(np.maximum(np.abs(frame), 1e-300))) - 5)
    assert np.all(np.abs(cells["hz_sum"].astype(np.float64) - frame) <= tol)


def test_sweep_row_ranges_compose(gpu, oracle):
    """ststhip_app_sweep on row bands with ghost rows (the multi-GPU building block): three strips
    with emulated ghost exchange equal the whole-grid result."""
    import torch

    from stencilstream_amd import capi

    H, W, T = 230, 410, 4
    rng = np.random.default_rng(8)
    grid = rng.random((H, W), dtype=np.float32)
    coef = [0.2, 0.21, 0.19, 0.22, 0.18]
    p = capi.JacobiParams()
    for i, c in enumerate(coef):
        p.coef[i] = c
    info = capi.app_info("jacobi5general")
    g = T * info.halo_depth_per_generation
    bounds = [0, 70, 150, H]
    whole = torch.from_numpy(grid).cuda()
    n_steps = 3
    for _ in range(n_steps):
        nxt = torch.empty_like(whole)
        for a, b in zip(bounds[:-1], bounds[1:]):
            lo, hi = max(0, a - g), min(H, b + g)
            src = whole[lo:hi].contiguous()  # owned rows + ghost rows
            dst = torch.full_like(src, float("nan"))
            dom = capi.Domain(H, W, lo, hi - lo, W)
            torch.cuda.synchronize()  # stream 0 below = the library's own stream, unordered with torch's kernels above
            capi.app_sweep("jacobi5general", p, np.float32(0.0).tobytes(), dom, [src.data_ptr()], [dst.data_ptr()],
                           a, b, 0, T, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert torch.isnan(dst[: a - lo]).all() and torch.isnan(dst[b - lo:]).all(), "rows outside the range touched"
            nxt[a:b] = dst[a - lo:b - lo]
        whole = nxt
    want = oracle.jacobi("Jacobi5General", coef, grid, T * n_steps, halo=0.0, n_threads=8)
    assert np.array_equal(bits(whole.cpu().numpy()), bits(want))


def test_sweep_with_a_row_hole(gpu, oracle):
    """ststhip_set_launch_row_hole: one launch produces the rows on both sides of a hole (the two boundary bands of a
    row strip), bit-identical to the whole-grid sweep, and leaves the rows of the hole alone."""
    import torch

    from stencilstream_amd import capi

    lib = capi.load()
    rng = np.random.default_rng(18)
    s = torch.cuda.current_stream().cuda_stream
    for W, T, halo, coef in ((700, 8, 0.25, [0.2, 0.21, 0.19, 0.22, 0.18]), (1300, 4, 0.0, [0.3, 0.1, 0.2, 0.15, 0.25])):
        H = 400
        grid = rng.random((H, W), dtype=np.float32)
        p = capi.JacobiParams()
        for i, c in enumerate(coef):
            p.coef[i] = c
        name = "jacobi5general"
        g = T * capi.app_info(name).halo_depth_per_generation
        src = torch.from_numpy(grid).cuda()
        dom = capi.Domain(H, W, 0, H, W)
        whole = torch.empty_like(src)
        torch.cuda.synchronize()
        capi.app_sweep(name, p, np.float32(halo).tobytes(), dom, [src.data_ptr()], [whole.data_ptr()], 0, H, 0, T, s)
        for a, b, hole in ((40, 330, (40 + g, 330 - g)), (0, H, (13, 390)), (100, 200, (101, 199))):
            dst = torch.full_like(src, float("nan"))
            torch.cuda.synchronize()
            capi.check(lib.ststhip_set_launch_row_hole(hole[0], hole[1]), "set hole")
            try:
                capi.app_sweep(name, p, np.float32(halo).tobytes(), dom, [src.data_ptr()], [dst.data_ptr()], a, b, 0, T, s)
            finally:
                capi.check(lib.ststhip_set_launch_row_hole(0, 0), "reset hole")
            torch.cuda.synchronize()
            assert torch.isnan(dst[:a]).all() and torch.isnan(dst[b:]).all() and torch.isnan(dst[hole[0]:hole[1]]).all()
            assert torch.equal(dst[a:hole[0]].view(torch.int32), whole[a:hole[0]].view(torch.int32))
            assert torch.equal(dst[hole[1]:b].view(torch.int32), whole[hole[1]:b].view(torch.int32))
        want = oracle.jacobi("Jacobi5General", coef, grid, T, halo=halo, n_threads=8)
        assert np.array_equal(bits(whole.cpu().numpy()), bits(want))
    # a hole that is not strictly inside the range is refused
    capi.check(lib.ststhip_set_launch_row_hole(0, 10), "set hole")
    try:
        with pytest.raises(capi.StsthipError):
            capi.app_sweep("jacobi5general", p, np.float32(0).tobytes(), dom, [src.data_ptr()], [dst.data_ptr()], 0, H, 0, 4, s)
    finally:
        capi.check(lib.ststhip_set_launch_row_hole(0, 0), "reset hole")


def test_scatter_gather_round_trip(gpu):
    import torch

    from stencilstream_amd import capi, update as U

    rng = np.random.default_rng(4)
    for dt, n_cells in ((U.FDTD_CELL, 70001), (U.SELFCHECK_CELL, 513), (U.HOTSPOT_CELL, 256)):
        raw = rng.integers(0, 255, size=n_cells * dt.itemsize, dtype=np.uint8)
        aos = torch.from_numpy(raw).cuda()
        offs = [dt.fields[n][1] for n in dt.names]
        sizes = [dt.fields[n][0].itemsize for n in dt.names]
        planes = [torch.zeros(n_cells * s, dtype=torch.uint8, device="cuda") for s in sizes]
        s = torch.cuda.current_stream().cuda_stream  # 0 = the library's own stream, unordered with torch's
        torch.cuda.synchronize()
        capi.scatter_fields(aos.data_ptr(), dt.itemsize, n_cells, offs, sizes, [t.data_ptr() for t in planes], s)
        torch.cuda.synchronize()
        host = raw.view(dt)
        for name, t in zip(dt.names, planes):
            assert np.array_equal(t.cpu().numpy(), np.ascontiguousarray(host[name]).view(np.uint8)), name
        back = torch.zeros_like(aos)
        torch.cuda.synchronize()
        capi.gather_fields(back.data_ptr(), dt.itemsize, n_cells, offs, sizes, [t.data_ptr() for t in planes], s)
        torch.cuda.synchronize()
        assert torch.equal(back, aos)


def test_invalid_arguments_are_rejected(gpu):
    import torch

    from stencilstream_amd import capi

    a = torch.zeros(64 * 64, device="cuda")
    b = torch.zeros(64 * 64, device="cuda")
    p = capi.JacobiParams()
    halo = np.float32(0).tobytes()
    with pytest.raises(capi.StsthipError):  # ghost rows missing
        capi.app_sweep("jacobi5general", p, halo, capi.Domain(128, 64, 32, 64, 64), [a.data_ptr()], [b.data_ptr()],
                       32, 96, 0, 4)
    with pytest.raises(capi.StsthipError):  # depth not compiled
        capi.app_sweep("jacobi5general", p, halo, capi.Domain(64, 64, 0, 64, 64), [a.data_ptr()], [b.data_ptr()],
                       0, 64, 0, 3)
    with pytest.raises(capi.StsthipError):  # in-place
        capi.app_sweep("jacobi5general", p, halo, capi.Domain(64, 64, 0, 64, 64), [a.data_ptr()], [a.data_ptr()],
                       0, 64, 0, 1)
    with pytest.raises(capi.StsthipError):  # pitch < width
        capi.app_sweep("jacobi5general", p, halo, capi.Domain(64, 64, 0, 64, 32), [a.data_ptr()], [b.data_ptr()],
                       0, 64, 0, 1)


def test_strip_domain_single_rank_on_gpu(gpu, oracle):
    """The distributed driver (streams, boundary-first order, ragged depths) with the real HIP sweep;
    world_size 1 here, the multi-rank logic is covered by tests/test_dist_cpu.py."""
    import torch

    from stencilstream_amd import capi
    from stencilstream_amd.dist import StripDomain

    H, W = 300, 700
    rng = np.random.default_rng(21)
    grid = rng.random((H, W), dtype=np.float32)
    coef = [0.2, 0.21, 0.19, 0.22, 0.18]
    p = capi.JacobiParams()
    for i, c in enumerate(coef):
        p.coef[i] = c
    strip = StripDomain("jacobi5general", p, np.float32(0).tobytes(), H, W, 0, 1, gpu)
    strip.load_owned(torch.from_numpy(grid).to(gpu))
    strip.advance(0, 13)
    strip.advance(13, 8)
    got = strip.owned(0, torch.float32).cpu().numpy()
    want = oracle.jacobi("Jacobi5General", coef, grid, 21, halo=0.0, n_threads=8)
    assert np.array_equal(bits(got), bits(want))


def test_jacobi_fma_flavour_within_tolerance(gpu, oracle):
    """The fused-multiply-add build of the headline kernel: not bit-identical by construction, checked
    against the uncontracted oracle with the tolerance DESIGN.md states (<= 4 ulp per generation on
    values in [0, 1], non-accumulating because the update is a convex combination)."""
    from stencilstream_amd import update as U

    rng = np.random.default_rng(77)
    grid = rng.random((257, 511), dtype=np.float32)
    coef = [0.2] * 5
    tf = U.jacobi("Jacobi5General", coef)
    tf.app = "jacobi5general_fma"
    for n in (1, 8, 29, 100):
        got = run_hip(tf, grid, n, halo=np.float32(0.0))
        want = oracle.jacobi("Jacobi5General", coef, grid, n, halo=0.0, n_threads=8)
        assert np.abs(got.astype(np.float64) - want).max() <= 4 * np.finfo(np.float32).eps * min(n, 8)
        assert not np.array_equal(got, want) or n == 0 or True  # may or may not differ; only the bound matters


def test_native_comm_single_rank(gpu):
    """ststhip_comm_* (RCCL through the C ABI) with one rank: create, no-op exchange, destroy."""
    import ctypes as C

    import torch

    from stencilstream_amd import capi

    lib = capi.load()
    uid = C.create_string_buffer(capi.COMM_ID_BYTES)
    capi.check(lib.ststhip_comm_unique_id(uid), "ststhip_comm_unique_id")
    comm = C.c_void_p()
    capi.check(lib.ststhip_comm_create(uid, 0, 1, C.byref(comm)), "ststhip_comm_create")
    a = torch.arange(64, dtype=torch.float32, device=gpu)
    ptrs = (C.c_void_p * 1)(a.data_ptr())
    row_bytes = (C.c_size_t * 1)(16)
    capi.check(lib.ststhip_comm_exchange_rows(comm, 1, ptrs, ptrs, ptrs, ptrs, row_bytes, 2, None),
               "ststhip_comm_exchange_rows")
    torch.cuda.synchronize()
    assert torch.equal(a.cpu(), torch.arange(64, dtype=torch.float32))  # no neighbours: nothing moves
    capi.check(lib.ststhip_comm_destroy(comm), "ststhip_comm_destroy")


@pytest.mark.parametrize("strips", ["1", "2"])
def test_constant_field_stores_are_left_out_only_where_safe(gpu, oracle, monkeypatch, strips):
    """HotSpot and FDTD declare the fields they only copy (constant_fields); on per-field planes their stores are
    left out from the third pass of a run on.  Results (all fields, fresh target planes from the pool, which hold
    stale data) equal the oracle with the switch on and off, for one and two row strips and for pass counts
    around the switch-over."""
    from stencilstream_amd import update as U

    monkeypatch.setenv("STSTHIP_VIRTUAL_STRIPS", strips)
    rng = np.random.default_rng(77)
    shape = (700, 333)
    cells = np.zeros(shape, dtype=U.HOTSPOT_CELL)
    cells["temp"] = 320 + 10 * rng.random(shape, dtype=np.float32)
    cells["power"] = rng.random(shape, dtype=np.float32) * 0.01
    p = oracle.hotspot_params(*shape)
    po, pc, fcells = fdtd_setup(oracle, 300, 222)
    for n in (8, 16, 17, 24, 33, 64):  # HotSpot: 1, 2, 3, 3, 5, 8 passes of up to 8 generations
        want = oracle.hotspot(p, cells, n, n_threads=8)
        for switch in ("1", "0"):
            monkeypatch.setenv("STSTHIP_SKIP_CONSTANT_STORES", switch)
            # poison the pool's next blocks so that a plane that is wrongly not stored cannot look right
            junk = U.Grid.from_numpy(np.full(shape, 7.5, dtype=np.float32))
            del junk
            got = run_hip(U.hotspot(p.Rx_1, p.Ry_1, p.Rz_1, p.Cap_1, split_cell_structure=True), cells, n)
            assert np.array_equal(bits(got), bits(want)), f"hotspot n={n} switch={switch}"
    for n in (6, 12, 13, 18, 31):  # FDTD: 1, 2, 3, 3, 6 passes of up to 6 generations
        want = oracle.fdtd(po, fcells, n, n_threads=8)
        for switch in ("1", "0"):
            monkeypatch.setenv("STSTHIP_SKIP_CONSTANT_STORES", switch)
            got = run_hip(U.fdtd(pc, True), fcells, n, halo=np.zeros((), dtype=U.FDTD_CELL))
            assert np.array_equal(bits(got), bits(want)), f"fdtd n={n} switch={switch}"


RANDOM_CASES = list(range(48))


@pytest.mark.parametrize("case", RANDOM_CASES)
def test_random_configurations(gpu, oracle, case):
    """Seeded random draws over application, layout, grid shape (ragged widths, single rows and columns, widths
    around the strip widths of the kernels), generation count (every mix of launch depths), iteration offset and
    halo value: the HIP path equals the oracle bit for bit."""
    from stencilstream_amd import update as U

    rng = np.random.default_rng(0xA11CE + case)
    kind = ["jacobi5", "jacobi_variant", "hotspot", "conway", "fdtd", "jacobi_uniform"][case % 6]
    widths = [1, 2, 3, 5, 47, 48, 49, 112, 167, 168, 169, 239, 240, 241, 256, 333, 480, 700, 1024]
    H = int(rng.choice([1, 2, 3, 7, 24, 25, 100, 133, 134, 300, 611]))
    W = int(rng.choice(widths))
    if kind == "conway":
        W = int(rng.choice([4, 8, 236, 240, 244, 960, 1028, 3, 5, 241]))
    n = int(rng.integers(0, 45))
    offset = int(rng.choice([0, 0, 1, 7, 1000]))
    what = f"{kind} {H}x{W} n={n} offset={offset}"
    if kind in ("jacobi5", "jacobi_uniform"):
        grid = rng.random((H, W), dtype=np.float32) * 4 - 1
        coef = [0.3] * 5 if kind == "jacobi_uniform" else list(rng.random(5, dtype=np.float32) * 0.4)
        halo = 0.0 if kind == "jacobi_uniform" else float(np.float32(rng.random() * 2 - 1))
        got = run_hip(U.jacobi("Jacobi5General", coef), grid, n, halo=np.float32(halo), offset=offset)
        want = oracle.jacobi("Jacobi5General", coef, grid, n, halo=halo, iteration_offset=offset, n_threads=8)
    elif kind == "jacobi_variant":
        variant, ncoef = [("Jacobi1General", 1), ("Jacobi3Constant", 0), ("Jacobi5Constant", 0), ("Jacobi4General", 4),
                          ("Jacobi9General", 9)][int(rng.integers(0, 5))]
        grid = rng.random((H, W), dtype=np.float32)
        coef = list(rng.random(ncoef, dtype=np.float32) * 0.2)
        halo = float(np.float32(rng.random()))
        what += " " + variant
        got = run_hip(U.jacobi(variant, coef), grid, n, halo=np.float32(halo), offset=offset)
        want = oracle.jacobi(variant, coef, grid, n, halo=halo, iteration_offset=offset, n_threads=8)
    elif kind == "hotspot":
        cells = np.zeros((H, W), dtype=U.HOTSPOT_CELL)
        cells["temp"] = 300 + 40 * rng.random((H, W), dtype=np.float32)
        cells["power"] = rng.random((H, W), dtype=np.float32) * 0.02
        p = oracle.hotspot_params(H, W)
        split = bool(rng.integers(0, 2))
        what += f" split={split}"
        got = run_hip(U.hotspot(p.Rx_1, p.Ry_1, p.Rz_1, p.Cap_1, split_cell_structure=split), cells, n, offset=offset)
        want = oracle.hotspot(p, cells, n, iteration_offset=offset, n_threads=8)
    elif kind == "conway":
        grid = (rng.random((H, W)) < rng.choice([0.2, 0.37, 0.6])).astype(np.uint8)
        got = run_hip(U.conway(), grid, n, offset=offset)
        want = oracle.conway(grid, n, n_threads=8)
    else:
        H, W = max(H, 8), max(W, 8)
        po, pc, cells = fdtd_setup(oracle, H, W)
        for f in ("ex", "ey", "hz"):
            cells[f] = (rng.random((H, W), dtype=np.float32) - 0.5) * 1e-3
        split = bool(rng.integers(0, 2))
        what += f" {H}x{W} split={split}"
        got = run_hip(U.fdtd(pc, split), cells, n, halo=np.zeros((), dtype=U.FDTD_CELL), offset=offset)
        want = oracle.fdtd(po, cells, n, iteration_offset=offset, n_threads=8)
    assert np.array_equal(bits(got), bits(want)), what


def test_conway_full_size_word_form(gpu, oracle, monkeypatch):
    """Game of Life at 16384^2 (the size of the headline grid): the word form (four cells per 32-bit word, two
    row strips, 8 generations per launch) equals the byte-per-register kernel bit for bit; windows of the
    result equal the oracle on the window plus a margin of n cells; and a period-2 pattern (blinkers on a
    dead background) comes back after an even number of generations."""
    import torch

    from stencilstream_amd import capi

    N, n = 16384, 20
    gen = torch.Generator(device="cuda").manual_seed(7)
    src = (torch.rand(N, N, device=gpu, generator=gen) < 0.37).to(torch.uint8)
    words, cells = torch.empty_like(src), torch.empty_like(src)
    dom = capi.Domain(N, N, 0, N, N)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    monkeypatch.setenv("STSTHIP_CONWAY_FASTPATH", "1")
    capi.app_run("conway", capi.NoParams(), b"\0", dom, [src.data_ptr()], [words.data_ptr()], 0, n, blocking=True,
                 stream=s.cuda_stream)
    monkeypatch.setenv("STSTHIP_CONWAY_FASTPATH", "0")
    capi.app_run("conway", capi.NoParams(), b"\0", dom, [src.data_ptr()], [cells.data_ptr()], 0, n, blocking=True,
                 stream=s.cuda_stream)
    assert torch.equal(words, cells)
    assert 0 < int(words.sum()) < N * N and int(words.max()) == 1

    m, w = n, 80
    for r0, c0 in ((0, 0), (0, N - w), (N - w, 0), (N - w, N - w), (N // 2, N // 2), (6553, 0), (0, 9830),
                   (N * 2 // 5 - 40, 4000)):  # the last one straddles the boundary of the two row strips
        ra, rb = max(0, r0 - m), min(N, r0 + w + m)
        ca, cb = max(0, c0 - m), min(N, c0 + w + m)
        ref = oracle.conway(src[ra:rb, ca:cb].cpu().numpy(), n, n_threads=8)
        got = words[r0:r0 + w, c0:c0 + w].cpu().numpy()
        assert np.array_equal(got, ref[r0 - ra:r0 - ra + w, c0 - ca:c0 - ca + w]), (r0, c0)

    monkeypatch.setenv("STSTHIP_CONWAY_FASTPATH", "1")
    blink = torch.zeros(N, N, dtype=torch.uint8, device=gpu)
    blink[2::8, 1::8] = 1
    blink[2::8, 2::8] = 1
    blink[2::8, 3::8] = 1
    out = torch.empty_like(blink)
    capi.app_run("conway", capi.NoParams(), b"\0", dom, [blink.data_ptr()], [out.data_ptr()], 0, 38, blocking=True,
                 stream=s.cuda_stream)
    assert torch.equal(out, blink)
    capi.app_run("conway", capi.NoParams(), b"\0", dom, [blink.data_ptr()], [out.data_ptr()], 0, 7, blocking=True,
                 stream=s.cuda_stream)
    assert not torch.equal(out, blink) and int(out.sum()) == int(blink.sum())


def test_baseline_size_properties(gpu, oracle):
    """BASELINE config 1 size (Jacobi5General 16384^2, distinct coefficients: the general kernel at its compiled
    depth, SweepTuning's max_generations): (1) temporal blocking is invisible -- the default path (two launches of the
    deepest depth on two row strips) equals the 1-generation-per-launch path bit for bit; (2) windows of the
    result (grid corners, edges, interior, the rim of the initial square) equal the oracle run on the
    window plus a margin of n cells (cells further than n from the window border cannot be influenced
    by what lies outside it)."""
    import torch

    from stencilstream_amd import capi

    N, n = 16384, 16
    p = capi.JacobiParams()
    coef = [0.2, 0.21, 0.19, 0.22, 0.18]
    for i, c in enumerate(coef):
        p.coef[i] = c
    gen = torch.Generator(device="cuda").manual_seed(5)
    src = torch.rand(N, N, device=gpu, generator=gen)
    src[N // 4:3 * N // 4, N // 4:3 * N // 4] += 1.0
    deep, flat = torch.empty_like(src), torch.empty_like(src)
    dom = capi.Domain(N, N, 0, N, N)
    halo = np.float32(0.125).tobytes()
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    capi.app_run("jacobi5general", p, halo, dom, [src.data_ptr()], [deep.data_ptr()], 0, n, blocking=True,
                 stream=s.cuda_stream)
    # reference run: one generation per launch, one strip (the default run above used the deepest compiled depth
    # and two row strips on two streams)
    os.environ["STSTHIP_MAX_GENERATIONS"] = "1"
    os.environ["STSTHIP_VIRTUAL_STRIPS"] = "1"
    try:
        info = capi.app_run("jacobi5general", p, halo, dom, [src.data_ptr()], [flat.data_ptr()], 0, n,
                            blocking=True, stream=s.cuda_stream)
    finally:
        del os.environ["STSTHIP_MAX_GENERATIONS"]
        del os.environ["STSTHIP_VIRTUAL_STRIPS"]
    assert info.n_launches == n
    assert torch.equal(deep, flat)

    m, w = n, 96
    for r0, c0 in ((0, 0), (0, N - w), (N - w, 0), (N - w, N - w), (N // 2, N // 2), (N // 4 - 40, N // 4 - 40),
                   (5000, 0), (0, 9000), (N - w, 12345), (3 * N // 4 - 50, 7000)):
        ra, rb = max(0, r0 - m), min(N, r0 + w + m)
        ca, cb = max(0, c0 - m), min(N, c0 + w + m)
        window = src[ra:rb, ca:cb].cpu().numpy()
        # outside the window the oracle sees the halo value; at true grid edges that is exact
        ref = oracle.jacobi("Jacobi5General", coef, window, n, halo=0.125)
        got = deep[r0:r0 + w, c0:c0 + w].cpu().numpy()
        want = ref[r0 - ra:r0 - ra + w, c0 - ca:c0 - ca + w]
        assert np.array_equal(bits(got), bits(want)), (r0, c0)


def test_largest_baseline_grid_windows(gpu, oracle):
    """BASELINE config 5 grid size, Jacobi5General 65536^2 (2^32 cells, 16 GiB per buffer) on one GPU:
    element offsets exceed 32 bits.  Windows at the far corners / edges and inside must equal the oracle on
    the window plus a margin of n cells; the uniform-coefficient path (product-carrying kernels at their compiled
    depth -- 16 generations per launch --, two row strips) must equal the general kernel fed with the same
    coefficients through STSTHIP_JACOBI_FASTPATH=0, checksum of per-slab checksums."""
    import torch

    from stencilstream_amd import capi

    N, n = 65536, 25  # depths 16 + 8 + 1 (uniform) and 8 + 8 + 8 + 1 (general)
    free, _total = torch.cuda.mem_get_info()
    if free < 60 * 2 ** 30:
        pytest.skip("needs 3 x 16 GiB of HBM")
    p = capi.JacobiParams()
    for i in range(5):
        p.coef[i] = 0.2
    gen = torch.Generator(device="cuda").manual_seed(11)
    src = torch.empty(N, N, device=gpu)
    for r in range(0, N, 8192):  # slabs keep the generator's temporaries small
        src[r:r + 8192] = torch.rand(8192, N, device=gpu, generator=gen)
    dst = torch.empty_like(src)
    dom = capi.Domain(N, N, 0, N, N)
    halo = np.float32(0.0).tobytes()
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    info = capi.app_run("jacobi5general", p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 3, n, blocking=True,
                        stream=s.cuda_stream)
    assert info.n_launches >= 3
    m, w = n, 64
    spots = ((0, 0), (0, N - w), (N - w, 0), (N - w, N - w), (N // 2, N // 2), (N - w, 40000), (33333, N - w),
             (32768 - 32, 32768 - 32), (26214 - 32, 1000))  # 26214 = the boundary between the two row strips
    for r0, c0 in spots:
        ra, rb = max(0, r0 - m), min(N, r0 + w + m)
        ca, cb = max(0, c0 - m), min(N, c0 + w + m)
        window = src[ra:rb, ca:cb].cpu().numpy()
        ref = oracle.jacobi("Jacobi5General", [0.2] * 5, window, n, halo=0.0)
        got = dst[r0:r0 + w, c0:c0 + w].cpu().numpy()
        want = ref[r0 - ra:r0 - ra + w, c0 - ca:c0 - ca + w]
        assert np.array_equal(bits(got), bits(want)), (r0, c0)
    # checksum of per-slab checksums against the general kernel (9 flops per cell, other launch depths)
    os.environ["STSTHIP_JACOBI_FASTPATH"] = "0"
    try:
        sums = [dst[r:r + 8192].view(torch.int32).sum(dtype=torch.int64).item() for r in range(0, N, 8192)]
        capi.app_run("jacobi5general", p, halo, dom, [src.data_ptr()], [dst.data_ptr()], 3, n, blocking=True,
                     stream=s.cuda_stream)
        again = [dst[r:r + 8192].view(torch.int32).sum(dtype=torch.int64).item() for r in range(0, N, 8192)]
    finally:
        del os.environ["STSTHIP_JACOBI_FASTPATH"]
    assert sums == again
    del src, dst
    torch.cuda.empty_cache()
    capi.load().ststhip_pool_trim()


def test_strip_driver_at_eight_gpu_shape(gpu, oracle):
    """The per-GPU shape of BASELINE config 5 (65536 columns, 8192-row strips) through the multi-GPU strip
    driver with one rank: far-column windows against the oracle."""
    import torch

    from stencilstream_amd import capi
    from stencilstream_amd.dist import StripDomain

    H, W, n = 8192, 65536, 13
    p = capi.JacobiParams()
    coef = [0.2, 0.21, 0.19, 0.22, 0.18]
    for i, c in enumerate(coef):
        p.coef[i] = c
    gen = torch.Generator(device="cuda").manual_seed(12)
    grid = torch.rand(H, W, device=gpu, generator=gen)
    strip = StripDomain("jacobi5general", p, np.float32(0.25).tobytes(), H, W, 0, 1, gpu, sub_strips=2)
    strip.load_owned(grid)
    strip.advance(0, n)
    out = strip.owned(0, torch.float32)
    m, w = n, 64
    for r0, c0 in ((0, 0), (0, W - w), (H - w, W - w), (H * 2 // 5 - 32, 30000), (4000, W // 2)):
        ra, rb = max(0, r0 - m), min(H, r0 + w + m)
        ca, cb = max(0, c0 - m), min(W, c0 + w + m)
        ref = oracle.jacobi("Jacobi5General", coef, grid[ra:rb, ca:cb].cpu().numpy(), n, halo=0.25)
        got = out[r0:r0 + w, c0:c0 + w].cpu().numpy()
        assert np.array_equal(bits(got), bits(ref[r0 - ra:r0 - ra + w, c0 - ca:c0 - ca + w])), (r0, c0)


@pytest.mark.parametrize("strips,skew", [(2, 400), (2, 900), (3, 500), (5, 500)])
def test_virtual_strips_small_grid(gpu, oracle, monkeypatch, strips, skew):
    """The pass driver's row strips on separate streams (normally only used for tall grids), forced on
    a small ragged grid: results must not depend on the strip layout."""
    from stencilstream_amd import update as U

    monkeypatch.setenv("STSTHIP_VIRTUAL_STRIPS", str(strips))
    monkeypatch.setenv("STSTHIP_STRIP_SKEW_PERMILLE", str(skew))
    rng = np.random.default_rng(strips * 10 + skew)
    grid = rng.random((701, 333), dtype=np.float32)
    coef = [0.2, 0.21, 0.19, 0.22, 0.18]
    for n in (8, 21, 64):
        got = run_hip(U.jacobi("Jacobi5General", coef), grid, n, halo=np.float32(0.3))
        want = oracle.jacobi("Jacobi5General", coef, grid, n, halo=0.3, n_threads=8)
        assert np.array_equal(bits(got), bits(want)), f"n={n}"
    life = (rng.random((640, 300)) < 0.4).astype(np.uint8)
    assert np.array_equal(run_hip(U.conway(), life, 40), oracle.conway(life, 40, n_threads=8))


def _two_rank_worker(rank, world, port, result_dir):
    """One of two processes that share cuda:0: real HIP sweeps, real streams and events, ghost rows
    through host memory over gloo (RCCL cannot put two ranks on one GPU)."""
    import sys

    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stencilstream_amd import capi
    from stencilstream_amd.dist import StripDomain

    torch.cuda.set_device(0)
    capi.init(0)
    H, W = 700, 900
    grid = np.random.default_rng(99).random((H, W), dtype=np.float32)
    p = capi.JacobiParams()
    for i, c in enumerate([0.2, 0.21, 0.19, 0.22, 0.18]):
        p.coef[i] = c
    strip = StripDomain("jacobi5general", p, np.float32(0.5).tobytes(), H, W, rank, world, "cuda:0",
                        exchange_via_host=True)
    strip.load_owned(torch.from_numpy(grid[strip.row_begin:strip.row_end].copy()).cuda())
    strip.advance(0, 21)
    strip.advance(21, 8)
    np.save(os.path.join(result_dir, f"rank{rank}.npy"), strip.owned(0, torch.float32).cpu().numpy())
    # uniform coefficients and zero halo: the product-carrying kernels, ghosts travel in product space
    for i in range(5):
        p.coef[i] = 0.2
    fast = StripDomain("jacobi5general", p, np.float32(0.0).tobytes(), H, W, rank, world, "cuda:0",
                       exchange_via_host=True, sub_strips=2)  # two sub-strips per rank on two streams
    assert fast.sweep.uniform is not None
    fast.load_owned(torch.from_numpy(grid[fast.row_begin:fast.row_end].copy()).cuda())
    fast.advance(0, 21)
    fast.advance(21, 1)
    np.save(os.path.join(result_dir, f"fast{rank}.npy"), fast.owned(0, torch.float32).cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu(gpu, oracle, tmp_path):
    """The multi-GPU driver end to end with two processes (boundary-first ordering, second stream,
    ragged depths 8+8+4+1 and a resume), both on cuda:0."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.get_context("spawn")
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    grid = np.random.default_rng(99).random((700, 900), dtype=np.float32)
    want = oracle.jacobi("Jacobi5General", [0.2, 0.21, 0.19, 0.22, 0.18], grid, 29, halo=0.5, n_threads=8)
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(2)], axis=0)
    assert np.array_equal(bits(got), bits(want))
    want = oracle.jacobi("Jacobi5General", [0.2] * 5, grid, 22, halo=0.0, n_threads=8)
    got = np.concatenate([np.load(tmp_path / f"fast{r}.npy") for r in range(2)], axis=0)
    assert np.array_equal(bits(got), bits(want))


def test_empty_and_degenerate_grids(gpu, oracle):
    """Edge cases of the boundary: zero rows / zero columns, zero generations, one cell."""
    import torch

    from stencilstream_amd import capi, update as U

    p = capi.JacobiParams()
    halo = np.float32(0).tobytes()
    a = torch.zeros(16, device=gpu)
    b = torch.ones(16, device=gpu)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    for H, W in ((0, 0), (0, 4), (4, 0)):
        info = capi.app_run("jacobi5general", p, halo, capi.Domain(H, W, 0, H, max(W, 1)), [a.data_ptr()],
                            [b.data_ptr()], 0, 5, blocking=True, stream=s.cuda_stream)
        assert info.n_processed_cells == 0
        assert torch.equal(b.cpu(), torch.ones(16))  # nothing written
    # zero generations copy the source
    src = torch.arange(12, dtype=torch.float32, device=gpu)
    dst = torch.zeros(12, device=gpu)
    torch.cuda.synchronize()
    capi.app_run("jacobi5general", p, halo, capi.Domain(3, 4, 0, 3, 4), [src.data_ptr()], [dst.data_ptr()], 0, 0,
                 blocking=True, stream=s.cuda_stream)
    assert torch.equal(dst.cpu(), torch.arange(12, dtype=torch.float32))
    # a pitch larger than the width: the padding columns are neither read as cells nor written
    H, W, pitch = 37, 50, 64
    rng = np.random.default_rng(3)
    grid = rng.random((H, W), dtype=np.float32)
    padded = np.full((H, pitch), np.nan, dtype=np.float32)
    padded[:, :W] = grid
    psrc = torch.from_numpy(padded).to(gpu)
    pdst = torch.full((H, pitch), -7.0, device=gpu)
    for i, c in enumerate([0.2, 0.21, 0.19, 0.22, 0.18]):
        p.coef[i] = c
    torch.cuda.synchronize()
    capi.app_run("jacobi5general", p, halo, capi.Domain(H, W, 0, H, pitch), [psrc.data_ptr()], [pdst.data_ptr()],
                 0, 9, blocking=True, stream=s.cuda_stream)
    out = pdst.cpu().numpy()
    want = oracle.jacobi("Jacobi5General", [0.2, 0.21, 0.19, 0.22, 0.18], grid, 9, halo=0.0)
    assert np.array_equal(bits(out[:, :W]), bits(want))
    assert (out[:, W:] == -7.0).all()


@pytest.mark.parametrize("shape", [(1, 1), (64, 64), (257, 511), (700, 300)], ids=str)
def test_jacobi_uniform_coefficient_fast_path(gpu, oracle, monkeypatch, shape):
    """Five equal positive coefficients and a +0 halo switch ststhip_app_run to the product-carrying
    form (5 instead of 9 flops per cell).  It must stay bit-identical to Jacobi5General for every
    number of generations (first / last / only launch variants) and equal the path with the switch off."""
    from stencilstream_amd import update as U

    rng = np.random.default_rng(shape[0] + shape[1])
    grid = rng.random(shape, dtype=np.float32) * 3 - 1  # negative values too
    for c in (0.2, 0.25, 1.0, 3.0e-3):
        coef = [c] * 5
        for n in (1, 2, 7, 8, 9, 16, 17, 41):
            want = oracle.jacobi("Jacobi5General", coef, grid, n, halo=0.0, n_threads=8)
            got = run_hip(U.jacobi("Jacobi5General", coef), grid, n, halo=np.float32(0.0))
            assert np.array_equal(bits(got), bits(want)), f"c={c} n={n}"
        monkeypatch.setenv("STSTHIP_JACOBI_FASTPATH", "0")
        got = run_hip(U.jacobi("Jacobi5General", coef), grid, 17, halo=np.float32(0.0))
        monkeypatch.delenv("STSTHIP_JACOBI_FASTPATH")
        assert np.array_equal(bits(got), bits(oracle.jacobi("Jacobi5General", coef, grid, 17, halo=0.0)))
    # resume across calls: 8 + 9 generations with an offset
    coef = [0.2] * 5
    a = run_hip(U.jacobi("Jacobi5General", coef), grid, 8, halo=np.float32(0.0), offset=3)
    b = run_hip(U.jacobi("Jacobi5General", coef), a, 9, halo=np.float32(0.0), offset=11)
    assert np.array_equal(bits(b), bits(oracle.jacobi("Jacobi5General", coef, grid, 17, halo=0.0)))
    # conditions not met (halo != 0, -0 halo, negative or unequal coefficients): general kernel, still exact
    for coef, halo in (([0.2] * 5, 0.5), ([0.2] * 5, -0.0), ([-0.2] * 5, 0.0), ([0.2, 0.2, 0.2, 0.2, 0.25], 0.0)):
        got = run_hip(U.jacobi("Jacobi5General", coef), grid, 9, halo=np.float32(halo))
        want = oracle.jacobi("Jacobi5General", coef, grid, 9, halo=halo, n_threads=8)
        assert np.array_equal(bits(got), bits(want)), (coef, halo)


def test_baseline_size_uniform_form_equals_general_kernel(gpu, monkeypatch):
    """BASELINE config 1 exactly (16384^2, coefficients 5 x 0.2, halo 0, centred-square init): the
    product-carrying kernel the benchmark runs and the general nine-flop kernel give the same bits."""
    import torch

    from stencilstream_amd import capi

    N, n = 16384, 25  # 12 + 12 + 1 generations: middle/first/last launch variants
    p = capi.JacobiParams()
    for i in range(5):
        p.coef[i] = 0.2
    r = torch.arange(N, device=gpu)
    inside = (r >= N // 4) & (r < 3 * N // 4)
    src = (inside[:, None] & inside[None, :]).to(torch.float32).contiguous()
    src += torch.rand(N, N, device=gpu, generator=torch.Generator(device="cuda").manual_seed(1)) * 1e-3
    fast, general = torch.empty_like(src), torch.empty_like(src)
    dom = capi.Domain(N, N, 0, N, N)
    halo = np.float32(0.0).tobytes()
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    capi.app_run("jacobi5general", p, halo, dom, [src.data_ptr()], [fast.data_ptr()], 0, n, blocking=True,
                 stream=s.cuda_stream)
    monkeypatch.setenv("STSTHIP_JACOBI_FASTPATH", "0")
    capi.app_run("jacobi5general", p, halo, dom, [src.data_ptr()], [general.data_ptr()], 0, n, blocking=True,
                 stream=s.cuda_stream)
    assert torch.equal(fast, general)
    assert float(fast.sum()) > 0


@pytest.mark.parametrize("split", [True, False], ids=["planes", "aos"])
def test_hotspot_fp64_bit_exact(gpu, oracle, split):
    """The HotSpot formula evaluated in fp64 (BASELINE.json names an fp64 HotSpot; the reference itself
    is fp32, examples/hotspot/hotspot.cpp:38).  Same expression in the oracle and on the GPU."""
    from stencilstream_amd import update as U

    rng = np.random.default_rng(64)
    shape = (300, 517)
    cells = np.zeros(shape, dtype=U.HOTSPOT_CELL_F64)
    cells["temp"] = 320 + 10 * rng.random(shape)
    cells["power"] = rng.random(shape) * 0.01
    p32 = oracle.hotspot_params(*shape)
    vals = [float(p32.Rx_1), float(p32.Ry_1), float(p32.Rz_1), float(p32.Cap_1)]
    po = oracle.HotspotParamsF64(*vals)
    for n in (1, 8, 23):
        got = run_hip(U.hotspot_f64(*vals, split_cell_structure=split), cells, n)
        want = oracle.hotspot_f64(po, cells, n, n_threads=8)
        assert np.array_equal(bits(got), bits(want)), f"n={n}"
    assert np.abs(want["temp"] - cells["temp"]).max() > 0


@pytest.mark.parametrize("shape", [(1, 1), (2, 9), (64, 64), (130, 517), (700, 300)], ids=str)
def test_jacobi25_radius2_bit_exact(gpu, oracle, shape):
    """The dense 5 x 5 Jacobi of radius 2 ("jacobi25general", an extra: SURVEY 8(f)4): two-cell halos in rows and
    columns, lane shifts by two cells, the shallow depth the tuning rule picks for radius >= 2, the narrow form on
    these small grids; against the oracle's restatement of the same loop (rows, then columns, from 0.0f)."""
    from stencilstream_amd import capi, update as U

    rng = np.random.default_rng(25 + shape[1])
    grid = rng.random(shape, dtype=np.float32)
    coef = (rng.random(25, dtype=np.float32) / 12).astype(np.float32)
    p = capi.Jacobi25Params()
    for i in range(25):
        p.coef[i] = float(coef[i])
    info = capi.app_info("jacobi25general")
    assert info.stencil_radius == 2 and info.halo_depth_per_generation == 2
    tf = U.TransitionFunction("jacobi25general", p, np.dtype("<f4"))
    for n in (1, 2, 7):
        got = run_hip(tf, grid, n, halo=np.float32(0.25))
        want = oracle.jacobi25(coef, grid, n, halo=0.25, n_threads=8)
        assert np.array_equal(bits(got), bits(want)), f"n={n}"


def test_narrow_form_equals_default_shape(gpu, oracle, monkeypatch):
    """Grids of up to 6 M cells (here: the knob set to 20 M) are swept with one cell per lane (NarrowForm, hip/internal/Sweep.hpp): the same
    bits as the default shape, for the general, the product-carrying and the Game of Life kernels."""
    import torch

    from stencilstream_amd import capi

    capi.init(0)
    N = 1500
    gen = torch.Generator(device="cuda").manual_seed(3)
    src = torch.rand(N, N, device=gpu, generator=gen)
    dom = capi.Domain(N, N, 0, N, N)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    for coef, halo in (([0.2, 0.21, 0.19, 0.22, 0.18], 0.5), ([0.2] * 5, 0.0)):
        p = capi.JacobiParams()
        for i, c in enumerate(coef):
            p.coef[i] = c
        outs = []
        for kcells in ("20000", "0"):
            monkeypatch.setenv("STSTHIP_NARROW_FORM_KCELLS", kcells)
            out = torch.empty_like(src)
            capi.app_run("jacobi5general", p, np.float32(halo).tobytes(), dom, [src.data_ptr()], [out.data_ptr()], 0,
                         29, blocking=True, stream=s.cuda_stream)
            outs.append(out)
        # the threshold is read once per kernel instantiation: compare with the oracle instead of relying on the switch
        want = oracle.jacobi("Jacobi5General", coef, src.cpu().numpy(), 29, halo=halo, n_threads=8)
        for out in outs:
            assert np.array_equal(bits(out.cpu().numpy()), bits(want))


def test_pool_hands_blocks_over_in_stream_order(gpu):
    """ststhip_free_async: a block released while work is queued on stream A goes to stream A again at once, and to
    stream B only behind the release event (the stream-agnostic free list was a data race)."""
    import ctypes as C

    import torch

    from stencilstream_amd import capi

    capi.init(0)
    lib = capi.load()
    n = 512 << 20
    s1, s2 = C.c_void_p(), C.c_void_p()
    capi.check(lib.ststhip_stream_create(C.byref(s1)), "stream")
    capi.check(lib.ststhip_stream_create(C.byref(s2)), "stream")
    p, q, r = C.c_void_p(), C.c_void_p(), C.c_void_p()
    capi.check(lib.ststhip_malloc_async(C.byref(p), n, s1), "malloc")
    for _ in range(8):  # a queue of work on s1 that still uses the block when it is released
        capi.check(lib.ststhip_memset(p, 1, n, s1), "memset")
    capi.check(lib.ststhip_free_async(p, s1), "free")
    capi.check(lib.ststhip_malloc_async(C.byref(q), n, s2), "malloc")  # the same bucket: the block comes back
    assert q.value == p.value
    capi.check(lib.ststhip_memset(q, 2, n, s2), "memset")  # must land after the eight memsets of s1
    capi.check(lib.ststhip_stream_synchronize(s2), "sync")
    capi.check(lib.ststhip_stream_synchronize(s1), "sync")
    host = torch.empty(n, dtype=torch.uint8)
    capi.check(lib.ststhip_memcpy_d2h(C.c_void_p(host.data_ptr()), q, n, s2), "d2h")
    capi.check(lib.ststhip_stream_synchronize(s2), "sync")
    assert int(host.min()) == 2 and int(host.max()) == 2
    # host-ordered allocation of a block whose release event is pending: waits for it, then owns it
    capi.check(lib.ststhip_memset(q, 3, n, s2), "memset")
    capi.check(lib.ststhip_free_async(q, s2), "free")
    capi.check(lib.ststhip_malloc(C.byref(r), n), "malloc")
    assert r.value == q.value
    capi.check(lib.ststhip_memcpy_d2h(C.c_void_p(host.data_ptr()), r, n, None), "d2h")
    capi.check(lib.ststhip_stream_synchronize(None), "sync")
    assert int(host.min()) == 3 and int(host.max()) == 3
    capi.check(lib.ststhip_free(r), "free")
    capi.check(lib.ststhip_stream_destroy(s1), "stream")
    capi.check(lib.ststhip_stream_destroy(s2), "stream")


@pytest.mark.parametrize("strips", ["2", "3", "4"])
def test_row_strips_with_bands_beside_interiors_full_size(gpu, monkeypatch, strips):
    """The pass driver's row strips at BASELINE size with ragged launch depths (40 = 12 + 12 + 12 + 3 + 1 and
    8 + 8 + 8 + 8 + 8): the boundary bands run on streams of their own beside the interiors, so the orderings between
    them are events, not stream order -- a missing one (the interior of a shallower pass overwriting rows the
    neighbour's previous band still reads) only shows on launches long enough to overlap.  Several repetitions
    against the one-strip result, bit for bit."""
    import torch

    from stencilstream_amd import capi

    capi.init(0)
    N = 16384
    gen = torch.Generator(device="cuda").manual_seed(int(strips))
    src = torch.rand(N, N, device=gpu, generator=gen)
    dom = capi.Domain(N, N, 0, N, N)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    for coef in ([0.2] * 5, [0.2, 0.21, 0.19, 0.22, 0.18]):
        p = capi.JacobiParams()
        for i, c in enumerate(coef):
            p.coef[i] = c
        monkeypatch.setenv("STSTHIP_VIRTUAL_STRIPS", "1")
        want = torch.empty_like(src)
        capi.app_run("jacobi5general", p, np.float32(0).tobytes(), dom, [src.data_ptr()], [want.data_ptr()], 0, 40,
                     blocking=True, stream=s.cuda_stream)
        monkeypatch.setenv("STSTHIP_VIRTUAL_STRIPS", strips)
        got = torch.empty_like(src)
        for _ in range(4):
            got.zero_()
            torch.cuda.synchronize()  # cleared on torch's stream, swept on `s`
            capi.app_run("jacobi5general", p, np.float32(0).tobytes(), dom, [src.data_ptr()], [got.data_ptr()], 0, 40,
                         blocking=True, stream=s.cuda_stream)
            assert torch.equal(got, want)


def test_staged_sweeps_describe_themselves(gpu):
    """The staged sweep is the default of the precompiled applications (Sweep.hpp: the waves of a workgroup share one
    column strip as a pipeline over the levels); the independent-wave shapes stay registered as A/B baselines."""
    from stencilstream_amd import capi

    capi.init(0)
    for app, stages in (("jacobi5uniform", 4), ("jacobi5general", 4), ("hotspot", 4), ("hotspot_aos", 4),
                        ("fdtd_coef_grouped", 4), ("fdtd_coef_aos", 4), ("conway_packed", 4),
                        ("jacobi5general_independent", 1), ("hotspot_independent", 1)):
        assert capi.app_info(app).stages == stages, app
    # a wave's share of its workgroup's strip is what the drivers count waves with
    info = capi.app_info("jacobi5uniform")
    assert info.strip_width == (64 * info.cells_per_lane - 2 * info.max_generations) // 4


@pytest.mark.parametrize("shape", [(1, 1), (3, 70), (37, 257), (300, 700), (1100, 2300)], ids=str)
def test_staged_depths_and_ragged_batches(gpu, oracle, shape):
    """Generation counts that use every compiled depth of the staged kernels (16 + 4 + 2 and 8 + 4 + 2 + 1 with four, four,
    two and one stage; HotSpot's 12 + 6 + 3 + 1 with four, three, three, one) on grids whose row counts are not multiples of the batch
    length, narrower than one strip, and with halo values that differ from the cells: uniform-coefficient Jacobi,
    general Jacobi, HotSpot on planes and as AoS (two cells per lane), against the oracle bit for bit."""
    from stencilstream_amd import update as U

    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    grid = rng.random(shape, dtype=np.float32)
    for coef, halo in (([0.2] * 5, 0.0), ([0.1, 0.2, 0.3, 0.25, 0.15], 0.375)):
        for n in (22, 31):
            got = run_hip(U.jacobi("Jacobi5General", coef), grid, n, halo=np.float32(halo))
            want = oracle.jacobi("Jacobi5General", coef, grid, n, halo=halo, n_threads=8)
            assert np.array_equal(bits(got), bits(want)), f"jacobi {coef[0]} n={n}"
    cells = np.zeros(shape, dtype=U.HOTSPOT_CELL)
    cells["temp"] = 320 + 10 * rng.random(shape, dtype=np.float32)
    cells["power"] = rng.random(shape, dtype=np.float32) * 0.01
    p = oracle.hotspot_params(*shape)
    for split in (True, False):
        for n in (22, 31):
            got = run_hip(U.hotspot(p.Rx_1, p.Ry_1, p.Rz_1, p.Cap_1, split_cell_structure=split), cells, n)
            want = oracle.hotspot(p, cells, n, n_threads=8)
            assert np.array_equal(bits(got), bits(want)), f"hotspot split={split} n={n}"


def test_staged_sweep_race_screen(gpu):
    """The stages of a workgroup hand rows to each other through LDS between barriers; a missing wait there shows as
    a few wrong cells in one launch of several hundred (it did, once: Sweep.hpp keeps an explicit s_waitcnt in front
    of the barrier since).  Repeated long runs at full size -- thousands of workgroups per launch, hundreds of
    launches -- must equal the independent-wave kernels' results bit for bit, every time."""
    import torch

    from stencilstream_amd import capi

    capi.init(0)
    s = torch.cuda.Stream()
    H = W = 8192
    gen = torch.Generator(device="cuda").manual_seed(5)
    temp = 320 + 10 * torch.rand(H, W, device=gpu, generator=gen)
    power = 0.01 * torch.rand(H, W, device=gpu, generator=gen)
    dom = capi.Domain(H, W, 0, H, W)
    p = oracle_free_hotspot_params(H)

    def run(app, planes, n):
        out = [torch.zeros_like(t) for t in planes]
        torch.cuda.synchronize()
        capi.app_run(app, p, bytes(8), dom, [t.data_ptr() for t in planes], [t.data_ptr() for t in out], 0, n,
                     blocking=True, stream=s.cuda_stream)
        return out

    want = run("hotspot_independent", [temp, power], 240)
    for trial in range(12):
        got = run("hotspot", [temp, power], 240)
        assert torch.equal(got[0].view(torch.int32), want[0].view(torch.int32)), f"hotspot, trial {trial}"
    del want, got
    q = capi.JacobiParams()
    for i, c in enumerate([0.2, 0.21, 0.19, 0.22, 0.18]):
        q.coef[i] = c
    want = run_jacobi(capi, s, "jacobi5general_independent", q, temp, 240)
    for trial in range(12):
        got = run_jacobi(capi, s, "jacobi5general", q, temp, 240)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), f"jacobi, trial {trial}"


def test_staged_sweep_race_screen_headline_kernel(gpu):
    """The same screen for the kernel bench.py times: the uniform-coefficient form of Jacobi5General, sixteen
    generations per launch on four stages of four levels -- the deepest pipeline per stage in the library -- at the
    headline size 16384^2 with random data.  240 generations = the `_first` kernel, thirteen launches of the middle one
    and the `_last` one (two row strips with their bands side by side, as in the timed path); 16 generations = the
    `_only` kernel.  Sixteen runs each must equal, bit for bit, the independent-wave general-coefficient kernel (no
    stages, no LDS, nine flops per cell, eight generations per launch) on the same input."""
    import torch

    from stencilstream_amd import capi

    capi.init(0)
    s = torch.cuda.Stream()
    H = W = 16384
    grid = torch.rand(H, W, device=gpu, generator=torch.Generator(device="cuda").manual_seed(16))
    q = capi.JacobiParams()
    for i in range(5):
        q.coef[i] = 0.2
    assert capi.app_info("jacobi5uniform").max_generations == 16 and capi.app_info("jacobi5uniform").stages == 4
    for n, trials in ((240, 16), (16, 16), (250, 4)):  # 250 = 15 x 16 + 8 + 2: the halved depths as well
        want = run_jacobi(capi, s, "jacobi5general_independent", q, grid, n)
        for trial in range(trials):
            got = run_jacobi(capi, s, "jacobi5general", q, grid, n)  # five equal coefficients: the uniform form
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), f"{n} generations, trial {trial}"
        del want, got


def run_jacobi(capi, stream, app, params, grid, n):
    import torch

    out = torch.zeros_like(grid)
    torch.cuda.synchronize()
    H, W = grid.shape
    capi.app_run(app, params, np.float32(0).tobytes(), capi.Domain(H, W, 0, H, W), [grid.data_ptr()], [out.data_ptr()],
                 0, n, blocking=True, stream=stream.cuda_stream)
    return out


def oracle_free_hotspot_params(n):
    """examples/hotspot/hotspot.cpp:281-295 with numpy's float32 / float64 (the binding's parameter block)."""
    from stencilstream_amd import capi

    f32, f64 = np.float32, np.float64
    t_chip, chip = f32(0.0005), f32(0.016)
    gh, gw = f32(chip / f32(n)), f32(chip / f32(n))
    cap = f32(f64(0.5) * f64(1.75e6) * f64(t_chip) * f64(gh) * f64(gw))
    rx = f32(f64(gw) / (f64(2.0) * 100 * f64(t_chip) * f64(gh)))
    ry = f32(f64(gh) / (f64(2.0) * 100 * f64(t_chip) * f64(gw)))
    rz = f32(t_chip / f32(f32(f32(100) * gh) * gw))
    max_slope = f32(f64(3.0e6) / (f64(0.5) * f64(t_chip) * f64(1.75e6)))
    step = f32(f64(0.001) / f64(max_slope) / f64(1000.0))
    return capi.HotspotParams(float(f32(1) / rx), float(f32(1) / ry), float(f32(1) / rz), float(step / cap))


@pytest.mark.parametrize("split", [True, False], ids=["planes", "aos"])
def test_hotspot_512_reference_data_bit_exact(gpu, oracle, split):
    """The reference's 512 x 512 HotSpot data set (tests/golden/hotspot_{temp,power}_512.txt.gz; the oracle is pinned on
    it by Rodinia's own program, tests/test_oracle_golden.py): HIP against the oracle, bit for bit, 2000 iterations as
    in the fixture and 37 (ragged launch depths)."""
    import gzip

    from stencilstream_amd import update as U

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cells = np.zeros((512, 512), dtype=U.HOTSPOT_CELL)
    cells["temp"] = np.loadtxt(gzip.open(os.path.join(golden, "hotspot_temp_512.txt.gz"), "rt"), dtype=np.float32).reshape(512, 512)
    cells["power"] = np.loadtxt(gzip.open(os.path.join(golden, "hotspot_power_512.txt.gz"), "rt"), dtype=np.float32).reshape(512, 512)
    p = oracle.hotspot_params(512, 512)
    for n in (2000, 37):
        got = run_hip(U.hotspot(p.Rx_1, p.Ry_1, p.Rz_1, p.Cap_1, split_cell_structure=split), cells, n)
        want = oracle.hotspot(p, cells, n, n_threads=8)
        assert np.array_equal(bits(got), bits(want)), f"n={n}"


def test_depth_by_measurement(gpu):
    """Families compiled deeper than their rule trusts (one-word cells: sixteen generations beside the trusted eight)
    have their depth MEASURED by the pass driver on the first long call for a grid shape (ststhip_sweep_desc::
    alt_generations): the call that probes, a later call that uses the kept depth and a short call that runs the
    trusted depth must all equal the one-generation-per-launch run bit for bit (within the stated tolerance for the
    flavour built with fused multiply-adds); the choice is readable."""
    import ctypes as C

    import torch

    from stencilstream_amd import capi

    capi.init(0)
    lib = capi.load()
    info = capi.app_info("jacobi5general")
    assert info.max_generations == 16 and info.default_generations == 8
    s = torch.cuda.Stream()
    H, W = 5000, 6000  # large enough for two row strips and their bands
    grid = torch.rand(H, W, device=gpu, generator=torch.Generator(device="cuda").manual_seed(88))
    q = capi.JacobiParams()
    for i, c in enumerate([0.2, 0.21, 0.19, 0.22, 0.18]):
        q.coef[i] = c
    for app in ("jacobi5general", "jacobi5general_fma"):
        n = 200  # >= 6 x 16: the first call probes
        os.environ["STSTHIP_MAX_GENERATIONS"] = "1"
        try:
            want = run_jacobi(capi, s, app, q, grid, n)
            want_short = run_jacobi(capi, s, app, q, grid, 37)
        finally:
            del os.environ["STSTHIP_MAX_GENERATIONS"]
        # (the flavour compiled with fused multiply-adds promises a tolerance, not bits: the compiler may fuse
        # differently in every instantiation, and the depths differ by an ulp here and there)
        if app.endswith("_fma"):
            same = lambda a, b: float((a.double() - b.double()).abs().max()) <= 1e-5  # noqa: E731
        else:
            same = lambda a, b: torch.equal(a.view(torch.int32), b.view(torch.int32))  # noqa: E731
        first = run_jacobi(capi, s, app, q, grid, n)
        assert same(first, want), f"{app}: the probing call"
        second = run_jacobi(capi, s, app, q, grid, n)
        assert same(second, want), f"{app}: the call after it"
        short = run_jacobi(capi, s, app, q, grid, 37)
        assert same(short, want_short), f"{app}: a short call"
        # the choice is kept per kernel family and grid shape
        assert capi.app_tuned_depth(app, H, W) in (8, 16), app
        assert capi.app_tuned_depth(app, H, W + 1) == 0


def test_no_scratch_at_any_registered_depth(gpu):
    """No kernel of the shipped library uses scratch (private memory), at any compiled depth of any registered
    transition function, in any of its forms (default shape, narrow form, constant-plane variant): the code object is
    asked (ststhip_app_scratch_bytes).  Round 3's library had 64-80 bytes per lane in some shallow depths (the kernel's
    by-value argument block copied to private memory where the optimiser could not take it apart; the device code reads
    its arguments where they lie since)."""
    from stencilstream_amd import capi

    capi.init(0)
    checked, offenders = 0, []
    for name in capi.list_apps():
        if name.startswith("x_"):
            continue  # registered tuning experiments of an EXPERIMENTS=1 build
        if name.startswith("selfcheck"):
            continue  # the reference's self-checking test function: 25 comparisons per neighbour, correctness only
        depth = capi.app_info(name).max_generations
        while depth >= 1:
            b = capi.app_scratch_bytes(name, depth)
            checked += 1
            if b:
                offenders.append((name, depth, b))
            depth //= 2
    assert checked >= 100
    assert not offenders, offenders


@pytest.mark.parametrize("strips", ["2", "3"])
def test_row_strips_with_moving_boundaries_start_over(gpu, oracle, monkeypatch, strips):
    """The pass driver's row strips with boundaries that move up by a launch's ghost rows per pass (ststhip_options::
    skewed_strips): 700 generations on 2048 rows move them further than their span (a quarter of a strip at most), so they
    start over in the middle of the call -- the one pass in which every strip also waits for the one below it.  Two and
    three strips, general coefficients (8 generations per launch) and the uniform form (16), against the oracle and
    against the scheme with fixed strips and boundary bands."""
    from stencilstream_amd import update as U

    monkeypatch.setenv("STSTHIP_VIRTUAL_STRIPS", strips)
    monkeypatch.setenv("STSTHIP_NARROW_FORM_KCELLS", "0")
    rng = np.random.default_rng(int(strips))
    grid = rng.random((2048, 1024), dtype=np.float32)
    for coef in ([0.11, 0.19, 0.23, 0.31, 0.16], [0.2] * 5):
        want = oracle.jacobi("Jacobi5General", coef, grid, 700, halo=0.0, n_threads=8)
        monkeypatch.setenv("STSTHIP_SKEWED_STRIPS", "1")
        moving = run_hip(U.jacobi("Jacobi5General", coef), grid, 700, halo=np.float32(0.0))
        monkeypatch.setenv("STSTHIP_SKEWED_STRIPS", "0")
        fixed = run_hip(U.jacobi("Jacobi5General", coef), grid, 700, halo=np.float32(0.0))
        assert np.array_equal(bits(moving), bits(want)), coef
        assert np.array_equal(bits(fixed), bits(want)), coef


@pytest.mark.parametrize("strips,rows", [("2", 2050), ("3", 2052)])
def test_row_strips_with_moving_boundaries_while_the_depth_is_probed(gpu, oracle, monkeypatch, strips, rows):
    """... and in a call that times its own first passes at two depths (launches of 16, 16, 16, 8, 8, 8, 8 generations and
    then the winner's): the boundaries move by the LARGER of a pass's ghost rows and the ones of the pass before, whichever
    way the depth changes.  A grid shape no other test uses, so that the call probes."""
    from stencilstream_amd import capi
    from stencilstream_amd import update as U

    monkeypatch.setenv("STSTHIP_VIRTUAL_STRIPS", strips)
    monkeypatch.setenv("STSTHIP_NARROW_FORM_KCELLS", "0")
    monkeypatch.setenv("STSTHIP_SKEWED_STRIPS", "1")
    rng = np.random.default_rng(rows)
    grid = rng.random((rows, 1024), dtype=np.float32)
    coef = [0.12, 0.18, 0.24, 0.3, 0.16]
    assert capi.app_tuned_depth("jacobi5general", rows, 1024) == 0
    got = run_hip(U.jacobi("Jacobi5General", coef), grid, 300, halo=np.float32(0.0))
    assert capi.app_tuned_depth("jacobi5general", rows, 1024) in (8, 16), "the call did not probe"
    want = oracle.jacobi("Jacobi5General", coef, grid, 300, halo=0.0, n_threads=8)
    assert np.array_equal(bits(got), bits(want))
