"""The oracle against every known answer we have for this path (CPU only).

Pins: (1) the reference's own self-checking known-answer test of the update machinery
(tests/cpu/StencilUpdate.cpp:35-41 with tests/TransFuncs.hpp:55-104), restated; (2) the known
answers of the reference's cpu backend recorded in SURVEY.md section 8c; (3) Rodinia's OpenMP
HotSpot built from the reference tree when it is present."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

KNOWN = json.load(open(os.path.join(GOLDEN, "known_answers.json")))


def load_conway():
    chars = [ch for ch in open(os.path.join(GOLDEN, KNOWN["conway"]["input"])).read() if ch in "X."]
    return (np.array(chars) == "X").astype(np.uint8).reshape(64, 64)


def conway_stdout(grid):
    return "".join("".join("X" if v else "." for v in row) + "\n" for row in grid)


def load_hotspot(O):
    t = np.loadtxt(os.path.join(GOLDEN, KNOWN["hotspot_64"]["temp"]), dtype=np.float32).reshape(64, 64)
    p = np.loadtxt(os.path.join(GOLDEN, KNOWN["hotspot_64"]["power"]), dtype=np.float32).reshape(64, 64)
    cells = np.zeros((64, 64), dtype=O.HOTSPOT_CELL)
    cells["temp"], cells["power"] = t, p
    return cells


def test_conway_known_answers(oracle):
    g = load_conway()
    assert int(g.sum()) == KNOWN["conway"]["live_cells"]["0"]
    for n, md5 in KNOWN["conway"]["stdout_md5"].items():
        out = oracle.conway(g, int(n))
        assert hashlib.md5(conway_stdout(out).encode()).hexdigest() == md5
    assert int(oracle.conway(g, 100).sum()) == KNOWN["conway"]["live_cells"]["100"]


@pytest.mark.parametrize("case", KNOWN["jacobi5general"], ids=lambda c: f"{c['n']}x{c['iterations']}")
def test_jacobi_known_answers(oracle, case):
    n = case["n"]
    if case.get("fma_build"):
        if not oracle.cpu_has_fma():
            pytest.skip("known answer from an FMA build; this CPU has no FMA")
        with oracle.use_fma_build():
            out = oracle.jacobi("Jacobi5General", case["coef"], oracle.jacobi_init(n, n), case["iterations"], n_threads=8)
        plain = oracle.jacobi("Jacobi5General", case["coef"], oracle.jacobi_init(n, n), case["iterations"], n_threads=8)
        assert np.abs(plain.view(np.int32).astype(np.int64) - out.view(np.int32)).max() <= 64  # a few ulp
    else:
        out = oracle.jacobi("Jacobi5General", case["coef"], oracle.jacobi_init(n, n), case["iterations"], n_threads=8)
    assert abs(out.astype(np.float64).sum() - case["sum"]) <= 5e-7 * case["sum"]
    for where, value in case["at"].items():
        r, c = map(int, where.split(","))
        assert out[r, c] == np.float32(value)


def test_hotspot_known_answers(oracle):
    cells = load_hotspot(oracle)
    out = oracle.hotspot(oracle.hotspot_params(64, 64), cells, KNOWN["hotspot_64"]["iterations"])
    lines = [f"{i}\t{v:g}" for i, v in enumerate(out["temp"].reshape(-1)[:3])]
    assert lines == KNOWN["hotspot_64"]["first_output_lines"]
    assert np.array_equal(out["power"], cells["power"])


@pytest.mark.parametrize("case", KNOWN["selfcheck_cases"], ids=str)
@pytest.mark.parametrize("radius", [1, 2])
def test_selfcheck_known_answer(oracle, case, radius):
    H, W, offset, n = case
    out = oracle.selfcheck(radius, oracle.selfcheck_input(H, W, offset), offset, n)
    # tests/StencilUpdateTest.hpp:53-62
    assert (out["r"] == np.arange(H)[:, None]).all() and (out["c"] == np.arange(W)[None, :]).all()
    assert (out["i_iteration"] == offset + n).all() and (out["i_subiteration"] == 0).all()
    assert (out["status"] == 0).all()


def test_selfcheck_detects_wrong_halo(oracle):
    # the function must poison cells when the machinery is wrong: feed a wrong iteration index
    out = oracle.selfcheck(1, oracle.selfcheck_input(16, 16, 3), 4, 1)
    assert (out["status"] == 1).all()


def test_zero_iterations_and_empty(oracle):
    g = np.random.default_rng(1).random((5, 7), dtype=np.float32)
    assert np.array_equal(oracle.jacobi("Jacobi5General", [0.2] * 5, g, 0), g)
    e = np.zeros((0, 7), dtype=np.float32)
    assert oracle.jacobi("Jacobi5General", [0.2] * 5, e, 3).shape == (0, 7)


def test_halo_is_reimposed_every_sweep(oracle):
    # 1x1 grid, Jacobi4Constant: (N+W+S+E)/4 with halo 2.0 stays 2.0 from the first generation on
    g = np.array([[7.0]], dtype=np.float32)
    for n in (1, 2, 5):
        assert oracle.jacobi("Jacobi4Constant", [], g, n, halo=2.0)[0, 0] == np.float32(2.0)


def test_iteration_offset_is_a_resume(oracle):
    # running 3+4 generations with an offset equals 7 generations (selfcheck verifies indices itself)
    a = oracle.selfcheck(1, oracle.selfcheck_input(20, 33, 10), 10, 3)
    b = oracle.selfcheck(1, a, 13, 4)
    c = oracle.selfcheck(1, oracle.selfcheck_input(20, 33, 10), 10, 7)
    assert np.array_equal(b, c) and (c["status"] == 0).all()


def test_threads_do_not_change_results(oracle):
    g = np.random.default_rng(2).random((67, 129), dtype=np.float32)
    a = oracle.jacobi("Jacobi9General", np.linspace(0.05, 0.2, 9), g, 6, n_threads=1)
    b = oracle.jacobi("Jacobi9General", np.linspace(0.05, 0.2, 9), g, 6, n_threads=8)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_hotspot_against_rodinia_openmp(oracle, tmp_path):
    """Second opinion from the reference tree: examples/hotspot/hotspot_openmp.cpp (third-party
    Rodinia code, different operation order) built unchanged into oracle/_ref."""
    exe = os.path.join(ROOT, "oracle", "_ref", "hotspot_openmp")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/hotspot_openmp not built (reference tree absent)")
    out_file = tmp_path / "out.txt"
    subprocess.check_call(
        [exe, "64", "64", "100", "4", os.path.join(GOLDEN, "hotspot_temp_64.txt"),
         os.path.join(GOLDEN, "hotspot_power_64.txt"), str(out_file)],
        stdout=subprocess.DEVNULL,
    )
    ref = np.loadtxt(out_file, dtype=np.float64)[:, 1].reshape(64, 64)
    out = oracle.hotspot(oracle.hotspot_params(64, 64), load_hotspot(oracle), 100)["temp"]
    assert np.abs(out.astype(np.float64) - ref).max() <= 2 * KNOWN["hotspot_64"]["rodinia_openmp_max_abs_diff"]


def load_hotspot_512(oracle):
    import gzip

    t = np.loadtxt(gzip.open(os.path.join(GOLDEN, "hotspot_temp_512.txt.gz"), "rt"), dtype=np.float32).reshape(512, 512)
    p = np.loadtxt(gzip.open(os.path.join(GOLDEN, "hotspot_power_512.txt.gz"), "rt"), dtype=np.float32).reshape(512, 512)
    cells = np.zeros((512, 512), dtype=oracle.HOTSPOT_CELL)
    cells["temp"], cells["power"] = t, p
    return cells


def test_hotspot_512_against_rodinia_fixture(oracle):
    """The reference's larger HotSpot data set (examples/hotspot/data/{temp,power}_512) against the committed output
    of Rodinia's OpenMP HotSpot built unchanged from the reference tree (tests/golden/make_hotspot_512_fixture.py):
    a second reference-compiled pin of the oracle's HotSpot restatement.  The data set sits near its steady state:
    2000 iterations move every temperature by 0.061 K.  Rodinia is another program (other operation order, six
    printed digits): the two agree to 0.0003 K on average -- half a per cent of the movement --, to 0.002 K on more
    than 99.8 % of the cells; the rest are a patch at the top edge where the update of the reference's expression
    order rounds to no change in fp32 and Rodinia's order does not."""
    fixture = np.load(os.path.join(GOLDEN, "hotspot_512_rodinia_2000.npz"))
    cells = load_hotspot_512(oracle)
    out = oracle.hotspot(oracle.hotspot_params(512, 512), cells, int(fixture["iterations"]), n_threads=8)["temp"]
    diff = np.abs(out.astype(np.float64) - fixture["temp"].astype(np.float64))
    moved = np.abs(out.astype(np.float64) - cells["temp"].astype(np.float64))
    assert np.median(moved) > 0.05  # the iterations changed the field by far more than the agreement asserted below
    assert diff.mean() <= 0.0006, diff.mean()
    assert (diff <= 0.002).mean() >= 0.998, (diff <= 0.002).mean()
    assert diff.max() <= 0.07, diff.max()


def test_fdtd_oracle_against_unchanged_reference_functor(oracle):
    """Pins the oracle's FDTD restatement (transition function, TDV, and the example's set-up
    arithmetic restated in tests/fdtd_harness.py): tests/golden/fdtd/*.csv were written by the
    reference's unchanged examples/fdtd sources (tests/golden/make_fdtd_golden.py says how).  The frames
    carry 6 significant digits (default ostream formatting, fdtd.cpp:118-161), so agreement is required
    to half a unit of the sixth digit; 1029 generations x 2 sub-iterations, three update calls with an
    iteration offset, source cut-off and detection window inside the run, a lossy outer ring."""
    import json

    from fdtd_harness import Experiment, load_csv

    here = os.path.join(GOLDEN, "fdtd")
    ex = Experiment(json.load(open(os.path.join(here, "experiment.json"))))
    stdout = open(os.path.join(here, "stdout.txt")).read()
    assert f"grid w/h          = {ex.grid_width()} cells" in stdout
    assert f"n. timesteps      = {ex.n_timesteps()}" in stdout
    assert f"n. snap timesteps = {ex.n_snap_timesteps()}" in stdout
    assert f"dt                = {float(ex.dt()):.6g} s/iteration" in stdout

    params = oracle.FdtdParams()
    for k, v in ex.kernel_constants().items():
        setattr(params, k, v if isinstance(v, int) else float(v))
    cells = ex.initial_grid(oracle.FDTD_CELL)
    assert len(np.unique(cells["cb"])) == 3  # two rings and the metal around them

    def agrees(got, path):
        want = load_csv(path)
        assert want.shape == got.shape
        # a printed value is the nearest 6-significant-digit decimal of the float
        tol = 0.5000001 * 10.0 ** (np.floor(np.log10(np.maximum(np.abs(want), 1e-300))) - 5)
        worst = np.max(np.abs(got.astype(np.float64) - want) / tol)
        assert worst <= 1.0, (path, worst)

    calls = ex.update_calls()
    assert [c[2] for c in calls] == [343, 686, 1029]
    for offset, n, label in calls:
        cells = oracle.fdtd(params, cells, n, iteration_offset=offset, n_threads=4)
        assert np.abs(cells["hz"]).max() > 0
        agrees(cells["hz"], os.path.join(here, f"hz.{label}.csv"))
    assert np.abs(cells["hz_sum"]).max() > 0
    agrees(cells["hz_sum"], os.path.join(here, f"hz_sum.{ex.n_timesteps()}.csv"))


def test_jacobi25_oracle_against_numpy():
    """The radius-2 dense Jacobi of the oracle (an extra, stencil_oracle.c) against the same loop in numpy float32:
    rows then columns, sum from 0.0f, halo outside the grid."""
    from oracle import oracle as O

    rng = np.random.default_rng(5)
    grid = rng.random((9, 11), dtype=np.float32)
    coef = (rng.random(25, dtype=np.float32) / 10).astype(np.float32)
    got = O.jacobi25(coef, grid, 2, halo=0.5)
    cur = grid
    for _ in range(2):
        padded = np.full((13, 15), np.float32(0.5), dtype=np.float32)
        padded[2:-2, 2:-2] = cur
        nxt = np.zeros_like(cur)
        for r in range(9):
            for c in range(11):
                s = np.float32(0)
                for dr in range(5):
                    for dc in range(5):
                        s = np.float32(s + np.float32(coef[dr * 5 + dc] * padded[r + dr, c + dc]))
                nxt[r, c] = s
        cur = nxt
    assert np.array_equal(got.view(np.uint32), cur.view(np.uint32))


def test_convection_oracle_against_unchanged_reference_functors(oracle, tmp_path):
    """The oracle's restatement of the two convection kernels (stencil_oracle.c: pseudo_transient_fn,
    thermal_solver_fn) against the reference's own source (examples/convection/convection.cpp:76-242) compiled by g++
    on this repository's stencil::cpu backend (build/examples/convection_dump_cpu): two time steps of 30 pseudo-
    transient iterations (three sub-iterations each) and one thermal step (two), all eleven fp64 fields of every cell
    compared as bits.  The reference's tests hold no vector for these kernels; this is what pins the restatement."""
    exe = os.path.join(ROOT, "build", "examples", "convection_dump_cpu")
    if not os.path.exists(exe):
        pytest.skip("build/examples/convection_dump_cpu not built (reference tree absent)")
    subprocess.check_call([exe, "40", "30", str(tmp_path)], stdout=subprocess.DEVNULL)
    differing, cells, moved = oracle.convection_dump_check(str(tmp_path), 30)
    assert cells == 120 * 40 and moved > 1e-6
    assert differing == {"round0_pt": 0, "round0_ts": 0, "round1_pt": 0, "round1_ts": 0}, differing
