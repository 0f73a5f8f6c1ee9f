"""The reference's example applications, built UNCHANGED from the reference tree against this
repository's headers (examples/Makefile -> build/examples/), run end to end.

*_cpu binaries use stencil::cpu (host), *_hip binaries stencil::cuda == stencil::hip (MI355X).
The binaries are build products (git-ignored); tests skip when they are absent (the reference tree
only exists in the build container, the binaries travel to the GPU box)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

EX = os.path.join(ROOT, "build", "examples")
KNOWN = json.load(open(os.path.join(GOLDEN, "known_answers.json")))


def exe(name):
    path = os.path.join(EX, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not built (needs the reference tree; run `make -C examples`)")
    return path


def run(cmd, **kw):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    return subprocess.run(cmd, check=True, capture_output=True, env=env, timeout=600, **kw)


def conway_text(grid):
    return "".join("".join("X" if v else "." for v in row) + "\n" for row in grid)


def conway_case(binary, grid, n):
    out = run([binary, str(grid.shape[0]), str(grid.shape[1]), str(n)], input=conway_text(grid).encode()).stdout.decode()
    rows = [line for line in out.splitlines() if line and set(line) <= {"X", "."}]
    return (np.array([list(r) for r in rows]) == "X").astype(np.uint8)


def gosper():
    chars = [ch for ch in open(os.path.join(GOLDEN, KNOWN["conway"]["input"])).read() if ch in "X."]
    return (np.array(chars) == "X").astype(np.uint8).reshape(64, 64)


# ------------------------------------------------------------------ CPU backend (BASELINE config 0)
def test_conway_cpu_known_answers():
    binary = exe("conway_cpu")
    for n, md5 in KNOWN["conway"]["stdout_md5"].items():
        out = run([binary, "64", "64", n], input=open(os.path.join(GOLDEN, KNOWN["conway"]["input"]), "rb").read())
        assert hashlib.md5(out.stdout).hexdigest() == md5


def test_conway_cpu_512x512_100_generations(oracle):
    """BASELINE.json configs[0]: 512x512, 100 generations, cpu backend, bit-exact."""
    binary = exe("conway_cpu")
    grid = np.zeros((512, 512), dtype=np.uint8)
    grid[:64, :64] = gosper()  # the reference's input at the NW corner
    assert np.array_equal(conway_case(binary, grid, 100), oracle.conway(grid, 100, n_threads=8))
    # Bernoulli(0.35) from a fixed seed
    rnd = (np.random.default_rng(0xC0FFEE).random((512, 512)) < 0.35).astype(np.uint8)
    assert np.array_equal(conway_case(binary, rnd, 100), oracle.conway(rnd, 100, n_threads=8))


def test_hotspot_cpu_reference_input(tmp_path, oracle):
    binary = exe("hotspot_cpu")
    out_file = tmp_path / "out.txt"
    res = run([binary, "64", "64", "100", os.path.join(GOLDEN, "hotspot_temp_64.txt"),
               os.path.join(GOLDEN, "hotspot_power_64.txt"), str(out_file)])
    assert b"Walltime:" in res.stdout and b"GFlops:" in res.stdout
    lines = open(out_file).read().splitlines()
    assert lines[:3] == KNOWN["hotspot_64"]["first_output_lines"]
    t = np.loadtxt(os.path.join(GOLDEN, "hotspot_temp_64.txt"), dtype=np.float32).reshape(64, 64)
    p = np.loadtxt(os.path.join(GOLDEN, "hotspot_power_64.txt"), dtype=np.float32).reshape(64, 64)
    cells = np.zeros((64, 64), dtype=oracle.HOTSPOT_CELL)
    cells["temp"], cells["power"] = t, p
    want = oracle.hotspot(oracle.hotspot_params(64, 64), cells, 100)["temp"].reshape(-1)
    assert lines == [f"{i}\t{v:g}" for i, v in enumerate(want)]


# ------------------------------------------------------------------ MI355X backend
@pytest.mark.gpu
def test_conway_hip(oracle):
    binary = exe("conway_hip")
    for n, md5 in KNOWN["conway"]["stdout_md5"].items():
        out = run([binary, "64", "64", n], input=open(os.path.join(GOLDEN, KNOWN["conway"]["input"]), "rb").read())
        assert hashlib.md5(out.stdout).hexdigest() == md5
    rnd = (np.random.default_rng(0xC0FFEE).random((512, 512)) < 0.35).astype(np.uint8)
    assert np.array_equal(conway_case(binary, rnd, 100), oracle.conway(rnd, 100, n_threads=8))


JACOBI_COEFS = {"Jacobi1General": 1, "Jacobi2Constant": 0, "Jacobi3Constant": 0, "Jacobi4Constant": 0,
                "Jacobi5Constant": 0, "Jacobi4General": 4, "Jacobi5General": 5, "Jacobi9General": 9}


@pytest.mark.gpu
@pytest.mark.parametrize("variant", sorted(JACOBI_COEFS))
def test_jacobi_hip_bit_exact(tmp_path, oracle, variant):
    """examples/jacobi/jacobi.cpp unchanged: argv, centred-square init, raw fp32 output."""
    binary = exe(f"jacobi_{variant}_hip")
    n = JACOBI_COEFS[variant]
    coef = [f"{0.2 - 0.01 * i:.4f}" for i in range(n)]
    H, W, its = 300, 700, 29
    out_file = tmp_path / "out.bin"
    res = run([binary, str(H), str(W), str(its), str(out_file)] + coef)
    assert b"Walltime:" in res.stdout
    got = np.fromfile(out_file, dtype=np.float32).reshape(H, W)
    want = oracle.jacobi(variant, [float(c) for c in coef], oracle.jacobi_init(H, W), its, halo=0.0, n_threads=8)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_jacobi_strips_example_single_rank(tmp_path, oracle):
    """examples/jacobi_strips.cpp (hip::StripUpdate with the reference's Jacobi5General; one process per GPU over RCCL):
    as one rank its output file equals the oracle -- and so the unchanged single-GPU example's.  More ranks need a GPU
    each; the same driver with several strips is covered by tests/test_cpp_api.py (mailbox exchange)."""
    binary = exe("jacobi_strips_hip")
    coef = ["0.2", "0.21", "0.19", "0.22", "0.18"]
    H, W, its = 300, 700, 29
    out_file = tmp_path / "out.bin"
    res = run([binary, str(H), str(W), str(its), str(out_file)] + coef)
    assert b"Walltime:" in res.stdout and b"Strips: 1" in res.stdout
    got = np.fromfile(out_file, dtype=np.float32).reshape(H, W)
    want = oracle.jacobi("Jacobi5General", [float(c) for c in coef], oracle.jacobi_init(H, W), its, halo=0.0, n_threads=8)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_jacobi_hip_show_config():
    out = run([exe("jacobi_Jacobi5General_hip"), "show-config"]).stdout.decode()
    cfg = json.loads(out)
    assert cfg == {"variant": "cuda", "n_coefficients": 5, "n_operations": 9}


@pytest.mark.gpu
@pytest.mark.parametrize("binary_name", ["hotspot_hip", "hotspot_aos_hip"])
def test_hotspot_hip_bit_exact(tmp_path, oracle, binary_name):
    binary = exe(binary_name)
    n, its = 256, 50
    # synthetic inputs of examples/hotspot/data/input_gen.jl:3-15
    temp = np.full((n, n), 30.0, dtype=np.float32)
    power = np.zeros((n, n), dtype=np.float32)
    lo, hi = n // 4 - 1, 3 * n // 4  # 1-based inclusive n/4 .. 3n/4
    power[lo:hi, lo:hi] = 0.5
    temp.tofile(tmp_path / "temp.bin")
    power.tofile(tmp_path / "power.bin")
    run([binary, str(n), str(n), str(its), str(tmp_path / "temp.bin"), str(tmp_path / "power.bin"),
         str(tmp_path / "out.bin")])
    got = np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(n, n)
    cells = np.zeros((n, n), dtype=oracle.HOTSPOT_CELL)
    cells["temp"], cells["power"] = temp, power
    want = oracle.hotspot(oracle.hotspot_params(n, n), cells, its, n_threads=8)["temp"]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("its", [40, 333])
def test_jacobi_hip_follows_its_upload_block_by_block(tmp_path, oracle, its):
    """The unchanged example with its grid uploaded in row blocks and the first passes run as row tiles behind them
    (hip/Grid.hpp start_upload_in_blocks, ststhip_set_source_arrival; blocks of 1 MiB so that a small grid has
    eight): the output file equals the oracle's, and the run with one copy in front of the first pass."""
    binary = exe("jacobi_Jacobi5General_hip")
    coef = ["0.2", "0.21", "0.19", "0.22", "0.18"]
    H, W = 2048, 1024
    want = oracle.jacobi("Jacobi5General", [float(c) for c in coef], oracle.jacobi_init(H, W), its, halo=0.0, n_threads=8)
    for mode in ({"STSTHIP_UPLOAD_BLOCK_MIB": "1"}, {"STSTHIP_STREAM_UPLOAD": "0"}):
        out_file = tmp_path / "out.bin"
        res = subprocess.run([binary, str(H), str(W), str(its), str(out_file)] + coef, check=True, capture_output=True,
                             env=dict(os.environ, STSTHIP_TRACE_STREAM="1", **mode), timeout=600)
        followed = b"[ststhip] block 7" in res.stderr
        assert followed == ("STSTHIP_UPLOAD_BLOCK_MIB" in mode), res.stderr.decode()[-400:]
        got = np.fromfile(out_file, dtype=np.float32).reshape(H, W)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), mode


@pytest.mark.gpu
@pytest.mark.parametrize("binary_name", ["hotspot_hip", "hotspot_aos_hip"])
def test_hotspot_hip_follows_its_upload_block_by_block(tmp_path, oracle, binary_name):
    """... and with per-field planes: every block is scattered into the planes behind its copy."""
    binary = exe(binary_name)
    n, its = 1024, 150
    rng = np.random.default_rng(9)
    temp = (30 + rng.random((n, n), dtype=np.float32)).astype(np.float32)
    power = (rng.random((n, n), dtype=np.float32) * 0.5).astype(np.float32)
    temp.tofile(tmp_path / "temp.bin")
    power.tofile(tmp_path / "power.bin")
    res = subprocess.run([binary, str(n), str(n), str(its), str(tmp_path / "temp.bin"), str(tmp_path / "power.bin"),
                          str(tmp_path / "out.bin")], check=True, capture_output=True,
                         env=dict(os.environ, STSTHIP_TRACE_STREAM="1", STSTHIP_UPLOAD_BLOCK_MIB="1"), timeout=600)
    assert b"[ststhip] block 3" in res.stderr, res.stderr.decode()[-400:]
    got = np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(n, n)
    cells = np.zeros((n, n), dtype=oracle.HOTSPOT_CELL)
    cells["temp"], cells["power"] = temp, power
    want = oracle.hotspot(oracle.hotspot_params(n, n), cells, its, n_threads=8)["temp"]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ------------------------------------------------------------------ the reference's own GPU compile mode (fused multiply-adds)
# The reference's CMake targets give its GPU builds no -ffp-contract flag (CMakeLists.txt:46-51): a*b+c is fused there, and
# north_star asks for "stated fp32 tolerance" for Jacobi and HotSpot, not bit-equality.  The *_hip_fma binaries are the
# same unchanged sources compiled that way (examples/Makefile); they are held against the plain oracle with the tolerances
# SURVEY 8(c) states: Jacobi abs <= 1e-5 after 1000 generations, HotSpot rel <= 1e-5 (the Jacobi bound is restated as
# 2e-5 below, with the measurement that says why), and against the oracle compiled with fused multiply-adds as well.
@pytest.mark.gpu
@pytest.mark.parametrize("variant,coef", [("Jacobi5General", ["0.2", "0.2", "0.2", "0.2", "0.2"]),
                                          ("Jacobi5General", ["0.2", "0.21", "0.19", "0.22", "0.18"]),
                                          ("Jacobi9General", [f"{0.1 + 0.002 * i:.3f}" for i in range(9)])],
                         ids=["5pt-uniform", "5pt-distinct", "9pt"])
def test_jacobi_hip_fma_within_stated_tolerance(tmp_path, oracle, variant, coef):
    binary = exe(f"jacobi_{variant}_hip_fma")
    H, W, its = 600, 700, 1000
    out_file = tmp_path / "out.bin"
    res = run([binary, str(H), str(W), str(its), str(out_file)] + coef)
    assert b"Walltime:" in res.stdout
    got = np.fromfile(out_file, dtype=np.float32).reshape(H, W)
    want = oracle.jacobi(variant, [float(c) for c in coef], oracle.jacobi_init(H, W), its, halo=0.0, n_threads=8)
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    # SURVEY 8(c) states abs <= 1e-5 after 1000 generations on the premise that the rounding differences do not
    # accumulate.  They do a little where the coefficients' sum rounds above one (5 x 0.2f = 1.0000000149): measured
    # 1.32e-5 for that case on this backend AND between the oracle's own plain and fused builds on the host (the same
    # number to the last digit), 1.5e-6 for distinct coefficients.  Stated bound: 2e-5.
    assert err <= 2e-5, err
    assert np.isfinite(got).all() and got.max() <= 1.0 + 1e-6 and got.min() >= 0.0
    # ... and against the oracle built the same way (oracle/Makefile: liboracle_fma.so, gcc -mfma -ffp-contract=fast:
    # the same source, multiply-adds fused): the GPU build fuses the same operations
    if oracle.cpu_has_fma():
        with oracle.use_fma_build():
            fused = oracle.jacobi(variant, [float(c) for c in coef], oracle.jacobi_init(H, W), its, halo=0.0, n_threads=8)
        err_fused = float(np.abs(got.astype(np.float64) - fused.astype(np.float64)).max())
        assert err_fused <= 1e-6, err_fused
        if variant == "Jacobi5General":
            assert np.array_equal(got.view(np.uint32), fused.view(np.uint32)), "five-point: bit-identical to the fused oracle"


@pytest.mark.gpu
@pytest.mark.parametrize("binary_name", ["hotspot_hip_fma", "hotspot_aos_hip_fma"])
def test_hotspot_hip_fma_within_stated_tolerance(tmp_path, oracle, binary_name):
    binary = exe(binary_name)
    n, its = 512, 1000
    temp = np.full((n, n), 30.0, dtype=np.float32)
    power = np.zeros((n, n), dtype=np.float32)
    lo, hi = n // 4 - 1, 3 * n // 4
    power[lo:hi, lo:hi] = 0.5
    temp.tofile(tmp_path / "temp.bin")
    power.tofile(tmp_path / "power.bin")
    run([binary, str(n), str(n), str(its), str(tmp_path / "temp.bin"), str(tmp_path / "power.bin"),
         str(tmp_path / "out.bin")])
    got = np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(n, n)
    cells = np.zeros((n, n), dtype=oracle.HOTSPOT_CELL)
    cells["temp"], cells["power"] = temp, power
    want = oracle.hotspot(oracle.hotspot_params(n, n), cells, its, n_threads=8)["temp"]
    rel = float((np.abs(got.astype(np.float64) - want) / np.abs(want)).max())
    assert rel <= 1e-5, rel  # rel <= 1e-5
    assert not np.array_equal(temp, got)


FDTD_EXPERIMENT = {
    "tau": 100e-15, "dx": 10e-9,
    "time": {"t_cutoff": 0.2, "t_detect": 0.3, "t_max": 0.5, "t_snap": 0.2},
    "source": {"frequency": 120e12, "phase": 0.1, "x": 30e-9, "y": -20e-9, "radius": 50e-9},
    "cavity_rings": [{"radius": 400e-9, "mu_r": 11.56, "eps_r": 1.0, "sigma": 0.0},
                     {"radius": 100e-9, "mu_r": 1.0, "eps_r": 2.0, "sigma": 0.0}],
}


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["", "_lut", "_render"], ids=["coef", "lut", "render"])
def test_fdtd_hip_equals_cpu_backend(tmp_path, variant):
    """examples/fdtd unchanged (coef / LUT / render material resolvers, all three TDV strategy
    names, snapshots => several update calls with iteration_offset): every CSV the MI355X build writes
    equals the cpu-backend build's, text for text."""
    hip_bin, cpu_bin = exe(f"fdtd{variant}_hip"), exe(f"fdtd{variant}_cpu")
    cfg = tmp_path / "experiment.json"
    cfg.write_text(json.dumps(FDTD_EXPERIMENT))
    outs = {}
    for name, binary in (("hip", hip_bin), ("cpu", cpu_bin)):
        d = tmp_path / name
        d.mkdir()
        res = run([binary, "-c", str(cfg), "-o", str(d)])
        assert b"n. timesteps      = 2142" in res.stdout and b"Walltime:" in res.stdout
        outs[name] = {f: open(d / f).read() for f in sorted(os.listdir(d))}
    assert sorted(outs["hip"]) == ["hz.1714.csv", "hz.2571.csv", "hz.857.csv", "hz_sum.2142.csv"]
    assert outs["hip"] == outs["cpu"]
    assert any(v != "0" for v in outs["hip"]["hz_sum.2142.csv"].replace("\n", ",").split(","))


CONVECTION_EXPERIMENT = {"ly": 1.0, "lx": 3.0, "py": 0.5, "px": 1.5, "res": 32, "eta0": 1.0, "DcT": 1.0,
                         "deltaT": 1.0, "Ra": 1e7, "Pra": 1e3, "iterMax": 400, "nt": 3, "nout": 1, "nerr": 100,
                         "epsilon": 1e-4, "dmp": 2}


@pytest.mark.gpu
def test_convection_hip_matches_cpu_backend(tmp_path):
    """examples/convection unchanged (fp64, 11-field cell, two transition functions, host-side
    convergence check between update calls)."""
    hip_bin, cpu_bin = exe("convection_hip"), exe("convection_cpu")
    cfg = tmp_path / "experiment.json"
    cfg.write_text(json.dumps(CONVECTION_EXPERIMENT))
    frames = {}
    for name, binary in (("hip", hip_bin), ("cpu", cpu_bin)):
        d = tmp_path / name
        d.mkdir()
        res = run([binary, str(cfg), str(d)])
        assert b"Total time" in res.stdout
        frames[name] = {f: np.loadtxt(d / f, delimiter=",") for f in sorted(os.listdir(d))}
    assert sorted(frames["hip"]) == sorted(frames["cpu"]) and frames["hip"]
    for f in frames["hip"]:
        a, b = frames["hip"][f], frames["cpu"][f]
        assert a.shape == b.shape
        assert np.allclose(a, b, rtol=1e-10, atol=1e-12), f


CONVECTION_EXPERIMENT_256 = dict(CONVECTION_EXPERIMENT, res=256, iterMax=300, nt=2, nerr=100)


@pytest.mark.gpu
def test_convection_res256_hip_and_device_reduction_match_cpu_backend(tmp_path):
    """res = 256 (767 x 255 cells of 88 bytes), two time steps of up to three blocks of 100 pseudo-transient
    iterations: the unchanged example on the MI355X backend and the driver with the device-side convergence
    check (examples/convection_device_reduce.cpp, stencil::hip::max_abs instead of the host scan of
    convection.cpp:412-438) against the cpu-backend build: same iteration counts per time step, same
    residuals to the printed digits, every CSV within rtol 1e-10."""
    bins = {"cpu": exe("convection_cpu"), "hip": exe("convection_hip"), "reduce": exe("convection_reduce_hip")}
    cfg = tmp_path / "experiment.json"
    cfg.write_text(json.dumps(CONVECTION_EXPERIMENT_256))
    frames, steps = {}, {}
    for name, binary in bins.items():
        d = tmp_path / name
        d.mkdir()
        env = dict(os.environ, OMP_NUM_THREADS="16")
        res = subprocess.run([binary, str(cfg), str(d)], check=True, capture_output=True, env=env, timeout=900)
        lines = [l for l in res.stdout.decode().splitlines() if l.startswith("it = ")]
        # "it = 1 (iter = 300, time = 1.2e-01), errV=1.234e-02, errP=5.678e-03"
        steps[name] = [(l.split("(iter = ")[1].split(",")[0], l.split("errV=")[1].split(",")[0],
                        l.split("errP=")[1].strip()) for l in lines]
        frames[name] = {f: np.loadtxt(d / f, delimiter=",") for f in sorted(os.listdir(d))}
    assert len(steps["cpu"]) == 2
    for name in ("hip", "reduce"):
        assert [s[0] for s in steps[name]] == [s[0] for s in steps["cpu"]], "iteration counts differ"
        for got, want in zip(steps[name], steps["cpu"]):
            assert abs(float(got[1]) - float(want[1])) <= 2e-3 * abs(float(want[1]))
            assert abs(float(got[2]) - float(want[2])) <= 2e-3 * abs(float(want[2]))
        assert sorted(frames[name]) == sorted(frames["cpu"]) and frames[name]
        for f in frames[name]:
            assert np.allclose(frames[name][f], frames["cpu"][f], rtol=1e-10, atol=1e-12), (name, f)
    # the two GPU drivers run the same kernels on the same data: identical files
    for f in frames["hip"]:
        assert np.array_equal(frames["hip"][f], frames["reduce"][f])


@pytest.mark.gpu
def test_reduce_max_abs_matches_numpy(gpu):
    """ststhip_reduce_max_abs: per-field index limits, f32 and f64 fields, NaNs skipped, empty ranges."""
    import torch

    from stencilstream_amd import capi

    capi.init(0)
    rng = np.random.default_rng(9)
    cell = np.dtype([("a", "<f8"), ("b", "<f4"), ("pad", "<f4"), ("c", "<f8")])
    H, W = 301, 517
    cells = np.zeros((H, W), dtype=cell)
    for name in ("a", "b", "c"):
        cells[name] = rng.standard_normal((H, W)) * 100
    cells["a"][H - 1, W - 1] = -1e9   # outside the limits below: must not count
    cells["c"][5, 7] = np.nan
    cells["b"][200, 300] = -12345.0
    dev = torch.from_numpy(cells.view(np.uint8).reshape(-1).copy()).to(gpu)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    fields = [(0, "f8", H - 1, W - 1), (8, "f4", H, W), (16, "f8", H, 10), (0, "f8", 0, W), (16, "f8", H, W)]
    got = capi.reduce_max_abs(dev.data_ptr(), cell.itemsize, H, W, fields, stream=s.cuda_stream)
    assert got[0] == np.abs(cells["a"][:H - 1, :W - 1]).max()
    assert got[1] == float(np.abs(cells["b"]).max()) == 12345.0
    assert got[2] == np.nanmax(np.abs(cells["c"][:, :10]))
    assert got[3] == -np.inf
    assert got[4] == np.nanmax(np.abs(cells["c"]))


@pytest.mark.gpu
@pytest.mark.parametrize("res,iterations", [(40, 7), (128, 100), (256, 100)])
def test_convection_kernels_bit_identical_to_cpu_backend(res, iterations):
    """The reference's PseudoTransientKernel (3 sub-iterations) and ThermalSolverKernel (2) on stencil::hip and on
    stencil::cpu in one binary (examples/convection_bits_test.cpp): all 11 fp64 fields of every cell identical as
    bits after two time steps."""
    res = run([exe("convection_bits_test"), str(res), str(iterations)])
    assert b"0 of" in res.stdout and b"differ" in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("res,iterations", [(40, 30), (128, 100), (256, 64)])
def test_convection_kernels_on_hip_equal_the_oracle(oracle, tmp_path, res, iterations):
    """The reference's unchanged convection kernels on stencil::hip (build/examples/convection_dump_hip: per-field
    planes, the generic staged rule: two generations per launch on two stages) against the oracle's restatement of
    them: two time steps of `iterations` pseudo-transient iterations + one thermal step, all eleven fp64 fields of
    every cell as bits."""
    exe = os.path.join(EX, "convection_dump_hip")
    if not os.path.exists(exe):
        pytest.fail("build/examples/convection_dump_hip missing: run __graft_entry__.build() where /root/reference exists")
    subprocess.check_call([exe, str(res), str(iterations), str(tmp_path)], stdout=subprocess.DEVNULL)
    differing, cells, moved = oracle.convection_dump_check(str(tmp_path), iterations)
    assert cells == (3 * res) * res and moved > 1e-6
    assert differing == {"round0_pt": 0, "round0_ts": 0, "round1_pt": 0, "round1_ts": 0}, differing
