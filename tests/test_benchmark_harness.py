"""Host-side logic of tools/benchmark.py, the counterpart of the reference's Julia benchmark harness
(scripts/benchmark-common.jl, examples/*/scripts/benchmark.jl).  No GPU, no example binaries."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import benchmark as B  # noqa: E402


def test_max_grid_follows_the_reference_rule():
    # benchmark-common.jl:196-209: min(memory / 3 / cell size, 2^31) cells, square root, clipped to a power of sqrt(2)
    for cell_size in (4, 8, 32, 88):
        wh = B.max_grid_wh(cell_size, clip_to_base=math.sqrt(2.0))
        cells = min(B.HBM_CAPACITY / 3 / cell_size, 2 ** 31)
        assert wh * wh <= cells < 2 * wh * wh * 1.001
        exponent = math.log(wh + 1, math.sqrt(2.0))
        assert abs(exponent - round(exponent)) < 1e-3  # a power of sqrt(2), floored to an integer
    assert B.max_grid_wh(4) == int(math.sqrt(2 ** 31))  # 32-bit indexable cap, as in the reference


def test_model_is_memory_bound_for_large_and_launch_bound_for_small_grids():
    a = B.APPS["jacobi"]
    big = B.model_runtime("jacobi", 16384, 800)
    assert math.isclose(big, 100 * 16384 ** 2 * 2 * a["cell_size"] / B.HBM_BYTES_PER_S)
    small = B.model_runtime("jacobi", 64, 800)
    assert math.isclose(small, 100 * B.LAUNCH_LATENCY)
    # iteration counts are whole launches and hit the target runtime by the model
    n = B.iterations_for("hotspot", 8192, 0.5)
    assert n % B.APPS["hotspot"]["generations_per_launch"] == 0
    assert 0.5 <= B.model_runtime("hotspot", 8192, n) < 0.5 + B.model_runtime("hotspot", 8192, 8)


def test_hotspot_inputs_follow_input_gen(tmp_path):
    # examples/hotspot/data/input_gen.jl:3-15: temp 30 everywhere, power 0.5 for 1-based rows/cols n/4 .. 3n/4
    n = 64
    temp, power = B.hotspot_inputs(n, str(tmp_path))
    t = np.fromfile(temp, dtype=np.float32).reshape(n, n)
    p = np.fromfile(power, dtype=np.float32).reshape(n, n)
    assert (t == 30.0).all()
    inside = np.zeros((n, n), bool)
    for r in range(1, n + 1):
        for c in range(1, n + 1):
            inside[r - 1, c - 1] = n // 4 <= r <= 3 * n // 4 and n // 4 <= c <= 3 * n // 4
    assert np.array_equal(p == 0.5, inside) and set(np.unique(p)) == {0.0, 0.5}


def test_fdtd_experiment_scales_grid_and_steps(tmp_path):
    # examples/fdtd/scripts/benchmark.jl:138-143 and the example's own derivation (Parameters.hpp:224-251)
    path = B.fdtd_experiment(1024, 600, str(tmp_path))
    e = json.load(open(path))
    grid_wh = int(round(2.0 * e["cavity_rings"][0]["radius"] / e["dx"])) + 2
    dt = e["dx"] / (B.C0 * math.sqrt(2.0)) * 0.99
    n_iters = math.ceil(e["time"]["t_max"] * e["tau"] / dt - 1e-9)
    assert grid_wh == 1024 and n_iters == 600
    assert "t_snap" not in e["time"]


def test_walltime_pattern_matches_the_examples_output():
    assert B.WALLTIME.search("Walltime: 0.111284 s\n").group(1) == "0.111284"
    assert B.WALLTIME.search("foo\nWalltime: 11.2282 s\nGFlops: 3").group(1) == "11.2282"
    assert B.WALLTIME.search("Walltime: fast") is None
