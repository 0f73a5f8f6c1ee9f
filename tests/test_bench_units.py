"""bench.py's bookkeeping without a GPU: the physical roofline fractions are recomputable from the committed counter
file, the counters of a leg are refused when the launch plan differs, the clock sampler parses the driver's files, and
the committed bench line agrees with the committed counters (what a reviewer would recompute by hand)."""
import importlib.util
import json
import os

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def test_physical_fractions_arithmetic(bench):
    # 1 TB in 0.25 s = 4 TB/s = half the peak; 1024 SIMDs x 0.25 s / 1.09 ns wave-instructions = the measured issue peak
    f = bench.physical_fractions(1e12, 0.25 / 1.09e-9 * 1024 * 0.5, 0.25)
    assert abs(f["physical_hbm"] - 0.5) < 1e-12
    assert abs(f["valu_issue"]["measured_1.09ns"] - 0.5) < 1e-9
    assert abs(f["valu_issue"]["guide_2_cycles_at_2.4GHz"] - 0.5 * (2.0 / 2.4) / 1.09) < 1e-9
    assert f["bound"] == "hbm"  # 0.5 of the VALU peak against 4 / 6.29 of the copy rate
    assert bench.physical_fractions(1e11, 0.25 / 1.09e-9 * 1024 * 0.9, 0.25)["bound"] == "valu"
    assert bench.physical_fractions(1e12, None, 0.25)["valu_issue"] is None


def test_committed_counters_describe_every_kernel_leg(bench):
    data = json.load(open(bench.COUNTER_FILE))
    assert data["problems"] == []
    cal = data["calibration"]
    for width in ("4B", "8B", "16B"):  # FETCH_SIZE x 2, WRITE_SIZE x 1 at every access width of the sweeps
        assert abs(cal["read"][f"{width}_plain"] - 2.0) < 0.01 and abs(cal["write"][f"{width}_plain"] - 1.0) < 0.01
    for leg in ("headline", "headline_full_grid", "general_coefficients", "general_coefficients_fma", "hotspot_8192",
                "fdtd_max_grid"):
        e = data["legs"][leg]
        assert e["hbm_bytes_per_call"] > 0 and e["valu_per_call"] > 0 and e["launches_per_call"] > 0
        assert abs(e["hbm_bytes_per_call"] - e["hbm_read_bytes_per_call"] - e["hbm_write_bytes_per_call"]) < 1.0
        assert 0.99 < sum(e["sq"][k] for k in ("frac_parked_at_waitcnt_or_barrier", "frac_issue_stalled", "frac_issuing")) < 1.01
        assert all(int(sh["scratch_bytes"] or 0) == 0 for sh in e["shapes"])
    # a headline launch moves at least the compulsory bytes (every cell read once, written once) and not much more
    full = data["legs"]["headline_full_grid"]
    per_launch = full["hbm_bytes_per_call"] / full["launches_per_call"]
    assert 1.0 <= per_launch / (2 * 16384 * 16384 * 4) < 1.35


def test_counters_are_refused_for_another_launch_plan(bench):
    profiled = json.load(open(bench.COUNTER_FILE))["legs"]["headline"]["launches_per_call"]
    assert profiled == 126  # two strips with a moving boundary: two launches per pass of 16 generations
    leg, why = bench.leg_counters("headline", profiled)
    assert leg is not None and why is None
    leg, why = bench.leg_counters("headline", profiled + 1)
    assert leg is None and "launch plan" in why
    leg, why = bench.leg_counters("no_such_leg", None)
    assert leg is None and "no leg" in why


def test_committed_bench_line_follows_from_the_committed_counters(bench):
    line = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    data = json.load(open(bench.COUNTER_FILE))
    assert line["verified"] is True and line["n_gpus"] == 1 and line["unit"] == "Gcell-updates/s"
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    timed = roof["timed_path"]
    step = data["legs"]["headline"]
    want = bench.physical_fractions(step["hbm_bytes_per_call"], step["valu_per_call"], line["ms_per_step"] * 1e-3)
    assert abs(timed["physical_hbm"] - want["physical_hbm"]) < 1e-9
    assert abs(timed["valu_issue"]["measured_1.09ns"] - want["valu_issue"]["measured_1.09ns"]) < 1e-9
    assert roof["bound"] == want["bound"]
    full = data["legs"]["headline_full_grid"]
    assert abs(roof["traffic"] - full["hbm_bytes_per_call"] / full["launches_per_call"]) < 1.0
    for name in ("general_coefficients", "general_coefficients_fma", "hotspot_8192", "fdtd_max_grid"):
        leg = line["legs"][name]
        r = leg["roofline"]
        w = bench.physical_fractions(data["legs"][name]["hbm_bytes_per_call"], data["legs"][name]["valu_per_call"], leg["s"])
        assert abs(r["physical_hbm"] - w["physical_hbm"]) < 1e-9 and r["bound"] == w["bound"], name
    assert line["legs"]["random_init"]["verified"] is True


def test_clock_sampler_parses_the_drivers_files(bench, tmp_path):
    sampler = bench.ClockSampler.__new__(bench.ClockSampler)
    f = tmp_path / "freq1_input"
    f.write_text("2405000000\n")
    sampler.files, sampler.samples = [str(f)], []
    assert sampler._read() == 2405.0
    g = tmp_path / "pp_dpm_sclk"
    g.write_text("0: 132Mhz\n1: 2100Mhz *\n2: 2400Mhz\n")
    sampler.files = [str(g)]
    assert sampler._read() == 2100.0
    sampler.samples = [2100.0, 2300.0]
    rep = sampler.report()
    assert rep["mean_MHz"] == 2200.0 and rep["min_MHz"] == 2100.0 and rep["samples"] == 2
