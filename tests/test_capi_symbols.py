"""CPU-side checks of the C-ABI library: it loads, exports every entry point include/ststhip.h
declares, describes the precompiled transition functions correctly, and refuses to compute
without a GPU (no fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ststhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ststhip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ("ststhip_init", "ststhip_app_sweep", "ststhip_app_run", "ststhip_scatter_fields",
                 "ststhip_gather_fields", "ststhip_comm_exchange_rows", "ststhip_launch"):
        assert must in names
    assert len(names) >= 35


def test_library_exports_every_declared_symbol(built_lib):
    raw = C.CDLL(os.path.join(ROOT, "stencilstream_amd", "libststhip.so"))
    missing = [n for n in declared_symbols() if not hasattr(raw, n)]
    assert not missing, f"libststhip.so lacks {missing}"


def test_abi_version(built_lib):
    assert built_lib.ststhip_abi_version() == 6


def test_options_come_from_the_environment_once(built_lib, monkeypatch):
    """ststhip_options: the tuning knobs are read once, not per launch; ststhip_reload_options re-reads them; the
    Python struct mirrors the C one (the last field of the C struct is where ctypes expects it)."""
    from stencilstream_amd import capi

    capi.reload_options()
    o = capi.options()
    assert (o.narrow_form_kcells, o.n_taper, o.skip_constant_stores, o.host_cache_mib) == (6000, -1, 1, 4096)
    # the knobs of ABI 6, at the end of the struct: upload in row blocks, strips with moving boundaries, sub-strips by rule
    assert (o.stream_upload, o.upload_block_mib, o.skewed_strips, o.strip_substrips) == (1, 0, 1, -1)
    monkeypatch.setenv("STSTHIP_STRIP_SUBSTRIPS", "2")
    monkeypatch.setenv("STSTHIP_UPLOAD_BLOCK_MIB", "5")
    capi.reload_options()
    assert (capi.options().strip_substrips, capi.options().upload_block_mib) == (2, 5)
    monkeypatch.delenv("STSTHIP_STRIP_SUBSTRIPS")
    monkeypatch.delenv("STSTHIP_UPLOAD_BLOCK_MIB")
    capi.reload_options()
    monkeypatch.setenv("STSTHIP_CHUNK_ROWS", "77")
    monkeypatch.setenv("STSTHIP_TAPER", "150:2,50:4")
    monkeypatch.setenv("STSTHIP_EXCHANGE_EVERY", "3")
    assert capi.options().chunk_rows == 0  # not re-read behind the host's back
    capi.reload_options()
    o = capi.options()
    assert o.chunk_rows == 77 and o.n_taper == 2 and list(o.taper_permille)[:2] == [150, 50] and list(o.taper_split)[:2] == [2, 4]
    assert o.exchange_every == 3
    monkeypatch.delenv("STSTHIP_CHUNK_ROWS")
    monkeypatch.delenv("STSTHIP_TAPER")
    monkeypatch.delenv("STSTHIP_EXCHANGE_EVERY")
    capi.reload_options()
    assert capi.options().chunk_rows == 0


def test_registry_describes_the_apps(built_lib):
    from stencilstream_amd import capi, update as U

    apps = capi.list_apps()
    for name in ("jacobi1general", "jacobi2constant", "jacobi3constant", "jacobi4constant",
                 "jacobi5constant", "jacobi4general", "jacobi5general", "jacobi9general", "hotspot",
                 "hotspot_aos", "hotspot_f64", "hotspot_f64_aos", "jacobi5uniform", "jacobi5uniform_first",
                 "jacobi5uniform_last", "jacobi5uniform_only", "jacobi5general_fma", "conway", "conway_packed", "selfcheck1", "selfcheck1_soa", "selfcheck2", "fdtd_coef",
                 "fdtd_coef_aos", "fdtd_coef_grouped", "jacobi5general_independent", "hotspot_independent"):
        assert name in apps
    j = capi.app_info("jacobi5general")
    assert (j.cell_size, j.n_planes, j.stencil_radius, j.n_subiterations, j.tdv_size) == (4, 1, 1, 1, 0)
    assert j.params_size == C.sizeof(capi.JacobiParams) and j.max_generations >= 4
    h = capi.app_info("hotspot")
    assert (h.cell_size, h.n_planes) == (U.HOTSPOT_CELL.itemsize, 2)
    assert list(h.plane_elem_size[:2]) == [4, 4] and list(h.field_offset[:2]) == [0, 4]
    assert h.params_size == C.sizeof(capi.HotspotParams)
    f = capi.app_info("fdtd_coef")
    assert (f.cell_size, f.n_planes, f.n_subiterations, f.tdv_size) == (32, 8, 2, 4)
    assert f.params_size == C.sizeof(capi.FdtdParams) and f.halo_depth_per_generation == 2
    s = capi.app_info("selfcheck1_soa")
    assert (s.cell_size, s.n_planes, s.n_subiterations, s.tdv_size) == (20, 5, 2, 8)
    assert capi.app_info("selfcheck2").stencil_radius == 2
    assert capi.app_info("conway").cell_size == 1


def test_unknown_app_is_an_error(built_lib):
    from stencilstream_amd import capi

    with pytest.raises(capi.StsthipError) as e:
        capi.app_info("no_such_kernel")
    assert e.value.status == 3


def test_no_gpu_means_loud_failure(built_lib):
    """On a box without a GPU the product must fail, not fall back to a CPU path."""
    import torch

    from stencilstream_amd import capi

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.StsthipError) as e:
        capi.init(0)
    assert e.value.status == 4 and "no CPU fallback" in str(e.value)
    dom = capi.Domain(8, 8, 0, 8, 8)
    with pytest.raises(capi.StsthipError):
        capi.app_run("jacobi5general", capi.JacobiParams(), b"\0" * 4, dom, [0x1000], [0x2000], 0, 1)


def test_missing_library_is_an_import_error(monkeypatch):
    from stencilstream_amd import capi

    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libststhip.so")
    with pytest.raises(ImportError):
        capi.load()
