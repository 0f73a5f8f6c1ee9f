"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  See oracle/stencil_oracle.h for what is restated and how it is
pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

HOTSPOT_CELL = np.dtype([("temp", "<f4"), ("power", "<f4")])
SELFCHECK_CELL = np.dtype(
    [("r", "<i4"), ("c", "<i4"), ("i_iteration", "<i4"), ("i_subiteration", "<i4"), ("status", "<i4")]
)
FDTD_CELL = np.dtype([(n, "<f4") for n in ("ex", "ey", "hz", "hz_sum", "ca", "cb", "da", "db")])

JACOBI_VARIANTS = {
    "Jacobi1General": 0,
    "Jacobi2Constant": 1,
    "Jacobi3Constant": 2,
    "Jacobi4Constant": 3,
    "Jacobi5Constant": 4,
    "Jacobi4General": 5,
    "Jacobi5General": 6,
    "Jacobi9General": 7,
}


class HotspotParams(C.Structure):
    _fields_ = [("Rx_1", C.c_float), ("Ry_1", C.c_float), ("Rz_1", C.c_float), ("Cap_1", C.c_float)]


HOTSPOT_CELL_F64 = np.dtype([("temp", "<f8"), ("power", "<f8")])


class HotspotParamsF64(C.Structure):
    _fields_ = [("Rx_1", C.c_double), ("Ry_1", C.c_double), ("Rz_1", C.c_double), ("Cap_1", C.c_double)]


class FdtdParams(C.Structure):
    _fields_ = [
        ("dt", C.c_float),
        ("t_0", C.c_float),
        ("tau", C.c_float),
        ("omega", C.c_float),
        ("cutoff_iteration", C.c_uint64),
        ("detect_iteration", C.c_uint64),
        ("source_radius_squared", C.c_float),
        ("source_r", C.c_float),
        ("source_c", C.c_float),
        ("source_distance_bound", C.c_float),
        ("double_center_rc", C.c_float),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.oracle_fdtd_tdv.restype = C.c_float
    return _LIB


class use_fma_build:
    """Context manager: route calls to the FMA-contracted build of the same source (x86 FMA
    hardware needed).  Exists only to reproduce known answers recorded from such a build."""

    def __enter__(self):
        global _LIB
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_fma.so"])
        self.saved = lib()
        _LIB = C.CDLL(os.path.join(_HERE, "liboracle_fma.so"))
        _LIB.oracle_fdtd_tdv.restype = C.c_float
        return self

    def __exit__(self, *exc):
        global _LIB
        _LIB = self.saved


class window:
    """Context manager: the arrays handed to the oracle are the window at (row0, col0) of a grid_h x grid_w
    grid; transition functions see global coordinates and the global grid range (stencil_oracle.c, window
    mode).  Results are exact further than n * n_subiterations * radius cells from window borders that are
    not borders of the grid."""

    def __init__(self, row0, col0, grid_h, grid_w):
        self.args = (row0, col0, grid_h, grid_w)

    def __enter__(self):
        lib().oracle_set_window(*[_sz(v) for v in self.args])
        return self

    def __exit__(self, *exc):
        lib().oracle_set_window(_sz(0), _sz(0), _sz(0), _sz(0))


def cpu_has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _sz(x):
    return C.c_size_t(int(x))


def jacobi(variant, coef, grid, n_iterations, halo=0.0, iteration_offset=0, n_threads=1):
    v = JACOBI_VARIANTS[variant] if isinstance(variant, str) else int(variant)
    src = np.ascontiguousarray(grid, dtype=np.float32)
    out = np.empty_like(src)
    cf = np.zeros(9, dtype=np.float32)
    cf[: len(coef)] = np.asarray(coef, dtype=np.float32)
    rc = lib().oracle_jacobi(
        C.c_int(v), _p(cf), _p(src), _p(out), _sz(src.shape[0]), _sz(src.shape[1]),
        C.c_float(halo), _sz(iteration_offset), _sz(n_iterations), C.c_int(n_threads),
    )
    assert rc == 0
    return out


def jacobi25(coef, grid, n_iterations, halo=0.0, n_threads=1):
    """Dense 5 x 5 Jacobi of radius 2 (an extra: SURVEY 8(f)4), coef[(dr+2)*5 + (dc+2)]."""
    src = np.ascontiguousarray(grid, dtype=np.float32)
    out = np.empty_like(src)
    cf = np.ascontiguousarray(coef, dtype=np.float32).reshape(25)
    rc = lib().oracle_jacobi25(_p(cf), _p(src), _p(out), _sz(src.shape[0]), _sz(src.shape[1]), C.c_float(halo),
                               _sz(n_iterations), C.c_int(n_threads))
    assert rc == 0
    return out


def jacobi_init(H, W):
    g = np.empty((H, W), dtype=np.float32)
    lib().oracle_jacobi_init(_p(g), _sz(H), _sz(W))
    return g


def hotspot_params(n_rows, n_columns):
    p = HotspotParams()
    lib().oracle_hotspot_params_for_grid(_sz(n_rows), _sz(n_columns), C.byref(p))
    return p


def hotspot(params, cells, n_iterations, iteration_offset=0, n_threads=1):
    src = np.ascontiguousarray(cells, dtype=HOTSPOT_CELL)
    out = np.empty_like(src)
    rc = lib().oracle_hotspot(
        C.byref(params), _p(src), _p(out), _sz(src.shape[0]), _sz(src.shape[1]),
        _sz(iteration_offset), _sz(n_iterations), C.c_int(n_threads),
    )
    assert rc == 0
    return out


def hotspot_f64(params, cells, n_iterations, iteration_offset=0, n_threads=1):
    src = np.ascontiguousarray(cells, dtype=HOTSPOT_CELL_F64)
    out = np.empty_like(src)
    rc = lib().oracle_hotspot_f64(
        C.byref(params), _p(src), _p(out), _sz(src.shape[0]), _sz(src.shape[1]),
        _sz(iteration_offset), _sz(n_iterations), C.c_int(n_threads),
    )
    assert rc == 0
    return out


def conway(grid, n_iterations, n_threads=1):
    src = np.ascontiguousarray(grid, dtype=np.uint8)
    out = np.empty_like(src)
    rc = lib().oracle_conway(
        _p(src), _p(out), _sz(src.shape[0]), _sz(src.shape[1]), _sz(n_iterations), C.c_int(n_threads)
    )
    assert rc == 0
    return out


def selfcheck(radius, cells, iteration_offset, n_iterations, n_threads=1):
    src = np.ascontiguousarray(cells, dtype=SELFCHECK_CELL)
    out = np.empty_like(src)
    rc = lib().oracle_selfcheck(
        _sz(radius), _p(src), _p(out), _sz(src.shape[0]), _sz(src.shape[1]),
        _sz(iteration_offset), _sz(n_iterations), C.c_int(n_threads),
    )
    assert rc == 0
    return out


def selfcheck_input(H, W, iteration_offset):
    """tests/StencilUpdateTest.hpp:37-45"""
    g = np.zeros((H, W), dtype=SELFCHECK_CELL)
    g["r"] = np.arange(H, dtype=np.int32)[:, None]
    g["c"] = np.arange(W, dtype=np.int32)[None, :]
    g["i_iteration"] = iteration_offset
    return g


def fdtd_tdv(params, iteration):
    return float(lib().oracle_fdtd_tdv(C.byref(params), _sz(iteration)))


def fdtd(params, cells, n_iterations, iteration_offset=0, n_threads=1):
    src = np.ascontiguousarray(cells, dtype=FDTD_CELL)
    out = np.empty_like(src)
    rc = lib().oracle_fdtd(
        C.byref(params), _p(src), _p(out), _sz(src.shape[0]), _sz(src.shape[1]),
        _sz(iteration_offset), _sz(n_iterations), C.c_int(n_threads),
    )
    assert rc == 0
    return out


# ---- convection (examples/convection/convection.cpp:36-242): eleven fp64 fields
CONVECTION_CELL = np.dtype([(n, "<f8") for n in ("T", "Pt", "Vx", "Vy", "tau_xx", "tau_yy", "sigma_xy", "dVxd_tau",
                                                  "dVyd_tau", "ErrV", "ErrP")])


class PseudoTransientParams(C.Structure):
    _fields_ = [("nx", C.c_size_t), ("ny", C.c_size_t)] + [(n, C.c_double) for n in (
        "roh0_g_alpha", "delta_eta_delta_T", "eta0", "deltaT", "dx", "dy", "delta_tau_iter", "beta", "rho", "dampX",
        "dampY", "DcT")]


class ThermalSolverParams(C.Structure):
    _fields_ = [("nx", C.c_size_t), ("ny", C.c_size_t)] + [(n, C.c_double) for n in ("dx", "dy", "dt", "DcT")]


def pseudo_transient(params, cells, n_iterations, iteration_offset=0, n_threads=1):
    cells = np.ascontiguousarray(cells, dtype=CONVECTION_CELL)
    out = np.empty_like(cells)
    H, W = cells.shape
    rc = lib().oracle_pseudo_transient(C.byref(params), _p(cells), _p(out), _sz(H), _sz(W), _sz(iteration_offset),
                                       _sz(n_iterations), int(n_threads))
    assert rc == 0
    return out


def thermal_solver(params, cells, n_iterations=1, iteration_offset=0, n_threads=1):
    cells = np.ascontiguousarray(cells, dtype=CONVECTION_CELL)
    out = np.empty_like(cells)
    H, W = cells.shape
    rc = lib().oracle_thermal_solver(C.byref(params), _p(cells), _p(out), _sz(H), _sz(W), _sz(iteration_offset),
                                     _sz(n_iterations), int(n_threads))
    assert rc == 0
    return out


def convection_dump_check(directory, n_iterations, n_threads=8):
    """The files a build/examples/convection_dump_* binary wrote into `directory` against this oracle: returns the
    number of cells that differ (all eleven fields compared as bits) per phase, and the number of cells compared."""
    raw = open(f"{directory}/params.bin", "rb").read()
    nx, ny = np.frombuffer(raw[:16], dtype="<u8")
    v = np.frombuffer(raw[16:], dtype="<f8")
    pt = PseudoTransientParams(int(nx), int(ny), *[float(x) for x in v[:12]])
    ts = ThermalSolverParams(int(nx), int(ny), *[float(x) for x in v[14:18]])
    shape = (int(nx) + 1, int(ny) + 1)
    grid = np.fromfile(f"{directory}/init.bin", dtype=CONVECTION_CELL).reshape(shape)
    differing = {}
    for r in range(2):
        grid = pseudo_transient(pt, grid, n_iterations, n_threads=n_threads)
        got = np.fromfile(f"{directory}/round{r}_pt.bin", dtype=CONVECTION_CELL).reshape(shape)
        differing[f"round{r}_pt"] = int((got.view("<u8").reshape(shape + (11,)) != grid.view("<u8").reshape(shape + (11,))).any(axis=-1).sum())
        grid = thermal_solver(ts, grid, 1, n_threads=n_threads)
        got = np.fromfile(f"{directory}/round{r}_ts.bin", dtype=CONVECTION_CELL).reshape(shape)
        differing[f"round{r}_ts"] = int((got.view("<u8").reshape(shape + (11,)) != grid.view("<u8").reshape(shape + (11,))).any(axis=-1).sum())
    moved = float(np.abs(grid["Vx"]).max())
    return differing, shape[0] * shape[1], moved
