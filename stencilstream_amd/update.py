"""Python mirror of stencil::hip::Grid / stencil::hip::StencilUpdate for the precompiled
transition functions (reference interface: StencilStream/cuda/Grid.hpp:50-188,
StencilStream/cuda/StencilUpdate.hpp:41-198).  Device memory and streams are torch's; every sweep
runs in libststhip.so through the C ABI.
"""
import ctypes as C
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import capi


@dataclass
class TransitionFunction:
    """A precompiled transition function: registry name + its parameter block."""

    app: str
    params: C.Structure = field(default_factory=capi.NoParams)
    cell_dtype: np.dtype = None  # numpy dtype of one AoS cell

    def info(self):
        return capi.app_info(self.app)


def jacobi(variant="Jacobi5General", coef=()):
    p = capi.JacobiParams()
    for i, c in enumerate(coef):
        p.coef[i] = float(np.float32(c))
    return TransitionFunction(variant.lower(), p, np.dtype("<f4"))


HOTSPOT_CELL = np.dtype([("temp", "<f4"), ("power", "<f4")])
SELFCHECK_CELL = np.dtype(
    [("r", "<i4"), ("c", "<i4"), ("i_iteration", "<i4"), ("i_subiteration", "<i4"), ("status", "<i4")]
)
FDTD_CELL = np.dtype([(n, "<f4") for n in ("ex", "ey", "hz", "hz_sum", "ca", "cb", "da", "db")])


def hotspot(Rx_1, Ry_1, Rz_1, Cap_1, split_cell_structure=True):
    p = capi.HotspotParams(Rx_1, Ry_1, Rz_1, Cap_1)
    return TransitionFunction("hotspot" if split_cell_structure else "hotspot_aos", p, HOTSPOT_CELL)


HOTSPOT_CELL_F64 = np.dtype([("temp", "<f8"), ("power", "<f8")])


def hotspot_f64(Rx_1, Ry_1, Rz_1, Cap_1, split_cell_structure=True):
    """HotSpot with the reference's formula evaluated in fp64 (the reference itself is fp32)."""
    p = capi.HotspotParamsF64(Rx_1, Ry_1, Rz_1, Cap_1)
    return TransitionFunction("hotspot_f64" if split_cell_structure else "hotspot_f64_aos", p, HOTSPOT_CELL_F64)


def conway():
    return TransitionFunction("conway", capi.NoParams(), np.dtype("u1"))


def selfcheck(radius=1, split_cell_structure=False):
    name = f"selfcheck{radius}" + ("_soa" if split_cell_structure else "")
    return TransitionFunction(name, capi.NoParams(), SELFCHECK_CELL)


def fdtd(params, split_cell_structure=True, layout=None):
    """layout: "grouped" (two planes of 16 bytes: fields / material coefficients; the fastest, default for a split
    request), "planes" (one plane per field, the reference's SoA protocol) or "aos"."""
    if layout is None:
        layout = "grouped" if split_cell_structure else "aos"
    app = {"grouped": "fdtd_coef_grouped", "planes": "fdtd_coef", "aos": "fdtd_coef_aos"}[layout]
    return TransitionFunction(app, params, FDTD_CELL)


class Grid:
    """H x W row-major AoS cells in HBM (a torch uint8 tensor of H*W*cell_size bytes)."""

    dimensions = 2

    def __init__(self, height, width, cell_dtype, device=None, _cells=None):
        self.cell_dtype = np.dtype(cell_dtype)
        self.height, self.width = int(height), int(width)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        n = self.height * self.width * self.cell_dtype.itemsize
        self.cells = _cells if _cells is not None else torch.zeros(max(n, 1), dtype=torch.uint8, device=self.device)

    @classmethod
    def from_numpy(cls, array, device=None):
        array = np.ascontiguousarray(array)
        assert array.ndim == 2
        g = cls(array.shape[0], array.shape[1], array.dtype, device)
        g.copy_from_buffer(array)
        return g

    def copy_from_buffer(self, array):
        array = np.ascontiguousarray(array)
        if array.shape != (self.height, self.width):
            raise ValueError("The target buffer has not the same size as the grid")
        if self.cell_dtype == np.dtype("u1") and array.dtype != np.bool_ and array.size and int(array.max()) > 1:
            # one-byte cells are the C++ `bool` cells of the Game of Life (examples/conway/conway.cpp:35):
            # bytes 0 / 1 only.  The kernel on words of four cells relies on it (ststhip.h, ststhip_app_run)
            raise ValueError("one-byte cells are C++ bools: every byte must be 0 or 1")
        raw = torch.from_numpy(array.view(np.uint8).reshape(-1).copy())
        if raw.numel():
            self.cells[: raw.numel()].copy_(raw)

    def copy_to_buffer(self, array):
        if array.shape != (self.height, self.width):
            raise ValueError("The target buffer has not the same size as the grid")
        array[...] = self.to_numpy()

    def to_numpy(self):
        n = self.height * self.width * self.cell_dtype.itemsize
        host = self.cells[:n].cpu().numpy()
        return host.view(self.cell_dtype).reshape(self.height, self.width).copy()

    def get_grid_height(self):
        return self.height

    def get_grid_width(self):
        return self.width

    def get_grid_range(self):
        return (self.height, self.width)

    def make_similar(self):
        return Grid(self.height, self.width, self.cell_dtype, self.device)

    def domain(self):
        return capi.Domain(self.height, self.width, 0, self.height, self.width)


@dataclass
class Params:
    """Same fields and order as cuda::StencilUpdate::Params (cuda/StencilUpdate.hpp:54-105)."""

    transition_function: TransitionFunction
    halo_value: object = None  # one cell (numpy scalar / tuple); default = zero-initialised Cell()
    iteration_offset: int = 0
    n_iterations: int = 1
    device: object = None
    blocking: bool = False
    profiling: bool = False


class StencilUpdate:
    def __init__(self, params):
        self.params = params
        self.n_processed_cells = 0
        self.walltime = 0.0
        self.kernel_runtime = 0.0
        self.info = params.transition_function.info()

    def get_params(self):
        return self.params

    def get_n_processed_cells(self):
        return self.n_processed_cells

    def get_walltime(self):
        return self.walltime

    def get_kernel_runtime(self):
        return self.kernel_runtime

    def _halo_bytes(self):
        dt = self.params.transition_function.cell_dtype
        cell = np.zeros((), dtype=dt)
        if self.params.halo_value is not None:
            cell[...] = self.params.halo_value
        return cell.tobytes()

    def __call__(self, source):
        p = self.params
        tf = p.transition_function
        info = self.info
        if source.cell_dtype.itemsize != info.cell_size:
            raise ValueError("grid cell type does not match the transition function")
        if p.n_iterations == 0:
            return source  # handle onto the same cells, as the reference's AoS path
        capi.init(source.device.index if source.device.index is not None else -1)
        started = time.perf_counter()
        with capi.on_stream(source.device) as torch_stream:
            result, run = self._run(source, torch_stream)
        self.walltime += time.perf_counter() - started
        self.kernel_runtime += run.kernel_time_s
        self.n_processed_cells += p.n_iterations * source.height * source.width
        return result

    def _run(self, source, torch_stream):
        p = self.params
        tf = p.transition_function
        info = self.info
        stream = torch_stream.cuda_stream

        result = source.make_similar()
        dom = source.domain()
        n_cells = source.height * source.width
        if info.n_planes == 1:
            run = capi.app_run(tf.app, tf.params, self._halo_bytes(), dom, [source.cells.data_ptr()],
                               [result.cells.data_ptr()], p.iteration_offset, p.n_iterations,
                               blocking=p.blocking, profiling=p.profiling, stream=stream)
        else:
            n = info.n_planes
            sizes = [info.plane_elem_size[i] for i in range(n)]
            offsets = [info.field_offset[i] for i in range(n)]
            a = [torch.empty(max(n_cells * s, 1), dtype=torch.uint8, device=source.device) for s in sizes]
            b = [torch.empty(max(n_cells * s, 1), dtype=torch.uint8, device=source.device) for s in sizes]
            capi.scatter_fields(source.cells.data_ptr(), info.cell_size, n_cells, offsets, sizes,
                                [t.data_ptr() for t in a], stream)
            run = capi.app_run(tf.app, tf.params, self._halo_bytes(), dom, [t.data_ptr() for t in a],
                               [t.data_ptr() for t in b], p.iteration_offset, p.n_iterations,
                               blocking=False, profiling=p.profiling, stream=stream)
            capi.gather_fields(result.cells.data_ptr(), info.cell_size, n_cells, offsets, sizes,
                               [t.data_ptr() for t in b], stream)
            for t in a + b:
                t.record_stream(torch_stream)
            if p.blocking:
                torch_stream.synchronize()
        result.cells.record_stream(torch_stream)
        return result, run
