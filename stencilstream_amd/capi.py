"""ctypes binding of libststhip.so (include/ststhip.h).

This is the binding a Python host uses for the precompiled transition functions; it contains no
compute of its own and no CPU fallback: if the HIP library is missing or no GPU is visible, calls
raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libststhip.so")
# (A/B tools only: another build of the same library, e.g. one with experiment translation units linked in)
if os.environ.get("STSTHIP_LIBRARY"):
    LIB_PATH = os.path.abspath(os.environ["STSTHIP_LIBRARY"])

STSTHIP_OK = 0
STATUS_NAMES = {
    0: "STSTHIP_OK",
    1: "STSTHIP_ERR_HIP",
    2: "STSTHIP_ERR_INVALID",
    3: "STSTHIP_ERR_UNKNOWN_APP",
    4: "STSTHIP_ERR_NO_DEVICE",
    5: "STSTHIP_ERR_COMM",
}
COMM_ID_BYTES = 128


class StsthipError(RuntimeError):
    def __init__(self, status, what, message):
        super().__init__(f"{what} failed with {STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class Domain(C.Structure):
    """ststhip_domain"""

    _fields_ = [
        ("global_height", C.c_uint64),
        ("global_width", C.c_uint64),
        ("row_origin", C.c_int64),
        ("local_rows", C.c_uint64),
        ("pitch", C.c_uint64),
        ("col_origin", C.c_int64),   # ABI 5: a block of a 2-D decomposition; local_cols = 0: whole rows
        ("local_cols", C.c_uint64),
    ]


class AppInfo(C.Structure):
    """ststhip_app_info"""

    _fields_ = [
        ("name", C.c_char_p),
        ("cell_size", C.c_uint32),
        ("params_size", C.c_uint32),
        ("stencil_radius", C.c_uint32),
        ("n_subiterations", C.c_uint32),
        ("n_planes", C.c_uint32),
        ("plane_elem_size", C.c_uint32 * 16),
        ("field_offset", C.c_uint32 * 16),
        ("max_generations", C.c_uint32),
        ("tdv_size", C.c_uint32),
        ("halo_depth_per_generation", C.c_uint32),
        ("strip_width", C.c_uint32),
        ("cells_per_lane", C.c_uint32),
        ("prefetch_rows", C.c_uint32),
        ("stages", C.c_uint32),
        ("default_generations", C.c_uint32),
    ]


class RunInfo(C.Structure):
    """ststhip_run_info"""

    _fields_ = [
        ("walltime_s", C.c_double),
        ("kernel_time_s", C.c_double),
        ("n_launches", C.c_uint64),
        ("n_processed_cells", C.c_uint64),
        ("n_streamed_passes", C.c_uint64),
    ]


class SourceBlock(C.Structure):
    """ststhip_source_block: rows up to `row_end` of a source that is arriving are in HBM once `ready` has completed"""

    _fields_ = [("row_end", C.c_uint64), ("ready", C.c_void_p)]


class JacobiParams(C.Structure):
    _fields_ = [("coef", C.c_float * 9)]


class Jacobi25Params(C.Structure):
    """ststhip_jacobi25_params: dense 5 x 5 Jacobi of radius 2"""

    _fields_ = [("coef", C.c_float * 25)]


class JacobiUniformParams(C.Structure):
    """parameter block of the jacobi5uniform* kernels: the common coefficient"""

    _fields_ = [("c", C.c_float)]


class HotspotParams(C.Structure):
    _fields_ = [("Rx_1", C.c_float), ("Ry_1", C.c_float), ("Rz_1", C.c_float), ("Cap_1", C.c_float)]


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
        ("reserved", C.c_uint32),
    ]


class ReduceField(C.Structure):
    """ststhip_reduce_field"""

    _fields_ = [("offset", C.c_uint32), ("type", C.c_uint32), ("row_limit", C.c_uint64), ("col_limit", C.c_uint64)]


class NoParams(C.Structure):
    _fields_ = [("unused", C.c_int)]


class Options(C.Structure):
    """ststhip_options (include/ststhip.h)."""
    _fields_ = [(n, C.c_int32) for n in ("chunk_rows", "tail_permille", "n_taper")] + \
               [("taper_permille", C.c_int32 * 3), ("taper_split", C.c_int32 * 3)] + \
               [(n, C.c_int32) for n in (
                   "narrow_form_kcells", "narrow_band_rows", "skip_constant_stores", "xcd_remap", "last_chunk_early",
                   "max_generations", "allow_spilling_depths", "virtual_strips", "two_strips_permille",
                   "two_strips_permille_outer", "strip_skew_permille", "bands_beside_interior", "band_stream_priority",
                   "bands_apart", "bands_one_launch", "comm_stream_priority", "jacobi_fastpath", "conway_fastpath",
                   "prepare_streams", "host_cache_mib", "tune_depth", "exchange_every", "stream_upload",
                   "upload_block_mib", "skewed_strips", "strip_substrips")] + \
               [("reserved", C.c_int32 * 3)]


_lib = None


def load():
    """Load libststhip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7 / libhsa-runtime64.so.1; importing it first makes the
    # loader resolve this library's HIP dependency to that same copy (one HIP runtime per process,
    # so torch tensors, streams and events are valid here).  A C++ host uses the system runtime.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C stencilstream_amd/csrc`; the MI355X backend has no CPU fallback"
        )
    lib = C.CDLL(LIB_PATH)
    lib.ststhip_last_error.restype = C.c_char_p
    lib.ststhip_set_last_error.argtypes = [C.c_char_p]
    lib.ststhip_set_last_error.restype = None
    vp, sz, u64, u32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32
    pp = C.POINTER(C.c_void_p)
    sigs = {
        "ststhip_abi_version": [],
        "ststhip_init": [C.c_int],
        "ststhip_reload_options": [],
        "ststhip_shutdown": [],
        "ststhip_device_count": [C.POINTER(C.c_int)],
        "ststhip_device_name": [C.c_char_p, sz],
        "ststhip_compute_units": [C.POINTER(C.c_int)],
        "ststhip_malloc": [pp, sz],
        "ststhip_free": [vp],
        "ststhip_malloc_async": [pp, sz, vp],
        "ststhip_free_async": [vp, vp],
        "ststhip_pool_trim": [],
        "ststhip_host_malloc": [pp, sz],
        "ststhip_host_free": [vp],
        "ststhip_memcpy_h2d": [vp, vp, sz, vp],
        "ststhip_memcpy_d2h": [vp, vp, sz, vp],
        "ststhip_memcpy_d2d": [vp, vp, sz, vp],
        "ststhip_memset": [vp, C.c_int, sz, vp],
        "ststhip_default_stream": [pp],
        "ststhip_stream_create": [pp],
        "ststhip_stream_destroy": [vp],
        "ststhip_stream_synchronize": [vp],
        "ststhip_stream_wait_event": [vp, vp],
        "ststhip_event_create": [pp],
        "ststhip_event_destroy": [vp],
        "ststhip_event_record": [vp, vp],
        "ststhip_event_synchronize": [vp],
        "ststhip_event_elapsed_ms": [vp, vp, C.POINTER(C.c_float)],
        "ststhip_launch": [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, pp, sz, vp],
        "ststhip_occupancy": [vp, C.c_uint, sz, C.POINTER(C.c_int)],
        "ststhip_kernel_scratch_bytes": [vp, C.POINTER(sz)],
        "ststhip_launch_concurrency": [],
        "ststhip_target_holds_constants": [],
        "ststhip_set_launch_concurrency": [C.c_int],
        "ststhip_launch_row_hole": [C.POINTER(u64), C.POINTER(u64)],
        "ststhip_set_launch_row_hole": [u64, u64],
        "ststhip_launch_columns": [C.POINTER(u64), C.POINTER(u64)],
        "ststhip_set_launch_columns": [u64, u64],
        "ststhip_suggest_row_strips": [C.c_char_p, u64, u64, u64],
        "ststhip_scatter_fields": [vp, sz, sz, C.c_int, C.POINTER(sz), C.POINTER(sz), pp, vp],
        "ststhip_gather_fields": [vp, sz, sz, C.c_int, C.POINTER(sz), C.POINTER(sz), pp, vp],
        "ststhip_reduce_max_abs": [vp, sz, u64, u64, u64, C.c_int, C.POINTER(ReduceField), C.POINTER(C.c_double), vp],
        "ststhip_app_count": [],
        "ststhip_app_info_at": [C.c_int, C.POINTER(AppInfo)],
        "ststhip_app_find": [C.c_char_p, C.POINTER(AppInfo)],
        "ststhip_app_sweep": [C.c_char_p, vp, vp, C.POINTER(Domain), pp, pp, u64, u64, u64, u32, vp],
        "ststhip_app_run": [C.c_char_p, vp, vp, C.POINTER(Domain), pp, pp, u64, u64, C.c_int, C.c_int,
                            vp, C.POINTER(RunInfo)],
        "ststhip_run_passes": [vp, vp, vp, C.POINTER(Domain), pp, pp, u64, u64, C.c_int, C.c_int, vp,
                               C.POINTER(RunInfo)],
        "ststhip_set_source_arrival": [C.POINTER(SourceBlock), u32],
        "ststhip_suggest_upload_blocks": [u64, u64, C.POINTER(u32)],
        "ststhip_upload_streams": [pp, pp],
        "ststhip_tuned_depth": [u64, u64, u64, C.POINTER(u32)],
        "ststhip_app_tuned_depth": [C.c_char_p, u64, u64, C.POINTER(u32)],
        "ststhip_app_scratch_bytes": [C.c_char_p, u32, C.POINTER(sz)],
        "ststhip_comm_unique_id": [C.c_char_p],
        "ststhip_comm_create": [C.c_char_p, C.c_int, C.c_int, pp],
        "ststhip_comm_destroy": [vp],
        "ststhip_comm_set_neighbours": [vp, C.c_int, C.c_int],
        "ststhip_comm_neighbours": [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "ststhip_comm_set_column_neighbours": [vp, C.c_int, C.c_int],
        "ststhip_comm_set_mesh": [vp, C.c_int, C.c_int],
        "ststhip_comm_exchange_columns": [vp, C.c_int, pp, pp, pp, pp, C.POINTER(sz), vp],
        "ststhip_block_create": [C.c_char_p, vp, vp, u64, u64, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, pp],
        "ststhip_block_create_custom": [vp, vp, vp, u64, u64, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, pp],
        "ststhip_block_geometry": [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)],
        "ststhip_block_upload": [vp, C.c_uint, vp, sz],
        "ststhip_block_download": [vp, C.c_uint, vp, sz],
        "ststhip_comm_exchange_rows": [vp, C.c_int, pp, pp, pp, pp, C.POINTER(sz), sz, vp],
        "ststhip_strip_create": [C.c_char_p, vp, vp, u64, u64, C.c_int, C.c_int, vp, vp, vp, pp],
        "ststhip_strip_create_custom": [vp, vp, vp, u64, u64, C.c_int, C.c_int, vp, vp, vp, pp],
        "ststhip_strip_destroy": [vp],
        "ststhip_strip_rows": [vp, C.POINTER(u64), C.POINTER(u64)],
        "ststhip_strip_plane": [vp, C.c_uint, pp, C.POINTER(sz)],
        "ststhip_strip_stream": [vp, pp],
        "ststhip_strip_synchronize": [vp],
        "ststhip_strip_warm_up": [vp],
        "ststhip_strip_advance": [vp, u64, u64, C.c_int],
        "ststhip_strip_counters": [vp, C.POINTER(u64), C.POINTER(u64)],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared entry point
        fn.argtypes = argtypes
        if name != "ststhip_last_error":
            fn.restype = C.c_int
    lib.ststhip_get_options.argtypes = []
    lib.ststhip_get_options.restype = C.POINTER(Options)
    _lib = lib
    return lib


DECLARED_SYMBOLS = None


def last_error():
    msg = load().ststhip_last_error()
    return msg.decode() if msg else ""


def check(status, what):
    if status != STSTHIP_OK:
        raise StsthipError(status, what, last_error())


def init(device=-1):
    check(load().ststhip_init(int(device)), "ststhip_init")


_env_seen = None


def _sync_options():
    """The library reads its STSTHIP_* knobs once; Python hosts (tests, tuning tools) change os.environ between
    calls, so the binding re-reads them when that part of the environment has changed since its last call."""
    global _env_seen
    now = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("STSTHIP_")))
    if now != _env_seen:
        if _env_seen is not None or now:
            load().ststhip_reload_options()
        _env_seen = now


def reload_options():
    """Re-read the STSTHIP_* tuning knobs from the environment (they are read once, at first use)."""
    check(load().ststhip_reload_options(), "ststhip_reload_options")


def options():
    """The tuning knobs in effect (ststhip_options of include/ststhip.h)."""
    return load().ststhip_get_options().contents


def app_info(name):
    info = AppInfo()
    check(load().ststhip_app_find(name.encode(), C.byref(info)), f"ststhip_app_find({name})")
    return info


def list_apps():
    lib = load()
    out = []
    for i in range(lib.ststhip_app_count()):
        info = AppInfo()
        check(lib.ststhip_app_info_at(i, C.byref(info)), "ststhip_app_info_at")
        out.append(info.name.decode())
    return out


_side_streams = {}


class on_stream:
    """Pick the HIP stream the library should use for work on torch tensors of `device`.

    torch's default stream has handle 0, which this library reads as "use the runtime's own
    stream"; instead a dedicated torch stream is used and ordered after/before the current one."""

    def __init__(self, device):
        import torch

        self.torch = torch
        self.device = device
        self.current = torch.cuda.current_stream(device)
        if self.current.cuda_stream != 0:
            self.stream = self.current
        else:
            key = device.index if device.index is not None else torch.cuda.current_device()
            if key not in _side_streams:
                _side_streams[key] = torch.cuda.Stream(device)
            self.stream = _side_streams[key]

    def __enter__(self):
        if self.stream is not self.current:
            self.stream.wait_stream(self.current)
        return self.stream

    def __exit__(self, *exc):
        if self.stream is not self.current:
            self.current.wait_stream(self.stream)


def _ptr_array(pointers):
    arr = (C.c_void_p * len(pointers))(*[C.c_void_p(int(p)) for p in pointers])
    return arr


def app_sweep(app, tf_params, halo_bytes, dom, src_ptrs, dst_ptrs, out_begin, out_end, iteration,
              n_generations, stream=0):
    """One launch: n_generations generations over global rows [out_begin, out_end)."""
    _sync_options()
    halo = C.create_string_buffer(bytes(halo_bytes), len(halo_bytes))
    src = _ptr_array(src_ptrs)
    dst = _ptr_array(dst_ptrs)
    check(
        load().ststhip_app_sweep(
            app.encode(), C.cast(C.byref(tf_params), C.c_void_p), C.cast(halo, C.c_void_p),
            C.byref(dom), src, dst, int(out_begin), int(out_end), int(iteration), int(n_generations),
            C.c_void_p(int(stream)),
        ),
        f"ststhip_app_sweep({app})",
    )


def app_run(app, tf_params, halo_bytes, dom, src_ptrs, dst_ptrs, iteration_offset, n_iterations,
            blocking=True, profiling=False, stream=0):
    """cuda::StencilUpdate::operator() for a precompiled transition function.

    The work is queued on `stream` (0 = the library's own stream), which is NOT torch's current stream: tensors that
    torch kernels have just filled or cleared must be complete first (torch.cuda.synchronize(), an event, or the
    `on_stream` helper below), or the sweep races with them."""
    _sync_options()
    halo = C.create_string_buffer(bytes(halo_bytes), len(halo_bytes))
    src = _ptr_array(src_ptrs)
    dst = _ptr_array(dst_ptrs)
    info = RunInfo()
    check(
        load().ststhip_app_run(
            app.encode(), C.cast(C.byref(tf_params), C.c_void_p), C.cast(halo, C.c_void_p),
            C.byref(dom), src, dst, int(iteration_offset), int(n_iterations), int(bool(blocking)),
            int(bool(profiling)), C.c_void_p(int(stream)), C.byref(info),
        ),
        f"ststhip_app_run({app})",
    )
    return info


def suggest_upload_blocks(rows, row_bytes):
    """Into how many row blocks a host should split the upload of `rows` rows of `row_bytes` bytes (1 = do not)."""
    n = C.c_uint32()
    check(load().ststhip_suggest_upload_blocks(int(rows), int(row_bytes), C.byref(n)), "ststhip_suggest_upload_blocks")
    return int(n.value)


def upload_stream():
    """The stream the runtime keeps for the copies of uploads in row blocks."""
    copies, work = C.c_void_p(), C.c_void_p()
    check(load().ststhip_upload_streams(C.byref(copies), C.byref(work)), "ststhip_upload_streams")
    return copies.value


def upload_in_blocks(planes, rows, n_blocks=None):
    """Copy `rows` rows of every plane -- `planes` = [(pinned host address, device address, bytes per row), ...] -- to
    the device in row blocks on the runtime's upload stream, and name the blocks for the calling thread's next
    `app_run` (ststhip_set_source_arrival).  Returns the events (`events_destroy` them after that call)."""
    lib = load()
    if n_blocks is None:
        n_blocks = suggest_upload_blocks(rows, sum(row_bytes for _, _, row_bytes in planes))
    up = upload_stream()
    blocks = (SourceBlock * n_blocks)()
    events = []
    for b in range(n_blocks):
        first, end = rows * b // n_blocks, rows * (b + 1) // n_blocks
        for host_ptr, device_ptr, row_bytes in planes:
            check(lib.ststhip_memcpy_h2d(C.c_void_p(device_ptr + first * row_bytes),
                                         C.c_void_p(host_ptr + first * row_bytes), (end - first) * row_bytes,
                                         C.c_void_p(up)), "ststhip_memcpy_h2d")
        ev = C.c_void_p()
        check(lib.ststhip_event_create(C.byref(ev)), "ststhip_event_create")
        check(lib.ststhip_event_record(ev, C.c_void_p(up)), "ststhip_event_record")
        events.append(ev)
        blocks[b].row_end = end
        blocks[b].ready = ev.value
    check(lib.ststhip_set_source_arrival(blocks, n_blocks), "ststhip_set_source_arrival")
    return events


def events_destroy(events):
    for ev in events:
        check(load().ststhip_event_destroy(ev), "ststhip_event_destroy")


def app_scratch_bytes(app, n_generations):
    """Largest scratch per work-item over the kernels a launch of `app` at this depth may start."""
    b = C.c_size_t()
    check(load().ststhip_app_scratch_bytes(app.encode(), int(n_generations), C.byref(b)), "ststhip_app_scratch_bytes")
    return int(b.value)


def app_tuned_depth(app, height, width):
    """Blocking depth the pass driver has measured to be the faster one for this app and grid shape (0: not measured)."""
    depth = C.c_uint32()
    check(load().ststhip_app_tuned_depth(app.encode(), int(height), int(width), C.byref(depth)), "ststhip_app_tuned_depth")
    return int(depth.value)


def set_launch_concurrency(n):
    """Number of row-range sweeps the caller keeps in flight side by side (chunk sizing hint)."""
    check(load().ststhip_set_launch_concurrency(int(n)), "ststhip_set_launch_concurrency")


def suggest_row_strips(app, rows, width, n_passes):
    """1 or 2: row strips (on separate streams) the pass driver would use for such a grid."""
    return int(load().ststhip_suggest_row_strips(app.encode(), int(rows), int(width), int(n_passes)))


def reduce_max_abs(cells_ptr, cell_size, height, width, fields, pitch=None, stream=0):
    """max |field| over row < row_limit, column < col_limit for every (offset, 'f4'|'f8', row_limit, col_limit)
    of `fields`, in one pass on the device; -inf where no cell counts."""
    n = len(fields)
    table = (ReduceField * n)(*[ReduceField(int(o), 1 if str(t) in ("f8", "<f8", "float64") else 0, int(r), int(c))
                                for o, t, r, c in fields])
    result = (C.c_double * n)()
    check(load().ststhip_reduce_max_abs(C.c_void_p(int(cells_ptr)), cell_size, height, width,
                                        width if pitch is None else pitch, n, table, result,
                                        C.c_void_p(int(stream))), "ststhip_reduce_max_abs")
    return list(result)


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_void_p)


def comm_unique_id():
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(load().ststhip_comm_unique_id(buf), "ststhip_comm_unique_id")
    return buf.raw


def comm_create(unique_id, rank, n_ranks):
    comm = C.c_void_p()
    check(load().ststhip_comm_create(C.create_string_buffer(unique_id, COMM_ID_BYTES), rank, n_ranks, C.byref(comm)),
          "ststhip_comm_create")
    return comm


def comm_destroy(comm):
    check(load().ststhip_comm_destroy(comm), "ststhip_comm_destroy")


def comm_set_neighbours(comm, up, down):
    """Ranks the ghost rows are exchanged with (-1: none); both may name the same rank, or the rank itself."""
    check(load().ststhip_comm_set_neighbours(comm, int(up), int(down)), "ststhip_comm_set_neighbours")


def comm_neighbours(comm):
    up, down = C.c_int(), C.c_int()
    check(load().ststhip_comm_neighbours(comm, C.byref(up), C.byref(down)), "ststhip_comm_neighbours")
    return up.value, down.value


def comm_exchange_rows(comm, send_up, send_down, recv_up, recv_down, row_bytes, n_rows, stream=0):
    """ststhip_comm_exchange_rows on device pointers (one list entry per plane)."""
    n = len(row_bytes)
    rb = (C.c_size_t * n)(*[int(b) for b in row_bytes])
    check(load().ststhip_comm_exchange_rows(comm, n, _ptr_array(send_up), _ptr_array(send_down), _ptr_array(recv_up),
                                            _ptr_array(recv_down), rb, int(n_rows), C.c_void_p(int(stream) or None)),
          "ststhip_comm_exchange_rows")


class Strip:
    """The native row-strip driver (ststhip_strip_*): one strip of a grid that is cut over several processes / GPUs.

    comm: a communicator from comm_create() joining exactly the strips; or exchange: a Python callable
    (n_planes, send_up, send_down, recv_up, recv_down, row_bytes, n_rows, stream) -> None with the lists of device
    pointers of ststhip_comm_exchange_rows, for hosts whose ranks RCCL cannot join (tests)."""

    def __init__(self, app, tf_params, halo_bytes, total_rows, width, rank, n_ranks, comm=None, exchange=None,
                 exchange_fn_address=None):
        """exchange_fn_address: instead of `exchange`, the address of a C function with the ststhip_exchange_fn
        signature (timing studies that must not pay for a Python callback per exchange)."""
        _sync_options()
        self._callback = None
        cb = None
        if exchange_fn_address is not None:
            cb = C.c_void_p(int(exchange_fn_address))
        elif exchange is not None:
            def trampoline(_ctx, n_planes, su, sd, ru, rd, rb, n_rows, stream):
                try:
                    exchange(n_planes, [su[i] for i in range(n_planes)], [sd[i] for i in range(n_planes)],
                             [ru[i] for i in range(n_planes)], [rd[i] for i in range(n_planes)],
                             [rb[i] for i in range(n_planes)], int(n_rows), stream)
                    return 0
                except Exception as e:  # noqa: BLE001 -- reported through the C ABI's status
                    load().ststhip_set_last_error(str(e).encode())
                    return 5
            self._callback = cb = EXCHANGE_FN(trampoline)
        halo = C.create_string_buffer(bytes(halo_bytes), len(halo_bytes))
        self.handle = C.c_void_p()
        self.tf_params = tf_params
        check(load().ststhip_strip_create(app.encode(), C.cast(C.byref(tf_params), C.c_void_p), C.cast(halo, C.c_void_p),
                                          int(total_rows), int(width), int(rank), int(n_ranks), comm,
                                          C.cast(cb, C.c_void_p) if cb is not None else None, None,
                                          C.byref(self.handle)), f"ststhip_strip_create({app})")
        a, b = C.c_uint64(), C.c_uint64()
        check(load().ststhip_strip_rows(self.handle, C.byref(a), C.byref(b)), "ststhip_strip_rows")
        self.row_begin, self.row_end = int(a.value), int(b.value)
        s = C.c_void_p()
        check(load().ststhip_strip_stream(self.handle, C.byref(s)), "ststhip_strip_stream")
        self.stream = s.value

    def plane(self, index=0):
        """(device pointer of the first owned row of the current buffers, bytes per row)"""
        ptr, row_bytes = C.c_void_p(), C.c_size_t()
        check(load().ststhip_strip_plane(self.handle, index, C.byref(ptr), C.byref(row_bytes)), "ststhip_strip_plane")
        return ptr.value, int(row_bytes.value)

    def upload(self, index, host_array):
        import numpy as np

        rows = np.ascontiguousarray(host_array)
        ptr, row_bytes = self.plane(index)
        assert rows.nbytes == row_bytes * (self.row_end - self.row_begin), "array does not match the strip's plane"
        check(load().ststhip_memcpy_h2d(C.c_void_p(ptr), rows.ctypes.data_as(C.c_void_p), rows.nbytes,
                                        C.c_void_p(self.stream)), "ststhip_memcpy_h2d")
        self.synchronize()

    def upload_from_device(self, index, device_ptr, nbytes):
        ptr, row_bytes = self.plane(index)
        assert nbytes == row_bytes * (self.row_end - self.row_begin), "buffer does not match the strip's plane"
        check(load().ststhip_memcpy_d2d(C.c_void_p(ptr), C.c_void_p(int(device_ptr)), nbytes, C.c_void_p(self.stream)),
              "ststhip_memcpy_d2d")

    def download(self, index, dtype):
        import numpy as np

        ptr, row_bytes = self.plane(index)
        out = np.empty((self.row_end - self.row_begin, row_bytes // np.dtype(dtype).itemsize), dtype=dtype)
        self.synchronize()
        check(load().ststhip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes,
                                        C.c_void_p(self.stream)), "ststhip_memcpy_d2h")
        self.synchronize()
        return out

    def warm_up(self):
        check(load().ststhip_strip_warm_up(self.handle), "ststhip_strip_warm_up")

    def advance(self, iteration_offset, n_generations, blocking=False):
        _sync_options()
        check(load().ststhip_strip_advance(self.handle, int(iteration_offset), int(n_generations), int(bool(blocking))),
              "ststhip_strip_advance")

    def synchronize(self):
        check(load().ststhip_strip_synchronize(self.handle), "ststhip_strip_synchronize")

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(load().ststhip_strip_counters(self.handle, C.byref(a), C.byref(b)), "ststhip_strip_counters")
        return int(a.value), int(b.value)

    def close(self):
        if self.handle:
            load().ststhip_strip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def _exchange_trampoline(exchange):
    """A Python exchange callable behind the ststhip_exchange_fn signature."""
    def trampoline(_ctx, n_planes, su, sd, ru, rd, rb, n_rows, stream):
        try:
            exchange(n_planes, [su[i] for i in range(n_planes)], [sd[i] for i in range(n_planes)],
                     [ru[i] for i in range(n_planes)], [rd[i] for i in range(n_planes)],
                     [rb[i] for i in range(n_planes)], int(n_rows), stream)
            return 0
        except Exception as e:  # noqa: BLE001 -- reported through the C ABI's status
            load().ststhip_set_last_error(str(e).encode())
            return 5
    return EXCHANGE_FN(trampoline)


def comm_set_mesh(comm, mesh_rows, mesh_cols):
    check(load().ststhip_comm_set_mesh(comm, int(mesh_rows), int(mesh_cols)), "ststhip_comm_set_mesh")


def comm_set_column_neighbours(comm, left, right):
    check(load().ststhip_comm_set_column_neighbours(comm, int(left), int(right)), "ststhip_comm_set_column_neighbours")


class Block(Strip):
    """One block of a grid cut into mesh_rows x mesh_cols blocks (ststhip_block_*), one process per block.

    comm: a communicator of the mesh's ranks wired with comm_set_mesh(); or exchange_rows / exchange_cols: Python
    callables with the signature of Strip's `exchange` (the column callable gets the packed ghost columns as one "row"
    per plane: send_up = to the left, send_down = to the right)."""

    def __init__(self, app, tf_params, halo_bytes, total_rows, total_cols, rank, mesh_rows, mesh_cols, comm=None,
                 exchange_rows=None, exchange_cols=None):
        _sync_options()
        self._callback = _exchange_trampoline(exchange_rows) if exchange_rows is not None else None
        self._callback_cols = _exchange_trampoline(exchange_cols) if exchange_cols is not None else None
        halo = C.create_string_buffer(bytes(halo_bytes), len(halo_bytes))
        self.handle = C.c_void_p()
        self.tf_params = tf_params
        as_ptr = lambda cb: C.cast(cb, C.c_void_p) if cb is not None else None  # noqa: E731
        check(load().ststhip_block_create(app.encode(), C.cast(C.byref(tf_params), C.c_void_p), C.cast(halo, C.c_void_p),
                                          int(total_rows), int(total_cols), int(rank), int(mesh_rows), int(mesh_cols), comm,
                                          as_ptr(self._callback), None, as_ptr(self._callback_cols), None,
                                          C.byref(self.handle)), f"ststhip_block_create({app})")
        r0, r1, c0, c1 = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(load().ststhip_block_geometry(self.handle, C.byref(r0), C.byref(r1), C.byref(c0), C.byref(c1)),
              "ststhip_block_geometry")
        self.row_begin, self.row_end, self.col_begin, self.col_end = (int(v.value) for v in (r0, r1, c0, c1))
        s = C.c_void_p()
        check(load().ststhip_strip_stream(self.handle, C.byref(s)), "ststhip_strip_stream")
        self.stream = s.value

    def upload(self, index, host_array):
        import numpy as np

        cells = np.ascontiguousarray(host_array)
        assert cells.shape[0] == self.row_end - self.row_begin, "array does not match the block's rows"
        check(load().ststhip_block_upload(self.handle, index, cells.ctypes.data_as(C.c_void_p), cells.strides[0]),
              "ststhip_block_upload")

    def download(self, index, dtype):
        import numpy as np

        out = np.empty((self.row_end - self.row_begin, self.col_end - self.col_begin), dtype=dtype)
        check(load().ststhip_block_download(self.handle, index, out.ctypes.data_as(C.c_void_p), out.strides[0]),
              "ststhip_block_download")
        return out


def scatter_fields(aos_ptr, cell_size, n_cells, offsets, sizes, plane_ptrs, stream=0):
    n = len(offsets)
    off = (C.c_size_t * n)(*offsets)
    siz = (C.c_size_t * n)(*sizes)
    check(
        load().ststhip_scatter_fields(C.c_void_p(int(aos_ptr)), cell_size, n_cells, n, off, siz,
                                      _ptr_array(plane_ptrs), C.c_void_p(int(stream))),
        "ststhip_scatter_fields",
    )


def gather_fields(aos_ptr, cell_size, n_cells, offsets, sizes, plane_ptrs, stream=0):
    n = len(offsets)
    off = (C.c_size_t * n)(*offsets)
    siz = (C.c_size_t * n)(*sizes)
    check(
        load().ststhip_gather_fields(C.c_void_p(int(aos_ptr)), cell_size, n_cells, n, off, siz,
                                     _ptr_array(plane_ptrs), C.c_void_p(int(stream))),
        "ststhip_gather_fields",
    )
