"""The native row-strip driver (ststhip_strip_*, include/ststhip.h): what a C++ / FFI host calls to advance one
strip of a grid that is cut over several GPUs, one process per GPU.  On one GPU box: a single strip against
ststhip_app_run, and two processes sharing cuda:0 whose ghost rows travel through host memory over gloo (the
exchange callback of ststhip_strip_create; RCCL cannot put two ranks on one device) against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def test_single_strip_equals_app_run(gpu, oracle):
    import torch

    from stencilstream_amd import capi

    capi.init(0)
    rng = np.random.default_rng(21)
    H, W = 1300, 1500
    grid = rng.random((H, W), dtype=np.float32)
    for coef, halo in (([0.2, 0.21, 0.19, 0.22, 0.18], 0.25), ([0.2] * 5, 0.0)):  # general and product-carrying kernels
        p = capi.JacobiParams()
        for i, c in enumerate(coef):
            p.coef[i] = c
        strip = capi.Strip("jacobi5general", p, np.float32(halo).tobytes(), H, W, 0, 1)
        assert (strip.row_begin, strip.row_end) == (0, H)
        strip.upload(0, grid)
        strip.advance(0, 29)
        strip.advance(29, 8, blocking=True)
        got = strip.download(0, np.float32)
        want = oracle.jacobi("Jacobi5General", coef, grid, 37, halo=halo, n_threads=8)
        assert np.array_equal(bits(got), bits(want))
        launches, exchanges = strip.counters()
        assert launches >= 5 and exchanges == 0
        strip.close()

    # two planes, fields the function only copies (their stores are not skipped by the strip driver: its buffer
    # sets alternate like the pass driver's, but ghost rows come from the neighbour)
    cells = np.zeros((H, W), dtype=oracle.HOTSPOT_CELL)
    cells["temp"] = 320 + 10 * rng.random((H, W), dtype=np.float32)
    cells["power"] = rng.random((H, W), dtype=np.float32) * 0.01
    hp = oracle.hotspot_params(H, W)
    strip = capi.Strip("hotspot", capi.HotspotParams(hp.Rx_1, hp.Ry_1, hp.Rz_1, hp.Cap_1), bytes(8), H, W, 0, 1)
    strip.upload(0, np.ascontiguousarray(cells["temp"]))
    strip.upload(1, np.ascontiguousarray(cells["power"]))
    strip.advance(0, 27, blocking=True)
    want = oracle.hotspot(hp, cells, 27, n_threads=8)
    assert np.array_equal(bits(strip.download(0, np.float32)), bits(np.ascontiguousarray(want["temp"])))
    assert np.array_equal(bits(strip.download(1, np.float32)), bits(np.ascontiguousarray(want["power"])))
    strip.close()

    # Game of Life on words of four cells (width a multiple of four, dead halo)
    life = (rng.random((700, 1028)) < 0.35).astype(np.uint8)
    strip = capi.Strip("conway", capi.NoParams(), b"\0", 700, 1028, 0, 1)
    strip.upload(0, life)
    strip.advance(0, 19, blocking=True)
    assert np.array_equal(strip.download(0, np.uint8), oracle.conway(life, 19, n_threads=8))
    strip.close()


def _rank(rank, world, port, result_dir):
    """One of `world` processes sharing cuda:0; ghost rows through host memory over gloo."""
    import ctypes as C
    import sys

    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stencilstream_amd import capi

    torch.cuda.set_device(0)
    capi.init(0)
    lib = capi.load()

    from stencilstream_amd.dist import host_exchange_callback

    exchange = host_exchange_callback(rank, world)  # ghost rows staged through the host over gloo

    H, W = 1000, 900
    grid = np.random.default_rng(77).random((H, W), dtype=np.float32)
    for tag, coef, halo in (("general", [0.2, 0.21, 0.19, 0.22, 0.18], 0.5), ("uniform", [0.2] * 5, 0.0)):
        p = capi.JacobiParams()
        for i, c in enumerate(coef):
            p.coef[i] = c
        strip = capi.Strip("jacobi5general", p, np.float32(halo).tobytes(), H, W, rank, world, exchange=exchange)
        strip.upload(0, grid[strip.row_begin:strip.row_end])
        strip.warm_up()
        strip.advance(0, 25)
        strip.advance(25, 12, blocking=True)
        np.save(os.path.join(result_dir, f"{tag}{rank}.npy"), strip.download(0, np.float32))
        launches, exchanges = strip.counters()
        assert exchanges >= 2  # one per group of launches (STSTHIP_EXCHANGE_EVERY; default 4 for thin strips)
        strip.close()
    # two planes (HotSpot) and the Game of Life on words of four cells: the middle rank of three sweeps both of its
    # boundary bands as one launch with a row hole
    from oracle import oracle as O

    rng = np.random.default_rng(91)
    cells = np.zeros((H, W), dtype=O.HOTSPOT_CELL)
    cells["temp"] = 320 + 10 * rng.random((H, W), dtype=np.float32)
    cells["power"] = rng.random((H, W), dtype=np.float32) * 0.01
    hp = O.hotspot_params(H, W)
    strip = capi.Strip("hotspot", capi.HotspotParams(hp.Rx_1, hp.Ry_1, hp.Rz_1, hp.Cap_1), bytes(8), H, W, rank, world,
                       exchange=exchange)
    strip.upload(0, np.ascontiguousarray(cells["temp"][strip.row_begin:strip.row_end]))
    strip.upload(1, np.ascontiguousarray(cells["power"][strip.row_begin:strip.row_end]))
    strip.warm_up()
    strip.advance(0, 21, blocking=True)
    np.save(os.path.join(result_dir, f"hotspot{rank}.npy"), strip.download(0, np.float32))
    strip.close()
    life = (rng.random((H, W)) < 0.35).astype(np.uint8)
    strip = capi.Strip("conway", capi.NoParams(), b"\0", H, W, rank, world, exchange=exchange)
    strip.upload(0, life[strip.row_begin:strip.row_end])
    strip.warm_up()
    strip.advance(0, 23, blocking=True)
    np.save(os.path.join(result_dir, f"conway{rank}.npy"), strip.download(0, np.uint8))
    strip.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,every", [(2, "0"), (3, "0"), (2, "1"), (3, "2")])
def test_strips_of_several_processes_on_one_gpu(gpu, oracle, tmp_path, world, every, monkeypatch):
    """`every` = STSTHIP_EXCHANGE_EVERY: launches per ghost exchange (0: the driver's rule, four for strips this
    thin; 1: bands and exchange at every launch; 2: groups of two)."""
    import socket

    monkeypatch.setenv("STSTHIP_EXCHANGE_EVERY", every)  # inherited by the spawned ranks

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rank, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    grid = np.random.default_rng(77).random((1000, 900), dtype=np.float32)
    for tag, coef, halo in (("general", [0.2, 0.21, 0.19, 0.22, 0.18], 0.5), ("uniform", [0.2] * 5, 0.0)):
        want = oracle.jacobi("Jacobi5General", coef, grid, 37, halo=halo, n_threads=8)
        got = np.concatenate([np.load(tmp_path / f"{tag}{r}.npy") for r in range(world)], axis=0)
        assert np.array_equal(bits(got), bits(want)), tag
    rng = np.random.default_rng(91)  # the same draws as the ranks
    cells = np.zeros((1000, 900), dtype=oracle.HOTSPOT_CELL)
    cells["temp"] = 320 + 10 * rng.random((1000, 900), dtype=np.float32)
    cells["power"] = rng.random((1000, 900), dtype=np.float32) * 0.01
    want = oracle.hotspot(oracle.hotspot_params(1000, 900), cells, 21, n_threads=8)
    got = np.concatenate([np.load(tmp_path / f"hotspot{r}.npy") for r in range(world)], axis=0)
    assert np.array_equal(bits(got), bits(np.ascontiguousarray(want["temp"]))), "hotspot"
    life = (rng.random((1000, 900)) < 0.35).astype(np.uint8)
    got = np.concatenate([np.load(tmp_path / f"conway{r}.npy") for r in range(world)], axis=0)
    assert np.array_equal(got, oracle.conway(life, 23, n_threads=8)), "conway"


def _rccl_rank(rank, world, id_file, result_dir):
    """One process per GPU, ghost rows over RCCL: the configuration bench.py --gpus N runs."""
    import sys
    import time

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from stencilstream_amd import capi

    torch.cuda.set_device(rank)
    capi.init(rank)
    if rank == 0:
        with open(id_file + ".tmp", "wb") as f:
            f.write(capi.comm_unique_id())
        os.replace(id_file + ".tmp", id_file)
    while not os.path.exists(id_file):
        time.sleep(0.05)
    comm = capi.comm_create(open(id_file, "rb").read(), rank, world)
    H, W = 4000, 3000
    grid = np.random.default_rng(5).random((H, W), dtype=np.float32)
    p = capi.JacobiParams()
    for i in range(5):
        p.coef[i] = 0.2
    strip = capi.Strip("jacobi5general", p, np.float32(0.0).tobytes(), H, W, rank, world, comm=comm)
    strip.upload(0, grid[strip.row_begin:strip.row_end])
    strip.warm_up()
    strip.advance(0, 40)
    strip.advance(40, 9, blocking=True)
    np.save(os.path.join(result_dir, f"rccl{rank}.npy"), strip.download(0, np.float32))
    strip.close()


def test_strips_over_rccl_on_several_gpus(gpu, oracle, tmp_path):
    """Needs at least two GPUs in the box (the pool's test boxes have one: skipped there)."""
    import torch
    import torch.multiprocessing as mp

    world = min(torch.cuda.device_count(), 4)
    if world < 2:
        pytest.skip("one GPU visible: RCCL cannot join two ranks on one device")
    mp.spawn(_rccl_rank, args=(world, str(tmp_path / "rccl.id"), str(tmp_path)), nprocs=world, join=True)
    grid = np.random.default_rng(5).random((4000, 3000), dtype=np.float32)
    want = oracle.jacobi("Jacobi5General", [0.2] * 5, grid, 49, halo=0.0, n_threads=8)
    got = np.concatenate([np.load(tmp_path / f"rccl{r}.npy") for r in range(world)], axis=0)
    assert np.array_equal(bits(got), bits(want))


# ------------------------------------------------------------------ the RCCL exchange path on ONE GPU
# RCCL cannot join two ranks on one device, and the pool's boxes have one: with the default chain wiring a one-rank
# communicator posts nothing.  A rank may however name ITSELF as both neighbours (ststhip_comm_set_neighbours): a self
# send / receive inside one group is legal, and it runs everything a two-GPU exchange runs on this side of the wire --
# the dlopen'ed ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd, the ncclChar byte counts, the pointer arithmetic
# of the strip driver's exchange, the stream order between bands, exchange and interior.  The rank's upper rows arrive
# in its ghost rows BELOW and its lower rows in those ABOVE: the strip is a ring of one, i.e. periodic in the rows.


def _loopback_comm():
    from stencilstream_amd import capi

    comm = capi.comm_create(capi.comm_unique_id(), 0, 1)
    assert capi.comm_neighbours(comm) == (-1, -1)  # a chain of one: no neighbours
    capi.comm_set_neighbours(comm, 0, 0)
    assert capi.comm_neighbours(comm) == (0, 0)
    return comm


@pytest.mark.parametrize("elem_sizes", [(4,), (4, 4, 8, 1, 2, 4, 16, 4)], ids=["1plane", "8planes"])
@pytest.mark.parametrize("g", [16, 64])
def test_rccl_loopback_exchange_rows(gpu, elem_sizes, g):
    """ststhip_comm_exchange_rows against itself: g ghost rows of 65536 columns per side, 1 and 8 planes of different
    element sizes, on a stream of its own behind a kernel that writes the rows to send (stream order)."""
    import torch

    from stencilstream_amd import capi

    capi.init(0)
    comm = _loopback_comm()
    W, owned = 65536, 3 * g
    rows = owned + 2 * g
    gen = torch.Generator(device="cuda").manual_seed(1000 + g)
    planes = [torch.randint(0, 256, (rows, W * e), dtype=torch.uint8, device=gpu, generator=gen) for e in elem_sizes]
    before = [p.clone() for p in planes]
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for p, b in zip(planes, before):  # the producer of the rows to send runs on the exchange's stream, in front of it
            p[g:2 * g] += 1
            b[g:2 * g] += 1
    row_bytes = [W * e for e in elem_sizes]
    at = lambda p, r, rb: p.data_ptr() + r * rb  # noqa: E731
    capi.comm_exchange_rows(comm,
                            [at(p, g, rb) for p, rb in zip(planes, row_bytes)],              # send_up: first owned rows
                            [at(p, g + owned - g, rb) for p, rb in zip(planes, row_bytes)],  # send_down: last owned rows
                            [at(p, 0, rb) for p, rb in zip(planes, row_bytes)],              # recv_up: ghost rows above
                            [at(p, g + owned, rb) for p, rb in zip(planes, row_bytes)],      # recv_down: ghost rows below
                            row_bytes, g, stream=s.cuda_stream)
    s.synchronize()
    for i, (p, b) in enumerate(zip(planes, before)):
        assert torch.equal(p[g:g + owned], b[g:g + owned]), f"plane {i}: owned rows changed"
        assert torch.equal(p[:g], b[owned:owned + g]), f"plane {i}: ghost rows above are not the ring's lower rows"
        assert torch.equal(p[g + owned:], b[g:2 * g]), f"plane {i}: ghost rows below are not the ring's upper rows"
    capi.comm_destroy(comm)


def _ring_of_one_exchange():
    """The same wiring as an exchange callback: device-to-device copies in stream order (the reference the RCCL
    loopback is compared with)."""
    import ctypes as C

    from stencilstream_amd import capi

    lib = capi.load()

    def exchange(n_planes, send_up, send_down, recv_up, recv_down, row_bytes, n_rows, stream):
        for p in range(n_planes):
            n = row_bytes[p] * n_rows
            capi.check(lib.ststhip_memcpy_d2d(C.c_void_p(recv_down[p]), C.c_void_p(send_up[p]), n, C.c_void_p(stream)), "d2d")
            capi.check(lib.ststhip_memcpy_d2d(C.c_void_p(recv_up[p]), C.c_void_p(send_down[p]), n, C.c_void_p(stream)), "d2d")

    return exchange


@pytest.mark.parametrize("every", ["0", "1", "2"])
def test_strip_driver_over_rccl_loopback(gpu, oracle, monkeypatch, every):
    """ststhip_strip_advance with RCCL as the exchange, on one GPU: the middle strip of three whose communicator names
    itself on both sides.  Its rows then evolve like a grid that is periodic in the rows -- which the oracle computes
    on three copies of the strip stacked (the middle copy, while the true rim is further away than the generations
    reach) -- and the same strip driven by device-to-device copies as the exchange callback must agree bit for bit.
    Jacobi (both kernels, one plane) and HotSpot (two planes); every = STSTHIP_EXCHANGE_EVERY (launches per exchange)."""
    from stencilstream_amd import capi

    monkeypatch.setenv("STSTHIP_EXCHANGE_EVERY", every)
    capi.init(0)
    comm = _loopback_comm()
    R, W = 1000, 900
    rng = np.random.default_rng(4242)
    grid = rng.random((R, W), dtype=np.float32)

    def both(app, params, halo, uploads, n_first, n_second, planes):
        results = []
        for kwargs in ({"comm": comm}, {"exchange": _ring_of_one_exchange()}):
            strip = capi.Strip(app, params, halo, 3 * R, W, 1, 3, **kwargs)
            assert (strip.row_begin, strip.row_end) == (R, 2 * R)
            for i, a in enumerate(uploads):
                strip.upload(i, a)
            strip.warm_up()
            strip.advance(0, n_first)
            strip.advance(n_first, n_second, blocking=True)
            results.append([strip.download(i, dt) for i, dt in planes])
            launches, exchanges = strip.counters()
            assert exchanges >= 3
            strip.close()
        for a, b in zip(*results):
            assert np.array_equal(bits(a), bits(b)), f"{app}: RCCL loopback differs from the copy exchange"
        return results[0]

    for coef, halo in (([0.2, 0.21, 0.19, 0.22, 0.18], 0.25), ([0.2] * 5, 0.0)):
        p = capi.JacobiParams()
        for i, c in enumerate(coef):
            p.coef[i] = c
        got = both("jacobi5general", p, np.float32(halo).tobytes(), [grid], 40, 13, [(0, np.float32)])[0]
        want = oracle.jacobi("Jacobi5General", coef, np.tile(grid, (3, 1)), 53, halo=halo, n_threads=8)[R:2 * R]
        assert np.array_equal(bits(got), bits(want)), f"jacobi {coef[0]}"

    cells = np.zeros((R, W), dtype=oracle.HOTSPOT_CELL)
    cells["temp"] = 320 + 10 * rng.random((R, W), dtype=np.float32)
    cells["power"] = rng.random((R, W), dtype=np.float32) * 0.01
    hp = oracle.hotspot_params(3 * R, W)
    got = both("hotspot", capi.HotspotParams(hp.Rx_1, hp.Ry_1, hp.Rz_1, hp.Cap_1), bytes(8),
               [np.ascontiguousarray(cells["temp"]), np.ascontiguousarray(cells["power"])], 25, 8,
               [(0, np.float32), (1, np.float32)])
    want = oracle.hotspot(hp, np.tile(cells, (3, 1)), 33, n_threads=8)[R:2 * R]
    assert np.array_equal(bits(got[0]), bits(np.ascontiguousarray(want["temp"])))
    assert np.array_equal(bits(got[1]), bits(np.ascontiguousarray(want["power"])))
    capi.comm_destroy(comm)


def test_strip_driver_over_rccl_loopback_config5_shape(gpu):
    """The exchange of BASELINE configs[4] at its true message size: one GPU's strip of the 65536^2 grid's width
    (2048 x 65536 here; 16 generations x 2 launches per group = 32 ghost rows of 256 KiB per side and exchange),
    100 generations over the RCCL loopback against the copy exchange, bit for bit, with the launch / exchange counts."""
    import torch

    from stencilstream_amd import capi

    capi.init(0)
    comm = _loopback_comm()
    R, W = 2048, 65536
    p = capi.JacobiParams()
    for i in range(5):
        p.coef[i] = 0.2
    init = torch.rand(R, W, device=gpu, generator=torch.Generator(device="cuda").manual_seed(65536))
    torch.cuda.synchronize()
    outs = []
    for kwargs in ({"comm": comm}, {"exchange": _ring_of_one_exchange()}):
        strip = capi.Strip("jacobi5general", p, np.float32(0).tobytes(), 3 * R, W, 1, 3, **kwargs)
        strip.upload_from_device(0, init.data_ptr(), init.numel() * 4)
        strip.warm_up()
        strip.advance(0, 100, blocking=True)
        ptr, row_bytes = strip.plane(0)
        out = torch.empty_like(init)
        capi.check(capi.load().ststhip_memcpy_d2d(out.data_ptr(), ptr, R * row_bytes, None), "d2d")
        capi.check(capi.load().ststhip_stream_synchronize(None), "sync")
        outs.append(out)
        launches, exchanges = strip.counters()
        assert launches >= 7 and exchanges >= 2, (launches, exchanges)
        strip.close()
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert not torch.equal(outs[0], init)
    capi.comm_destroy(comm)


# ------------------------------------------------------------------ 2-D block decomposition (ststhip_block_*)
# the grid cut into mesh_rows x mesh_cols blocks, one process per block; ghost columns travel packed,
# corners by exchanging columns first and rows (over the full buffer width) second.


def test_single_block_equals_oracle(gpu, oracle):
    """A mesh of one block: the block driver's launches (column range named per launch, buffers with a column origin and
    ghost columns) against the oracle -- Jacobi both kernels, HotSpot on two planes, the Game of Life (cells, not words)."""
    from stencilstream_amd import capi

    capi.init(0)
    rng = np.random.default_rng(31)
    H, W = 700, 1100
    grid = rng.random((H, W), dtype=np.float32)
    for coef, halo in (([0.2, 0.21, 0.19, 0.22, 0.18], 0.25), ([0.2] * 5, 0.0)):
        p = capi.JacobiParams()
        for i, c in enumerate(coef):
            p.coef[i] = c
        block = capi.Block("jacobi5general", p, np.float32(halo).tobytes(), H, W, 0, 1, 1)
        assert (block.row_begin, block.row_end, block.col_begin, block.col_end) == (0, H, 0, W)
        block.upload(0, grid)
        block.advance(0, 29)
        block.advance(29, 8, blocking=True)
        want = oracle.jacobi("Jacobi5General", coef, grid, 37, halo=halo, n_threads=8)
        assert np.array_equal(bits(block.download(0, np.float32)), bits(want)), f"jacobi {coef[0]}"
        block.close()
    life = (rng.random((300, 1028)) < 0.35).astype(np.uint8)
    block = capi.Block("conway", capi.NoParams(), b"\0", 300, 1028, 0, 1, 1)
    block.upload(0, life)
    block.advance(0, 19, blocking=True)
    assert np.array_equal(block.download(0, np.uint8), oracle.conway(life, 19, n_threads=8))
    block.close()


def _mesh_rank(rank, mesh_rows, mesh_cols, port, result_dir):
    """One block of a mesh_rows x mesh_cols mesh; all processes share cuda:0, ghost cells through host memory over gloo."""
    import sys

    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    world = mesh_rows * mesh_cols
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stencilstream_amd import capi
    from stencilstream_amd.dist import host_exchange_callback

    torch.cuda.set_device(0)
    capi.init(0)
    r, c = divmod(rank, mesh_cols)
    rows_cb = host_exchange_callback(rank, world, rank - mesh_cols if r > 0 else None,
                                     rank + mesh_cols if r + 1 < mesh_rows else None)
    cols_cb = host_exchange_callback(rank, world, rank - 1 if c > 0 else None, rank + 1 if c + 1 < mesh_cols else None)
    H, W = 900, 1000
    grid = np.random.default_rng(177).random((H, W), dtype=np.float32)
    for tag, coef, halo in (("general", [0.2, 0.21, 0.19, 0.22, 0.18], 0.5), ("uniform", [0.2] * 5, 0.0)):
        p = capi.JacobiParams()
        for i, k in enumerate(coef):
            p.coef[i] = k
        block = capi.Block("jacobi5general", p, np.float32(halo).tobytes(), H, W, rank, mesh_rows, mesh_cols,
                           exchange_rows=rows_cb, exchange_cols=cols_cb)
        block.upload(0, grid[block.row_begin:block.row_end, block.col_begin:block.col_end])
        block.warm_up()
        block.advance(0, 25)
        block.advance(25, 12, blocking=True)
        np.save(os.path.join(result_dir, f"{tag}{rank}.npy"), block.download(0, np.float32))
        launches, exchanges = block.counters()
        assert launches >= 4 and exchanges >= 2
        block.close()
    from oracle import oracle as O

    rng = np.random.default_rng(191)
    cells = np.zeros((H, W), dtype=O.HOTSPOT_CELL)
    cells["temp"] = 320 + 10 * rng.random((H, W), dtype=np.float32)
    cells["power"] = rng.random((H, W), dtype=np.float32) * 0.01
    hp = O.hotspot_params(H, W)
    block = capi.Block("hotspot", capi.HotspotParams(hp.Rx_1, hp.Ry_1, hp.Rz_1, hp.Cap_1), bytes(8), H, W, rank, mesh_rows,
                       mesh_cols, exchange_rows=rows_cb, exchange_cols=cols_cb)
    sub = cells[block.row_begin:block.row_end, block.col_begin:block.col_end]
    block.upload(0, np.ascontiguousarray(sub["temp"]))
    block.upload(1, np.ascontiguousarray(sub["power"]))
    block.warm_up()
    block.advance(0, 21, blocking=True)
    np.save(os.path.join(result_dir, f"hotspot{rank}.npy"), block.download(0, np.float32))
    block.close()
    life = (rng.random((H, W)) < 0.35).astype(np.uint8)
    block = capi.Block("conway", capi.NoParams(), b"\0", H, W, rank, mesh_rows, mesh_cols, exchange_rows=rows_cb,
                       exchange_cols=cols_cb)
    block.upload(0, life[block.row_begin:block.row_end, block.col_begin:block.col_end])
    block.warm_up()
    block.advance(0, 23, blocking=True)
    np.save(os.path.join(result_dir, f"conway{rank}.npy"), block.download(0, np.uint8))
    block.close()
    # FDTD on two planes of 16-byte halves: two sub-iterations, time-dependent values (one device table per call), the
    # column halo of its real dependency cone (halo_columns_per_generation), coordinates in the transition function
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_parity_gpu import fdtd_setup

    po, pc, fcells = fdtd_setup(O, H, W)
    raw = fcells.view(np.float32).reshape(H, W, 8)
    block = capi.Block("fdtd_coef_grouped", pc, bytes(32), H, W, rank, mesh_rows, mesh_cols, exchange_rows=rows_cb,
                       exchange_cols=cols_cb)
    rs, cs = slice(block.row_begin, block.row_end), slice(block.col_begin, block.col_end)
    n_cols = block.col_end - block.col_begin
    block.upload(0, np.ascontiguousarray(raw[rs, cs, :4]).reshape(-1, n_cols * 4))
    block.upload(1, np.ascontiguousarray(raw[rs, cs, 4:]).reshape(-1, n_cols * 4))
    block.warm_up()
    block.advance(0, 30)
    block.advance(30, 11, blocking=True)
    fields = block.download(0, np.dtype("V16")).view(np.float32).reshape(-1, n_cols, 4)
    np.save(os.path.join(result_dir, f"fdtd{rank}.npy"), fields)
    block.close()
    dist.barrier()
    dist.destroy_process_group()


def _assemble(tmp_path, tag, mesh_rows, mesh_cols):
    return np.concatenate([np.concatenate([np.load(tmp_path / f"{tag}{r * mesh_cols + c}.npy") for c in range(mesh_cols)], axis=1)
                           for r in range(mesh_rows)], axis=0)


@pytest.mark.parametrize("mesh,every", [((2, 2), "0"), ((1, 3), "1"), ((3, 1), "0")], ids=["2x2", "1x3", "3x1"])
def test_blocks_of_several_processes_on_one_gpu(gpu, oracle, tmp_path, mesh, every, monkeypatch):
    """2 x 2, 1 x 3 and 3 x 1 processes on one GPU (the box allows six processes on the card, this one included) against the oracle's run of the whole grid, bit for bit: Jacobi (general and
    uniform-coefficient kernels), HotSpot on two planes, the Game of Life, FDTD (two planes of 16-byte halves, sub-iterations,
    time-dependent values over two calls); corners included (the inner corner of a 2 x 2 mesh is reached by all four
    blocks' ghost cells)."""
    import socket

    import torch.multiprocessing as mp

    monkeypatch.setenv("STSTHIP_EXCHANGE_EVERY", every)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_mesh_rank, args=(mesh[0], mesh[1], port, str(tmp_path)), nprocs=mesh[0] * mesh[1], join=True)
    H, W = 900, 1000
    grid = np.random.default_rng(177).random((H, W), dtype=np.float32)
    for tag, coef, halo in (("general", [0.2, 0.21, 0.19, 0.22, 0.18], 0.5), ("uniform", [0.2] * 5, 0.0)):
        want = oracle.jacobi("Jacobi5General", coef, grid, 37, halo=halo, n_threads=8)
        assert np.array_equal(bits(_assemble(tmp_path, tag, *mesh)), bits(want)), tag
    rng = np.random.default_rng(191)
    cells = np.zeros((H, W), dtype=oracle.HOTSPOT_CELL)
    cells["temp"] = 320 + 10 * rng.random((H, W), dtype=np.float32)
    cells["power"] = rng.random((H, W), dtype=np.float32) * 0.01
    want = oracle.hotspot(oracle.hotspot_params(H, W), cells, 21, n_threads=8)
    assert np.array_equal(bits(_assemble(tmp_path, "hotspot", *mesh)), bits(np.ascontiguousarray(want["temp"]))), "hotspot"
    life = (rng.random((H, W)) < 0.35).astype(np.uint8)
    assert np.array_equal(_assemble(tmp_path, "conway", *mesh), oracle.conway(life, 23, n_threads=8)), "conway"
    from test_parity_gpu import fdtd_setup

    po, _pc, fcells = fdtd_setup(oracle, H, W)
    want = oracle.fdtd(po, fcells, 41, n_threads=8).view(np.float32).reshape(H, W, 8)[:, :, :4]
    got = _assemble(tmp_path, "fdtd", *mesh)
    assert np.array_equal(bits(got), bits(np.ascontiguousarray(want))), "fdtd"
    assert np.abs(want[:, :, 2]).max() > 0


def test_block_driver_over_rccl_loopback(gpu, oracle):
    """The block driver with RCCL as the exchange on one GPU: the middle block of a 3 x 3 mesh whose communicator names
    itself on all four sides -- a torus of one.  Its cells then evolve like a grid that is periodic in rows AND columns,
    which the oracle computes on 3 x 3 copies of the block (the middle copy; the true rim is further away than the
    generations reach).  Ghost columns (packed, ncclSend / ncclRecv of the staging buffers) and corners included."""
    from stencilstream_amd import capi

    capi.init(0)
    comm = capi.comm_create(capi.comm_unique_id(), 0, 1)
    capi.comm_set_neighbours(comm, 0, 0)
    capi.comm_set_column_neighbours(comm, 0, 0)
    R, Wc = 600, 700
    rng = np.random.default_rng(555)
    grid = rng.random((R, Wc), dtype=np.float32)
    for coef, halo in (([0.2, 0.21, 0.19, 0.22, 0.18], 0.25), ([0.2] * 5, 0.0)):
        p = capi.JacobiParams()
        for i, c in enumerate(coef):
            p.coef[i] = c
        block = capi.Block("jacobi5general", p, np.float32(halo).tobytes(), 3 * R, 3 * Wc, 4, 3, 3, comm=comm)
        assert (block.row_begin, block.row_end, block.col_begin, block.col_end) == (R, 2 * R, Wc, 2 * Wc)
        block.upload(0, grid)
        block.warm_up()
        block.advance(0, 40)
        block.advance(40, 13, blocking=True)
        got = block.download(0, np.float32)
        launches, exchanges = block.counters()
        assert exchanges >= 3
        block.close()
        want = oracle.jacobi("Jacobi5General", coef, np.tile(grid, (3, 3)), 53, halo=halo, n_threads=8)[R:2 * R, Wc:2 * Wc]
        assert np.array_equal(bits(got), bits(want)), f"jacobi {coef[0]}"
    capi.comm_destroy(comm)


@pytest.mark.parametrize("every", ["0", "2"])
def test_strip_as_two_substrips_with_a_moving_boundary(gpu, oracle, monkeypatch, every):
    """The strip driver sweeps a tall enough strip as an upper and a lower launch per pass whose common boundary moves up
    by a launch's ghost rows (two launches in flight; ststhip_options::strip_substrips, here forced): the middle strip of
    three over the RCCL loopback, 600 generations in two calls -- more than the boundary's span, so it starts over in the
    middle of a call -- against the oracle on three stacked copies, and against the same strip swept as one launch per
    pass.  Both Jacobi kernels; the launch counters show that two sub-strips ran."""
    from stencilstream_amd import capi

    monkeypatch.setenv("STSTHIP_EXCHANGE_EVERY", every)
    monkeypatch.setenv("STSTHIP_SKEWED_STRIPS", "1")  # (the sub-strips are the strips' moving boundary inside a rank)
    capi.init(0)
    comm = _loopback_comm()
    R, W = 1536, 640
    rng = np.random.default_rng(1536)
    grid = rng.random((R, W), dtype=np.float32)
    for coef, halo in (([0.2, 0.21, 0.19, 0.22, 0.18], 0.25), ([0.2] * 5, 0.0)):
        p = capi.JacobiParams()
        for i, c in enumerate(coef):
            p.coef[i] = c
        results, launches = [], []
        for sub in ("2", "1"):
            monkeypatch.setenv("STSTHIP_STRIP_SUBSTRIPS", sub)
            capi.reload_options()
            strip = capi.Strip("jacobi5general", p, np.float32(halo).tobytes(), 3 * R, W, 1, 3, comm=comm)
            strip.upload(0, grid)
            strip.warm_up()
            strip.advance(0, 450)
            strip.advance(450, 150, blocking=True)
            results.append(strip.download(0, np.float32))
            launches.append(strip.counters()[0])
            strip.close()
        assert launches[0] > 1.5 * launches[1], launches  # two launches per pass (and the bands) against one
        want = oracle.jacobi("Jacobi5General", coef, np.tile(grid, (3, 1)), 600, halo=halo, n_threads=8)[R:2 * R]
        assert np.array_equal(bits(results[0]), bits(want)), f"two sub-strips, coefficients {coef[0]}"
        assert np.array_equal(bits(results[1]), bits(want)), f"one launch per pass, coefficients {coef[0]}"
    capi.comm_destroy(comm)
