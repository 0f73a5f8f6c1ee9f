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
