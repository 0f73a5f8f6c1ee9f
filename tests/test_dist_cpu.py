"""The multi-GPU host logic (row strips, ghost exchange, boundary-first ordering, ragged pass
depths) on CPU with the gloo backend and world_size 2 and 3.  The sweep is injected: here the CPU
oracle plays the kernel so that the decomposition logic can be compared with the oracle run on the
whole grid.  (On GPUs the same class drives ststhip_app_sweep and RCCL.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class OracleSweep:
    """Test double for the HIP sweep with the same contract as ststhip_app_sweep."""

    n_planes = 1
    max_generations = 4
    halo_per_generation = 1

    def __init__(self, kind, coef=None, shape=None):
        self.kind, self.coef, self.shape = kind, coef, shape
        self.plane_elem_size = [4 if kind == "jacobi" else 1]
        if kind == "hotspot":  # two per-field planes (temp, power), as the HIP sweep of "hotspot" takes them
            self.n_planes, self.plane_elem_size = 2, [4, 4]
        self.calls = []

    def __call__(self, src, dst, dom, out_begin, out_end, iteration, depth, stream):
        from oracle import oracle as O

        assert out_begin < out_end
        dt = np.float32 if self.kind == "jacobi" else np.uint8
        H, origin = dom.global_height, dom.row_origin
        g = depth * self.halo_per_generation
        need_lo, need_hi = max(0, out_begin - g), min(H, out_end + g)
        assert need_lo >= origin and need_hi <= origin + dom.local_rows, "sweep would read rows it does not hold"
        if self.kind == "hotspot":
            # the window's own top / bottom rows reflect like grid edges; that error travels one row per
            # generation and never reaches the output rows, g rows away (at true grid edges it is exact)
            cells = np.zeros((need_hi - need_lo, dom.global_width), dtype=O.HOTSPOT_CELL)
            cells["temp"] = src[0].numpy().view(np.float32)[need_lo - origin:need_hi - origin]
            cells["power"] = src[1].numpy().view(np.float32)[need_lo - origin:need_hi - origin]
            out = self._hotspot_window(O, cells, need_lo, H, depth)
            for plane, field in ((0, "temp"), (1, "power")):
                dst[plane].numpy().view(np.float32)[out_begin - origin:out_end - origin] = \
                    out[field][out_begin - need_lo:out_end - need_lo]
            self.calls.append((out_begin, out_end, depth))
            return
        buf = src[0].numpy().view(dt)
        sub = np.ascontiguousarray(buf[need_lo - origin:need_hi - origin])
        if self.kind == "jacobi":
            out = O.jacobi("Jacobi5General", self.coef, sub, depth, halo=0.0)
        else:
            out = O.conway(sub, depth)
        dst[0].numpy().view(dt)[out_begin - origin:out_end - origin] = out[out_begin - need_lo:out_end - need_lo]
        self.calls.append((out_begin, out_end, depth))


    def _hotspot_window(self, O, cells, need_lo, H, depth):
        """The constants come from the GLOBAL grid shape (hotspot.cpp:281-295); the oracle reflects at the first and
        last row of whatever it is given, which is right where the window touches the global edge and lands on
        rows nobody reads where it does not."""
        p = O.hotspot_params(*self.shape)
        return O.hotspot(p, cells, depth)


def worker(rank, world, port, kind, H, W, gens, seed, result_dir, sub_strips=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys

    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stencilstream_amd.dist import StripDomain

    rng = np.random.default_rng(seed)
    if kind == "jacobi":
        grid = rng.random((H, W), dtype=np.float32)
        sweep = OracleSweep("jacobi", [0.2, 0.21, 0.19, 0.22, 0.18])
    elif kind == "hotspot":
        grid = (320 + 10 * rng.random((H, W), dtype=np.float32)).astype(np.float32)
        power = (rng.random((H, W), dtype=np.float32) * 0.01).astype(np.float32)
        sweep = OracleSweep("hotspot", shape=(H, W))
    else:
        grid = (rng.random((H, W)) < 0.4).astype(np.uint8)
        sweep = OracleSweep("conway")
    strip = StripDomain(None, None, None, H, W, rank, world, "cpu", sweep=sweep, sub_strips=sub_strips)
    a, b = strip.row_begin, strip.row_end
    if kind == "hotspot":
        strip.load_owned(torch.from_numpy(grid[a:b].copy()), torch.from_numpy(power[a:b].copy()))
    else:
        strip.load_owned(torch.from_numpy(grid[a:b].copy()))
    done = 0
    for chunk in gens:  # several advance() calls: resume semantics
        strip.advance(done, chunk)
        done += chunk
    mine = strip.owned(0, torch.uint8 if kind == "conway" else torch.float32).numpy()
    np.save(os.path.join(result_dir, f"rank{rank}.npy"), mine)
    if kind == "hotspot":  # the second plane travels with the first and comes back unchanged
        assert np.array_equal(strip.owned(1, torch.float32).numpy(), power[a:b])
    # boundary bands must be swept before the interior in every pass: the first launch is a band
    # (at most one halo depth of rows) next to a neighbour
    first = sweep.calls[0]
    assert first[1] - first[0] <= sweep.max_generations * sweep.halo_per_generation
    if sub_strips in (None, 1):
        assert (first[0] == a and rank > 0) or (first[1] == b and rank + 1 < world)
    # every pass covers the owned rows exactly once, whatever the sub-strips
    depths = [d for chunk in gens for d in __import__("stencilstream_amd.dist", fromlist=["x"]).pass_depths(
        chunk, sweep.max_generations)]
    at = 0
    for depth in depths:
        rows = []
        while sum(e - s for s, e in rows) < b - a:
            s0, e0, d0 = sweep.calls[at]
            assert d0 == depth
            rows.append((s0, e0))
            at += 1
        rows.sort()
        assert rows[0][0] == a and rows[-1][1] == b
        assert all(rows[i][1] == rows[i + 1][0] for i in range(len(rows) - 1))
    assert at == len(sweep.calls)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("kind,H,W,gens", [("jacobi", 97, 40, [7]), ("jacobi", 64, 33, [4, 4, 1]),
                                           ("conway", 90, 50, [10]), ("hotspot", 84, 37, [9, 3])])
def test_strips_equal_whole_grid(oracle, tmp_path, world, kind, H, W, gens):
    seed = 1234
    mp.spawn(worker, args=(world, free_port(), kind, H, W, gens, seed, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(seed)
    total = sum(gens)
    if kind == "jacobi":
        grid = rng.random((H, W), dtype=np.float32)
        want = oracle.jacobi("Jacobi5General", [0.2, 0.21, 0.19, 0.22, 0.18], grid, total, halo=0.0)
    elif kind == "hotspot":
        cells = np.zeros((H, W), dtype=oracle.HOTSPOT_CELL)
        cells["temp"] = (320 + 10 * rng.random((H, W), dtype=np.float32)).astype(np.float32)
        cells["power"] = (rng.random((H, W), dtype=np.float32) * 0.01).astype(np.float32)
        want = np.ascontiguousarray(oracle.hotspot(oracle.hotspot_params(H, W), cells, total)["temp"])
    else:
        grid = (rng.random((H, W)) < 0.4).astype(np.uint8)
        want = oracle.conway(grid, total)
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)], axis=0)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


@pytest.mark.parametrize("world,sub_strips", [(2, 2), (2, 3), (3, 2)])
def test_sub_strips_equal_whole_grid(oracle, tmp_path, world, sub_strips):
    """Sub-strips inside a rank (the tail-overlap scheme of the single-GPU pass driver) change the
    launch pattern only, never the result; ragged depths 4+4+1 and a resume."""
    seed, H, W, gens = 77, 131, 37, [9, 6]
    mp.spawn(worker, args=(world, free_port(), "jacobi", H, W, gens, seed, str(tmp_path), sub_strips),
             nprocs=world, join=True)
    grid = np.random.default_rng(seed).random((H, W), dtype=np.float32)
    want = oracle.jacobi("Jacobi5General", [0.2, 0.21, 0.19, 0.22, 0.18], grid, sum(gens), halo=0.0)
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)], axis=0)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_split_rows_and_depths():
    from stencilstream_amd.dist import pass_depths, split_rows

    assert split_rows(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert split_rows(16384 * 8, 8)[7] == (16384 * 7, 16384 * 8)
    assert pass_depths(1000, 8) == [8] * 125
    assert pass_depths(13, 8) == [8, 4, 1]
    assert pass_depths(0, 8) == []
    assert pass_depths(5, 1) == [1] * 5


def test_thin_strips_are_rejected():
    from stencilstream_amd.dist import StripDomain

    with pytest.raises(ValueError):
        StripDomain(None, None, None, 20, 8, 0, 4, "cpu", sweep=OracleSweep("jacobi", [0.2] * 5))
