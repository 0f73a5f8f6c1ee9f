"""A source that is still arriving (ABI 6, ststhip_set_source_arrival): the pass driver starts on the row blocks of an
upload that have arrived -- its first passes run as row tiles skewed in time behind the blocks -- instead of idling
for the whole transfer.  Which passes run as tiles is decided while they run (a timing matter); the RESULT must be
that of the same call on a grid that is already in HBM, bit for bit, and equal to the oracle's.

Through the C ABI, as a foreign-language host would use it: pinned host memory, copies in row blocks on the runtime's
upload stream, one event per block."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def _run(app, tf_params, halo_bytes, planes, n, offset=0, blocks=0, stream_upload=True):
    """One ststhip_app_run over `planes` (host arrays, H x W x bytes); blocks > 0: uploaded in that many row blocks
    and followed by the driver, else in HBM before the call.  Returns (result planes, run info)."""
    import torch

    from stencilstream_amd import capi

    capi.init(0)
    h, w = planes[0].shape[:2]
    dom = capi.Domain(h, w, 0, h, w)
    dev = [torch.empty(p.size, dtype=torch.uint8, device="cuda") for p in planes]
    out = [torch.empty(p.size, dtype=torch.uint8, device="cuda") for p in planes]
    events = []
    if blocks:
        pinned = [torch.from_numpy(p.reshape(-1)).pin_memory() for p in planes]
        torch.cuda.synchronize()
        events = capi.upload_in_blocks(
            [(pin.data_ptr(), d.data_ptr(), p.shape[1] * p.shape[2]) for pin, d, p in zip(pinned, dev, planes)],
            h, n_blocks=blocks)
    else:
        for d, p in zip(dev, planes):
            d.copy_(torch.from_numpy(p.reshape(-1)))
    torch.cuda.synchronize()
    info = capi.app_run(app, tf_params, halo_bytes, dom, [d.data_ptr() for d in dev], [o.data_ptr() for o in out],
                        offset, n, blocking=True)
    capi.events_destroy(events)
    return [o.cpu().numpy().reshape(p.shape) for o, p in zip(out, planes)], info


def _jacobi_params(coef):
    from stencilstream_amd import capi

    p = capi.JacobiParams()
    for i, c in enumerate(coef):
        p.coef[i] = c
    return p


CASES = [
    # (rows, columns, generations, blocks)
    (2048, 1024, 200, 8),   # long enough for the depth probes behind the streamed passes
    (2048, 1024, 40, 4),    # no probes
    (1536, 640, 64, 6),     # ragged blocks; few generations: most of the call runs behind the source
    (4100, 520, 333, 5),    # generations that are no multiple of any depth
]


@pytest.mark.parametrize("rows,cols,n,blocks", CASES, ids=str)
@pytest.mark.parametrize("coef", [[0.11, 0.19, 0.23, 0.31, 0.16], [0.2] * 5], ids=["general", "uniform"])
def test_jacobi_behind_an_arriving_source(gpu, oracle, rows, cols, n, blocks, coef):
    rng = np.random.default_rng(rows + cols + n)
    grid = rng.random((rows, cols), dtype=np.float32)
    planes = [grid.view(np.uint8).reshape(rows, cols, 4)]
    halo = np.float32(0.0).tobytes()
    plain, info0 = _run("jacobi5general", _jacobi_params(coef), halo, planes, n)
    assert info0.n_streamed_passes == 0
    streamed, info = _run("jacobi5general", _jacobi_params(coef), halo, planes, n, blocks=blocks)
    assert info.n_streamed_passes >= 2, "the driver did not follow the blocks"
    assert np.array_equal(streamed[0], plain[0])
    want = oracle.jacobi("Jacobi5General", coef, grid, n, halo=0.0, n_threads=8)
    assert np.array_equal(streamed[0].reshape(-1), bits(want).reshape(-1))
    # ... and again: the depth for this shape is known now (no probes), the plan behind the tiles is another
    again, info2 = _run("jacobi5general", _jacobi_params(coef), halo, planes, n, blocks=blocks)
    assert info2.n_streamed_passes >= 2
    assert np.array_equal(again[0], plain[0])


def test_blocks_the_driver_cannot_follow(gpu, oracle, monkeypatch):
    """Blocks of fewer rows than two launches' ghost rows, and STSTHIP_STREAM_UPLOAD=0: the call waits for the whole
    upload itself (the caller never does) and gives the same cells."""
    rng = np.random.default_rng(5)
    grid = rng.random((300, 700), dtype=np.float32)
    planes = [grid.view(np.uint8).reshape(300, 700, 4)]
    coef = [0.3, 0.1, 0.2, 0.15, 0.25]
    want = oracle.jacobi("Jacobi5General", coef, grid, 50, halo=0.0, n_threads=8)
    got, info = _run("jacobi5general", _jacobi_params(coef), np.float32(0).tobytes(), planes, 50, blocks=20)
    assert info.n_streamed_passes == 0
    assert np.array_equal(got[0].reshape(-1), bits(want).reshape(-1))
    monkeypatch.setenv("STSTHIP_STREAM_UPLOAD", "0")
    grid = rng.random((2048, 512), dtype=np.float32)
    planes = [grid.view(np.uint8).reshape(2048, 512, 4)]
    want = oracle.jacobi("Jacobi5General", coef, grid, 50, halo=0.0, n_threads=8)
    got, info = _run("jacobi5general", _jacobi_params(coef), np.float32(0).tobytes(), planes, 50, blocks=4)
    assert info.n_streamed_passes == 0
    assert np.array_equal(got[0].reshape(-1), bits(want).reshape(-1))


def test_conway_behind_an_arriving_source(gpu, oracle):
    """One-byte cells on words of four (the packed form) and as bytes (odd width)."""
    from stencilstream_amd import capi

    for rows, cols in ((2048, 1024), (1800, 1001)):
        rng = np.random.default_rng(rows)
        grid = (rng.random((rows, cols)) < 0.35)
        planes = [grid.view(np.uint8).reshape(rows, cols, 1)]
        got, info = _run("conway", capi.NoParams(), b"\x00", planes, 300, blocks=6)
        assert info.n_streamed_passes >= 2
        want = oracle.conway(grid, 300, n_threads=8)
        assert np.array_equal(got[0].reshape(rows, cols), want.view(np.uint8))


def test_hotspot_planes_behind_an_arriving_source(gpu, oracle):
    """Two planes per block (temperature, power): the constant plane is not stored again from the third pass on --
    tiles included."""
    from stencilstream_amd import capi

    rows, cols, n = 2048, 768, 120
    rng = np.random.default_rng(77)
    cells = np.zeros((rows, cols), dtype=oracle.HOTSPOT_CELL)
    cells["temp"] = 320 + 10 * rng.random((rows, cols), dtype=np.float32)
    cells["power"] = rng.random((rows, cols), dtype=np.float32) * 0.01
    hp = oracle.hotspot_params(rows, cols)
    params = capi.HotspotParams(hp.Rx_1, hp.Ry_1, hp.Rz_1, hp.Cap_1)
    planes = [np.ascontiguousarray(cells[f]).view(np.uint8).reshape(rows, cols, 4) for f in ("temp", "power")]
    got, run = _run("hotspot", params, bytes(8), planes, n, blocks=8)
    assert run.n_streamed_passes >= 2
    want = oracle.hotspot(hp, cells, n, n_threads=8)
    assert np.array_equal(got[0].reshape(-1), bits(np.ascontiguousarray(want["temp"])).reshape(-1))
    assert np.array_equal(got[1].reshape(-1), bits(np.ascontiguousarray(want["power"])).reshape(-1))


def test_fdtd_behind_an_arriving_source(gpu, oracle):
    """Sub-iterations (two ghost rows per generation), the call's device table of time-dependent values, a plane of
    coefficients that is never stored again, coordinates in the transition function."""
    from test_parity_gpu import fdtd_setup

    rows, cols, n = 1536, 512, 60
    po, pc, fcells = fdtd_setup(oracle, rows, cols)
    raw = fcells.view(np.float32).reshape(rows, cols, 8)
    planes = [np.ascontiguousarray(raw[:, :, :4]).view(np.uint8).reshape(rows, cols, 16),
              np.ascontiguousarray(raw[:, :, 4:]).view(np.uint8).reshape(rows, cols, 16)]
    got, run = _run("fdtd_coef_grouped", pc, bytes(32), planes, n, blocks=6)
    assert run.n_streamed_passes >= 2
    want = oracle.fdtd(po, fcells, n, n_threads=8).view(np.float32).reshape(rows, cols, 8)
    assert np.array_equal(got[0].reshape(-1), bits(np.ascontiguousarray(want[:, :, :4])).reshape(-1))
    assert np.array_equal(got[1].reshape(-1), bits(np.ascontiguousarray(want[:, :, 4:])).reshape(-1))
