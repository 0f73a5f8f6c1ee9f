"""Runs the C++ API test binaries (tests/cpp): the reference's unit tests of Stencil, Grid and
StencilUpdate restated for stencil::cpu (host) and stencil::hip (MI355X, through the templates)."""
import os
import subprocess

import pytest

from conftest import ROOT

OUT = os.path.join(ROOT, "build", "tests")


def test_host_api():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), os.path.join(OUT, "host_api_test")])
    res = subprocess.run([os.path.join(OUT, "host_api_test")], capture_output=True, timeout=600)
    assert res.returncode == 0, res.stdout.decode() + res.stderr.decode()
    assert b"0 failures" in res.stdout


@pytest.mark.gpu
def test_hip_api():
    binary = os.path.join(OUT, "hip_api_test")
    if not os.path.exists(binary):
        pytest.fail("build/tests/hip_api_test missing: run __graft_entry__.build()")
    res = subprocess.run([binary], capture_output=True, timeout=600)
    assert res.returncode == 0, res.stdout.decode() + res.stderr.decode()
    assert b"0 failures" in res.stdout


@pytest.mark.gpu
def test_grid_upload_does_not_block_and_host_writes_wait_for_it():
    """tests/cpp/grid_upload_test.hip: hip::Grid queues its upload in front of the update and returns; a write
    accessor, copy_from_buffer and the grid's destruction wait for it."""
    binary = os.path.join(OUT, "grid_upload_test")
    if not os.path.exists(binary):
        pytest.fail("build/tests/grid_upload_test missing: run __graft_entry__.build()")
    res = subprocess.run([binary], capture_output=True, timeout=600)
    assert res.returncode == 0, res.stdout.decode() + res.stderr.decode()
    assert b"0 failures" in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_strip_driver_from_a_plain_cpp_host(ranks):
    """tests/cpp/strip_host_test.cpp: a g++-built C++ program (no Python, no torch, no HIP headers) forks `ranks`
    processes that each own one strip on cuda:0 and call ststhip_strip_create / advance; ghost rows through a
    shared-memory mailbox (the exchange callback); the pieces equal the single-strip run bit for bit."""
    binary = os.path.join(OUT, "strip_host_test")
    if not os.path.exists(binary):
        pytest.fail("build/tests/strip_host_test missing: run __graft_entry__.build()")
    res = subprocess.run([binary, str(ranks)], capture_output=True, timeout=600)
    assert res.returncode == 0, res.stdout.decode() + res.stderr.decode()
    assert b"0 cells differ" in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_strip_update_template_with_user_functors(tmp_path, ranks):
    """tests/cpp/strip_template_test.hip: stencil::hip::StripUpdate -- USER transition functions (time-dependent values
    and sub-iterations on AoS cells; a two-field cell on per-field planes) over `ranks` processes on cuda:0, ghost rows
    through a mailbox file; the rows of all ranks equal the single-strip run, which the binary itself checks against
    hip::StencilUpdate on the whole grid."""
    import numpy as np

    binary = os.path.join(OUT, "strip_template_test")
    if not os.path.exists(binary):
        pytest.fail("build/tests/strip_template_test missing: run __graft_entry__.build()")
    size, ramp_at, plate_at, height, width = (int(x) for x in subprocess.check_output([binary, "layout"]).split())

    def run(n, path):
        with open(path, "wb") as f:
            f.truncate(size)
        procs = [subprocess.Popen([binary, str(r), str(n), str(path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                 for r in range(n)]
        for p in procs:
            out, _ = p.communicate(timeout=600)
            assert p.returncode == 0, out.decode()
        return np.fromfile(path, dtype=np.uint8)

    many = run(ranks, tmp_path / "mailbox_many")
    one = run(1, tmp_path / "mailbox_one")
    cells = height * width
    assert np.array_equal(many[ramp_at:ramp_at + 4 * cells], one[ramp_at:ramp_at + 4 * cells]), "Ramp"
    assert np.array_equal(many[plate_at:plate_at + 8 * cells], one[plate_at:plate_at + 8 * cells]), "Conduction"
    assert one[ramp_at:ramp_at + 4 * cells].view(np.float32).std() > 0


@pytest.mark.gpu
def test_block_update_template_with_user_functors():
    """tests/cpp/block_template_test.hip: stencil::hip::BlockUpdate -- USER transition functions (time-dependent values,
    sub-iterations and the cells' own coordinates on AoS cells; a two-field cell on per-field planes) on meshes of 1 x 1,
    2 x 2, 1 x 3 and 3 x 1 blocks (threads of one process on cuda:0, ghost columns and rows through an in-process mailbox):
    every mesh's assembled cells equal hip::StencilUpdate on the whole grid, bit for bit."""
    binary = os.path.join(OUT, "block_template_test")
    if not os.path.exists(binary):
        pytest.fail("build/tests/block_template_test missing: run __graft_entry__.build()")
    res = subprocess.run([binary], capture_output=True, timeout=600)
    assert res.returncode == 0, (res.stdout + res.stderr).decode()[-2000:]
    for mesh in (b"1 x 1", b"2 x 2", b"1 x 3", b"3 x 1"):
        assert b"mesh " + mesh + b" equals hip::StencilUpdate" in res.stdout
