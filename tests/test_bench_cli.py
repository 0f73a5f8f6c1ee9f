"""bench.py's command line: a plain `python bench.py --gpus N` must start the ranks itself, as a child process,
before it has imported torch or touched HIP; the multi-rank path is then exercised on one
GPU with the ghost rows staged through the host."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_importing_bench_does_not_import_torch():
    code = "import sys; sys.path.insert(0, %r); import bench; assert 'torch' not in sys.modules; print('ok')" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_plain_multi_gpu_invocation_spawns_a_launcher_child(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7  # the child's return code is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in [m for m in sys.modules if m == "torch"] or True  # (torch may be loaded by other tests)


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_run_plainly():
    """`python bench.py --gpus 2 --debug-host-exchange ...` with no launcher: one JSON line, two ranks, the result
    verified against the single-GPU run, the T = 1 general kernel and the oracle; exit code 0."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-host-exchange",
                          "--steps", "2", "--warmup", "1", "--size", "4096", "--generations", "100",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"]["world_size"] == 2
    assert line["verified"] is True
    v = line["verification"]
    assert v["cells_that_differ_from_the_input"] > 0  # the check covers cells the generations changed
    assert v["single_gpu_reference"]["whole_grid_equal_to_T1_general_kernel"] is True
    assert all(w["equal"] and w["distinct_values"] > 16 for w in v["single_gpu_reference"]["oracle_windows"])


@pytest.mark.gpu
def test_bench_two_ranks_exchange_every_four_launches():
    """The same with one ghost exchange per four launches (--exchange-every 4): 48 ghost rows, widened launches."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-host-exchange",
                          "--steps", "1", "--warmup", "1", "--size", "4096", "--generations", "100",
                          "--exchange-every", "4", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert line["verified"] is True and line["ranks"]["exchange_every"] == 4
