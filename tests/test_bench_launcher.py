"""bench.py's own launcher: `--gpus N` must never silently run one rank."""
import os
import subprocess
import sys

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(kw)
    return env


def test_world_size_contradicting_gpus_is_an_error():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"],
                       env=_env(WORLD_SIZE="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "contradicts WORLD_SIZE" in r.stderr
    assert "{" not in r.stdout


def test_launcher_command(monkeypatch):
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    argv = ["--gpus", "3", "--steps", "2", "--cubes", "64"]
    rc = bench.launch_ranks(bench.parse(argv), argv)
    cmd = seen["cmd"]
    assert rc == 7                                   # the child's exit code is the parent's
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=3" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-len(argv):] == argv and cmd[-len(argv) - 1].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_gpus_2_without_launcher_starts_ranks_and_fails_without_a_gpu():
    """No GPU in the build container: the two ranks the launcher starts must each fail loudly (no CPU
    fallback), and the parent must hand that failure on -- not print an n_gpus = 1 line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: covered by tests/test_hip_multirank.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cubes", "16",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=_env(PHIFEM_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "no GPU visible" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_watchdog_leaves_with_its_exit_code_and_marker(tmp_path):
    """The RCCL watchdog (phifem_amd.dist_solver.Watchdog): a call WITHOUT a bound of its own (communicator set-up,
    self-test) that does not return within PHX_DIST_TIMEOUT_S makes the rank exit with code 87 and leave the marker
    file the launcher looks for; a call that returns in time leaves nothing; a call the library bounds itself
    (`bounded_inside`: the solve, whose every host wait has the limit) runs without a timer however long it takes,
    and a TimeoutError out of it takes the same exit.  PHIFEM_DIST_TIMEOUT_S is the older name of the variable."""
    import subprocess
    import sys
    mark = tmp_path / "mark"
    code = ("import time, sys\n"
            "sys.path.insert(0, %r)\n"
            "from phifem_amd.dist_solver import Watchdog\n"
            "with Watchdog('quick call', 0):\n"
            "    pass\n"
            "print('first ok', flush=True)\n"
            "with Watchdog('long healthy solve', 0, bounded_inside=True):\n"
            "    time.sleep(1.5)\n"
            "print('second ok', flush=True)\n"
            "with Watchdog('stuck collective', 3):\n"
            "    time.sleep(30)\n"
            "print('not reached')\n") % ROOT
    env = dict(os.environ, PHX_DIST_TIMEOUT_S="0.5", PHIFEM_WATCHDOG_FILE=str(mark))
    env.pop("PHIFEM_DIST_TIMEOUT_S", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 87, (r.returncode, r.stderr[-500:])
    assert "first ok" in r.stdout and "second ok" in r.stdout and "not reached" not in r.stdout
    assert "stuck collective" in r.stderr and "rank 3" in r.stderr
    assert mark.exists()
    # the older variable name still sets the limit, and a library timeout inside a bounded call leaves the same way
    mark.unlink()
    code2 = ("import sys\n"
             "sys.path.insert(0, %r)\n"
             "from phifem_amd.dist_solver import Watchdog, dist_timeout_s\n"
             "assert dist_timeout_s() == 7.0\n"
             "with Watchdog('solve', 1, bounded_inside=True):\n"
             "    raise TimeoutError('the stream did not drain')\n") % ROOT
    env2 = dict(os.environ, PHIFEM_DIST_TIMEOUT_S="7", PHIFEM_WATCHDOG_FILE=str(mark))
    env2.pop("PHX_DIST_TIMEOUT_S", None)
    r2 = subprocess.run([sys.executable, "-c", code2], env=env2, capture_output=True, text=True, timeout=120)
    assert r2.returncode == 87 and mark.exists(), (r2.returncode, r2.stderr[-500:])
