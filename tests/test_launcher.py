"""Launcher contract: env rendezvous variables per rank; a failing rank takes the others down.
(The reference has no launcher of its own — Ray brings its actors up, `worker_group.py:17-62`.)"""
import os
import subprocess
import sys
import textwrap
import time

import pytest

from blades_amd.parallel import launcher

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent("""
        import os, sys, time
        r = int(os.environ["RANK"])
        open(os.path.join(sys.argv[1], f"rank{r}.txt"), "w").write(
            " ".join(os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")))
        if sys.argv[2] == "fail":
            if r == 1:
                sys.exit(7)
            time.sleep(60)
    """))
    return str(p)


@pytest.mark.parametrize("native", [False, True])
def test_launch_env_and_teardown(tmp_path, native):
    script = _script(tmp_path)
    base = [sys.executable, "-m", "blades_amd.parallel.launcher", "--nproc", "3", "--master-port", "29641"]
    if native:
        launcher.build_native()
        base.append("--native")
    rc = subprocess.call(base + ["--", script, str(tmp_path), "ok"], cwd=ROOT)
    assert rc == 0
    for r in range(3):
        assert (tmp_path / f"rank{r}.txt").read_text() == f"{r} {r} 3 127.0.0.1 29641"
    t0 = time.time()
    rc = subprocess.call(base + ["--", script, str(tmp_path), "fail"], cwd=ROOT)
    assert rc == 7
    assert time.time() - t0 < 30, "surviving ranks were not taken down"


def test_launch_two_rank_allreduce(tmp_path):
    p = tmp_path / "ar.py"
    p.write_text(textwrap.dedent("""
        import torch, torch.distributed as dist
        from blades_amd.parallel.runtime import init_runtime, shutdown_runtime
        rt = init_runtime(use_gpu=False)
        t = torch.tensor([float(rt.rank + 1)])
        dist.all_reduce(t)
        assert float(t) == 3.0
        shutdown_runtime()
    """))
    env = dict(os.environ, PYTHONPATH=ROOT)
    rc = subprocess.call([sys.executable, "-m", "blades_amd.parallel.launcher", "--nproc", "2", "--", str(p)],
                         cwd=ROOT, env=env)
    assert rc == 0


def test_stall_watchdog_ends_a_hung_job(capfd):
    """A rank that stops marking progress (hung kernel / collective) ends the job with status 124 after
    ``stall_timeout_s`` — not after the process group's 600 s collective timeout."""
    import time

    from blades_amd.parallel.launcher import launch

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_stall_worker.py")
    t0 = time.time()
    rc = launch([sys.executable, worker], 2, stall_timeout_s=2.0, grace_s=2.0)
    assert rc == 124
    assert time.time() - t0 < 30
    err = capfd.readouterr().err
    assert "rank 1" in err and "no progress" in err


def test_native_stall_watchdog(capfd):
    exe = launcher.build_native()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_stall_worker.py")
    t0 = time.time()
    rc = subprocess.call([exe, "--nproc", "2", "--stall-timeout", "2", "--grace", "2", "--", sys.executable, worker])
    assert rc == 124 and time.time() - t0 < 30
    assert "rank 1" in capfd.readouterr().err
