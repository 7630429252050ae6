"""One-process-per-GPU launcher (replaces the reference's Ray actor bring-up,
``fllib/core/execution/worker_group.py:17-62``).

Two front ends over the same contract — every rank gets ``RANK``, ``LOCAL_RANK``,
``WORLD_SIZE``, ``MASTER_ADDR``, ``MASTER_PORT`` in its environment and
:func:`blades_amd.parallel.runtime.init_runtime` picks them up:

* :func:`launch` / ``python -m blades_amd.parallel.launcher`` — pure Python;
* ``blades_launch`` — the native launcher built from ``csrc/launch.cpp`` (see :func:`build_native`).

A rank that fails takes the others down (SIGTERM to the exact child pids, SIGKILL after a grace
period), so a dead rank cannot leave its peers blocked in an RCCL collective until the watchdog fires.
A rank that *hangs* (stuck kernel, deadlocked collective) is caught by the stall watchdog: ranks mark progress
through :func:`blades_amd.parallel.runtime.heartbeat` (once per FL round); with ``stall_timeout_s`` set, a rank that
has started beating and then stays silent that long ends the job with status 124 and a line naming the rank and
its last mark — the replacement for Ray's actor health probes (``fllib/core/execution/actor_manager.py``).
"""
import argparse
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time
from typing import List, Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
NATIVE = os.path.join(_HERE, "blades_launch")


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, addr: str, port: int, pin_gpus: bool = False,
             heartbeat_dir: Optional[str] = None) -> dict:
    env = dict(os.environ)
    if heartbeat_dir:
        env["BLADES_HEARTBEAT_DIR"] = heartbeat_dir
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR=addr, MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if pin_gpus:
        env.update(HIP_VISIBLE_DEVICES=str(rank), LOCAL_RANK="0", BLADES_LOCAL_RANK=str(rank))
    return env


def launch(cmd: Sequence[str], nproc: int, master_addr: str = "127.0.0.1", master_port: Optional[int] = None,
           pin_gpus: bool = False, grace_s: float = 10.0, poll_s: float = 0.05,
           stall_timeout_s: Optional[float] = None) -> int:
    """Run ``cmd`` as ``nproc`` ranks; returns the first non-zero exit status (0 if all succeed, 124 if the stall
    watchdog ended the job)."""
    port = master_port or free_port()
    if stall_timeout_s is None and os.environ.get("BLADES_STALL_TIMEOUT_S"):
        stall_timeout_s = float(os.environ["BLADES_STALL_TIMEOUT_S"])
    hb = tempfile.TemporaryDirectory(prefix="blades_hb_") if stall_timeout_s else None
    procs: List[subprocess.Popen] = [
        subprocess.Popen(list(cmd), env=rank_env(r, nproc, master_addr, port, pin_gpus, hb.name if hb else None))
        for r in range(nproc)]
    code, deadline = 0, None
    next_check = time.time() + 1.0
    try:
        while any(p.poll() is None for p in procs):
            if hb is not None and code == 0 and time.time() >= next_check:
                next_check = time.time() + min(1.0, stall_timeout_s / 4)
                for r, p in enumerate(procs):
                    path = os.path.join(hb.name, f"rank{r}")
                    if p.poll() is not None or not os.path.exists(path):
                        continue  # finished, or still starting up (first mark comes from init_runtime)
                    quiet = time.time() - os.path.getmtime(path)
                    if quiet > stall_timeout_s:
                        try:
                            tag = open(path).read()
                        except OSError:
                            tag = "?"
                        print(f"[launcher] rank {r} (pid {p.pid}) made no progress for {quiet:.0f} s "
                              f"(last mark: {tag!r}); ending the job", file=sys.stderr)
                        code, deadline = 124, time.time() + grace_s
                        for q in procs:
                            if q.poll() is None:
                                q.send_signal(signal.SIGTERM)
                        break
            for r, p in enumerate(procs):
                rc = p.poll()
                if rc not in (None, 0) and code == 0:
                    code = rc if rc > 0 else 128 - rc
                    print(f"[launcher] rank {r} (pid {p.pid}) ended with status {code}", file=sys.stderr)
                    deadline = time.time() + grace_s
                    for q in procs:
                        if q.poll() is None:
                            q.send_signal(signal.SIGTERM)
            if deadline is not None and time.time() > deadline:
                for q in procs:
                    if q.poll() is None:
                        q.kill()
                deadline = time.time() + 3600
            time.sleep(poll_s)
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.send_signal(signal.SIGTERM)
        code = code or 130
    for p in procs:
        rc = p.wait()
        if rc != 0 and code == 0:
            code = rc if rc > 0 else 128 - rc
    if hb is not None:
        hb.cleanup()
    return code


def build_native(verbose: bool = False) -> str:
    """Compile the C++ launcher next to this file (g++, no GPU code)."""
    src = os.path.join(_HERE, "csrc", "launch.cpp")
    if os.path.exists(NATIVE) and os.path.getmtime(NATIVE) >= os.path.getmtime(src):
        return NATIVE
    cmd = ["g++", "-O2", "-std=c++17", "-o", NATIVE, src]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return NATIVE


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--nproc-per-node", "--nproc", type=int, default=1)
    ap.add_argument("--master-addr", default="127.0.0.1")
    ap.add_argument("--master-port", type=int, default=None)
    ap.add_argument("--pin-gpus", action="store_true", help="one visible GPU per rank (HIP_VISIBLE_DEVICES)")
    ap.add_argument("--stall-timeout", type=float, default=None,
                    help="seconds without a progress mark from a rank before the job is ended (status 124)")
    ap.add_argument("--native", action="store_true", help="hand over to the C++ launcher")
    ap.add_argument("cmd", nargs=argparse.REMAINDER, help="-- program [args...]; a .py file is run with this python")
    a = ap.parse_args(argv)
    cmd = [c for c in a.cmd if c != "--"] if a.cmd[:1] == ["--"] else list(a.cmd)
    if not cmd:
        ap.error("nothing to launch")
    if cmd[0].endswith(".py") or cmd[0] == "-m":
        cmd = [sys.executable] + cmd
    if a.native:
        exe = build_native()
        full = [exe, "--nproc", str(a.nproc_per_node), "--master-addr", a.master_addr, "--master-port",
                str(a.master_port or free_port())] + (["--pin-gpus"] if a.pin_gpus else []) + (
                    ["--stall-timeout", str(int(a.stall_timeout))] if a.stall_timeout else []) + ["--"] + cmd
        return subprocess.call(full)
    return launch(cmd, a.nproc_per_node, a.master_addr, a.master_port, a.pin_gpus, stall_timeout_s=a.stall_timeout)


if __name__ == "__main__":
    sys.exit(main())
