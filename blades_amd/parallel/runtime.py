"""Process runtime: one PyTorch-ROCm process per GPU, RCCL over xGMI.

This replaces the reference's Ray actor layer (``fllib/core/execution``): ranks
are launched with ``python -m torch.distributed.run`` (or
``blades_amd.parallel.launcher``), read ``RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*``
from the environment and join one process group — backend ``nccl`` (= RCCL on
ROCm) when the rank owns a GPU, ``gloo`` otherwise (CPU tests).
"""
import datetime
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

from blades_amd.constants import DEFAULT_MASTER_ADDR, DEFAULT_MASTER_PORT


@dataclass
class Runtime:
    rank: int = 0
    world_size: int = 1
    local_rank: int = 0
    device: torch.device = torch.device("cpu")
    backend: Optional[str] = None
    owns_group: bool = False
    # ``BLADES_FORCE_COLLECTIVES=1``: join a process group and take every distributed code path (padded column
    # shards, exchange, all-reduces, object gathers) even with one rank.  This is how a one-GPU box exercises
    # the RCCL backend end to end (tests/test_rccl_single_rank.py).
    force_collectives: bool = False

    @property
    def is_main(self) -> bool:
        return self.rank == 0

    @property
    def distributed(self) -> bool:
        return self.world_size > 1 or self.force_collectives


_runtime: Optional[Runtime] = None


def heartbeat(tag: str = "") -> None:
    """Progress mark of this rank for the launcher's stall watchdog (``launcher.launch(stall_timeout_s=...)``):
    rewrites ``$BLADES_HEARTBEAT_DIR/rank<r>`` with the tag.  A no-op (one dict lookup) without a watchdog.  The
    algorithm calls it once per round, so a rank stuck in a kernel or waiting in a collective for a stuck peer stops
    beating and the launcher ends the job, naming the rank — long before the process group's own collective
    timeout (``BLADES_COLLECTIVE_TIMEOUT_S``, default 600 s), which stays as the backstop."""
    d = os.environ.get("BLADES_HEARTBEAT_DIR")
    if not d:
        return
    try:
        path = os.path.join(d, f"rank{os.environ.get('RANK', '0')}")
        with open(path + ".tmp", "w") as f:
            f.write(tag)
        os.replace(path + ".tmp", path)
    except OSError:
        pass


def init_runtime(use_gpu: Optional[bool] = None, timeout_s: Optional[int] = None) -> Runtime:
    """Idempotent.  ``use_gpu=None`` -> GPU iff one is visible."""
    global _runtime
    if _runtime is not None:
        return _runtime
    if timeout_s is None:
        timeout_s = int(os.environ.get("BLADES_COLLECTIVE_TIMEOUT_S", "600"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if use_gpu is None:
        use_gpu = torch.cuda.is_available()
    if use_gpu:
        ndev = torch.cuda.device_count()
        device = torch.device("cuda", local_rank % max(ndev, 1))
        torch.cuda.set_device(device)
    else:
        device = torch.device("cpu")
    backend = None
    owns = False
    force = os.environ.get("BLADES_FORCE_COLLECTIVES", "0") == "1"
    if world > 1 or force:
        backend = "nccl" if use_gpu else "gloo"
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", DEFAULT_MASTER_ADDR)
            os.environ.setdefault("MASTER_PORT", DEFAULT_MASTER_PORT)
            kwargs = {}
            if use_gpu:
                kwargs["device_id"] = device
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
            owns = True
    _runtime = Runtime(rank, world, local_rank, device, backend, owns, force)
    heartbeat("init")
    return _runtime


def get_runtime() -> Runtime:
    return _runtime if _runtime is not None else init_runtime()


def set_runtime(rt: Optional[Runtime]) -> None:
    """Override (tests / embedding in an existing process group)."""
    global _runtime
    _runtime = rt


def shutdown_runtime() -> None:
    global _runtime
    if _runtime is not None and _runtime.owns_group and dist.is_initialized():
        dist.destroy_process_group()
    _runtime = None
