"""Process-wide session (reference ``fllib/session.py:9-47``: ``FLSession``, ``get_global_model`` … resolved through
Ray's ``main_actor``).  One process per GPU means "the current actor" is simply this rank's worker."""
from typing import Optional

import torch

from blades_amd.parallel.worker import current_worker


class FLSession:
    @property
    def current_actor(self):
        return get_current_worker()


_fl_session: Optional[FLSession] = None


def get_current_worker():
    w = current_worker()
    if w is None:
        raise ValueError("no worker is bound to this process yet (build an Algorithm or a WorkerGroup first)")
    return w


def init_session() -> None:
    global _fl_session
    _fl_session = FLSession()


def get_session() -> FLSession:
    if _fl_session is None:
        raise Exception("FLSession not initialized")
    return _fl_session


def get_global_model(from_state: bool = False) -> torch.nn.Module:
    """The rank-local replica of the global model (every rank holds the same weights after a server step)."""
    return get_current_worker().task.model


def get_global_optimizer() -> torch.optim.Optimizer:
    opts = get_current_worker().task._optimizers
    if not opts:
        raise ValueError("the task has no optimizer yet (Client.setup / Task.init_optimizer creates it)")
    return opts[0]
