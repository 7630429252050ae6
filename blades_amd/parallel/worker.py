"""In-process worker for the sequential (reference-semantics) execution path.

Reference ``fllib/core/execution/worker.py``: a Ray actor owning one ``Task``
shared by the clients scheduled on it, swapping *optimizer state only* per
client (``switch_client``).  Here a Worker is a plain object in the rank's
process; ``get_session`` finds the current one through a module global instead
of Ray's private actor table.
"""
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Optional

import torch

from blades_amd.tasks import TaskSpec

_current_worker: Optional["Worker"] = None


def current_worker() -> Optional["Worker"]:
    return _current_worker


def set_current_worker(w: Optional["Worker"]) -> None:
    global _current_worker
    _current_worker = w


@dataclass
class WorkerSpec:
    task_spec: Optional[TaskSpec] = None
    worker_hyperparameters: Dict[str, Any] = field(default_factory=dict)
    worker_scaling_config: Any = None

    def get_params_dict(self) -> Dict[str, Any]:
        return {"task_spec": self.task_spec, "worker_scaling_config": self.worker_scaling_config}


class Worker:
    def __init__(self, task_spec: TaskSpec, worker_scaling_config=None, device=None):
        self._task_spec = task_spec
        self._scaling = worker_scaling_config
        if device is None:
            num_gpus = getattr(worker_scaling_config, "num_gpus_per_worker", 0) if worker_scaling_config else 0
            device = "cuda" if (num_gpus > 0 and torch.cuda.is_available()) else "cpu"
        self.device = device
        self._task = None
        self._client_states: Dict[str, Any] = {}
        self._current_client = None
        self.dataset = None
        set_current_worker(self)

    @property
    def distributed(self) -> int:
        return getattr(self._scaling, "num_workers", 0) if self._scaling else 0

    def setup(self):
        if self._task is None:
            self._task = self._task_spec.build(self.device)
        return self

    @property
    def task(self):
        if self._task is None:
            self.setup()
        return self._task

    def bind_task(self, task):
        """Shares an existing Task (e.g. the server's) instead of building one."""
        self._task = task

    def switch_client(self, client_id):
        """Swaps the shared task's optimizer state to ``client_id``'s."""
        if client_id == self._current_client:
            return
        task = self.task
        if task._optimizers is not None:
            if self._current_client is not None:
                self._client_states[self._current_client] = task.get_optimizer_states()
            if client_id in self._client_states:
                task.set_optimizer_states(self._client_states[client_id])
            else:
                # fresh optimizer state (new momentum buffers) for a new client
                for opt in task._optimizers:
                    opt.state.clear()
        self._current_client = client_id

    def apply(self, func: Callable, *args, **kwargs):
        set_current_worker(self)
        return func(self, *args, **kwargs)
