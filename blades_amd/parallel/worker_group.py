"""WorkerGroup facade.

Reference ``fllib/core/execution/worker_group.py``: Ray Train ``BackendExecutor``
+ a fault-tolerant actor pool with data affinity.  The MI355X design has no
actors — one process per GPU, the client→rank map is static
(``ops.host.lpt_assign``) — so the facade keeps the *call surface*
(``sync_state`` / ``foreach_actor`` / ``execute_with_actor_pool`` /
``local_execute`` / ``local_worker``) over the rank-local worker.  Calls that
the reference fans out to remote actors run on the local worker; cross-rank
data movement is done by the engine with RCCL collectives, not through here.
"""
from typing import Any, Callable, List, Sequence

from .scaling_config import WorkerGroupScalingConfig
from .worker import Worker, WorkerSpec


class _Result:
    def __init__(self, value=None, error=None, actor_id=0):
        self._value, self._error, self.actor_id = value, error, actor_id
        self.ok = error is None

    def get(self):
        if self._error is not None:
            raise self._error
        return self._value


class _ResultList(list):
    def ignore_errors(self):
        return _ResultList(r for r in self if r.ok)


class WorkerGroup:
    def __init__(self, worker_spec: WorkerSpec, max_remote_requests_in_flight_per_actor: int = 1, device=None):
        scaling = worker_spec.worker_scaling_config or WorkerGroupScalingConfig()
        self._scaling = scaling
        self._worker = Worker(worker_spec.task_spec, scaling, device=device)
        self._is_local = True
        from .actor_manager import ActorManager

        self._manager = ActorManager(self._worker, max_remote_requests_in_flight_per_actor=
                                     max_remote_requests_in_flight_per_actor)

    @property
    def workers(self) -> list:
        """Remote workers — always empty (there are none in this design)."""
        return []

    def local_worker(self) -> Worker:
        return self._worker

    def sync_state(self, state_name: str, state: Any):
        from blades_amd.constants import GLOBAL_MODEL

        if state_name == GLOBAL_MODEL:
            self._worker.task.model.load_state_dict(state)
        else:
            setattr(self._worker, state_name, state)

    def sync_weights(self, weights):
        self._worker.task.set_weights(weights)

    def foreach_actor(self, func, *, healthy_only=True, remote_actor_ids=None, timeout_seconds=None):
        funcs = func if isinstance(func, (list, tuple)) else [func]
        out = _ResultList()
        for f in funcs:
            try:
                out.append(_Result(self._worker.apply(f)))
            except Exception as e:  # mirror FaultTolerantActorManager: collect, don't raise
                out.append(_Result(error=e))
        return out

    def execute_with_actor_pool(self, func: Callable, values: Sequence, affinity_actors=None) -> List:
        """Collective over the ranks of the job (``parallel/actor_manager.py``): each value runs on one rank of its
        affinity list and every rank receives all results in submission order.  Single process: a plain loop."""
        return self._manager.execute_with_actor_pool(func, values, affinity_actors)

    def local_execute(self, func: Callable, values: Sequence) -> List:
        return [self._worker.apply(func, v) for v in values]

    def shutdown(self):
        pass
