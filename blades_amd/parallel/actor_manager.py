"""Task farm over ranks: the SPMD counterpart of the reference's actor pool.

Reference: ``fllib/core/execution/actor_manager.py:7-81`` (``_ActorPool.submit`` with an affinity list, ordered
``get_next`` drain) and its older duplicate ``fllib/utils/actor_pool.py``.  There, a driver process hands one
closure per client to whichever *idle* Ray actor of the client's affinity list answers first.

Here there is no driver and there are no actors: every rank runs the same program.  ``execute_with_actor_pool``
is therefore a *collective*: every rank computes the same deterministic value→rank assignment (least-loaded rank
of each value's affinity list, ties to the lower rank — what an idle-first pool converges to when all tasks cost
the same), runs the values it owns on its local worker, and one ``all_gather_object`` hands every rank the
complete result list in submission order.  With ``world_size == 1`` it degenerates to a sequential loop.

The bulk client-training path never goes through here (it is the batched engine plus RCCL all-to-all); this is
the general-purpose "run this closure once per item somewhere" utility — evaluation sweeps, per-client
bookkeeping, user extensions written against the reference API.
"""
from collections import deque
from typing import Any, Callable, List, Optional, Sequence

import torch.distributed as dist


def assign_to_ranks(n_values: int, world_size: int, affinity: Optional[Sequence[Optional[Sequence[int]]]] = None,
                    cost: Optional[Sequence[float]] = None) -> List[int]:
    """Deterministic value→rank map.  ``affinity[i]`` (ranks allowed to run value ``i``; ``None`` = any) mirrors the
    reference's ``affinity_actors``; ``cost`` weights the load balance (default: every value costs 1)."""
    if affinity is not None and len(affinity) != n_values:
        raise ValueError(f"Number of values ({n_values}) does not match number of affinity_actors ({len(affinity)})")
    load = [0.0] * world_size
    owner = []
    for i in range(n_values):
        allowed = range(world_size) if affinity is None or affinity[i] is None else affinity[i]
        allowed = [int(r) for r in allowed]
        if not allowed or min(allowed) < 0 or max(allowed) >= world_size:
            raise ValueError(f"value {i}: affinity {list(allowed)} names no rank of a {world_size}-rank job")
        r = min(allowed, key=lambda a: (load[a], a))
        load[r] += 1.0 if cost is None else float(cost[i])
        owner.append(r)
    return owner


class _ActorPool:
    """``ray.util.ActorPool`` call surface (``submit`` / ``has_next`` / ``get_next`` / ``map``) over in-process
    workers.  Work is executed at ``get_next`` time, in submission order, on the first worker of the task's affinity
    list (or round-robin without one) — there is no concurrency to schedule around inside one rank: the GPU is
    already saturated by the batched kernels a task launches."""

    def __init__(self, actors: Sequence[Any]):
        self._actors = list(actors)
        if not self._actors:
            raise ValueError("actor pool needs at least one worker")
        self._pending = deque()
        self._rr = 0

    def get_actor_by_id(self, actor_id):
        for a in self._actors:
            if getattr(a, "actor_id", id(a)) == actor_id:
                return a
        return None

    def submit(self, fn: Callable, value: Any, affinity_actors: Optional[Sequence[Any]] = None) -> None:
        if affinity_actors:
            actor = next((a for a in affinity_actors if a in self._actors), None)
            if actor is None:
                raise ValueError("none of the affinity actors belongs to this pool")
        else:
            actor = self._actors[self._rr % len(self._actors)]
            self._rr += 1
        self._pending.append((fn, actor, value))

    def has_next(self) -> bool:
        return bool(self._pending)

    def get_next(self, timeout=None):
        if not self._pending:
            raise StopIteration("No more results to get")
        fn, actor, value = self._pending.popleft()
        return fn(actor, value)

    def map(self, fn: Callable, values: Sequence[Any]):
        for v in values:
            self.submit(fn, v)
        while self.has_next():
            yield self.get_next()


ActorPool = _ActorPool  # reference: fllib/utils/actor_pool.py


class ActorManager:
    """``execute_with_actor_pool(func, values, affinity_actors)`` across the ranks of the job.

    ``worker`` is the rank-local :class:`~blades_amd.parallel.worker.Worker`; ``group`` a ``torch.distributed``
    process group (``None`` = default group, or no distribution at all when none is initialised).
    """

    def __init__(self, worker, group=None, max_remote_requests_in_flight_per_actor: int = 1):
        self._worker = worker
        self._group = group
        self._pool = _ActorPool([worker])

    # -- topology -------------------------------------------------------------------------------------------------
    @property
    def world_size(self) -> int:
        return dist.get_world_size(self._group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def rank(self) -> int:
        return dist.get_rank(self._group) if dist.is_available() and dist.is_initialized() else 0

    def actors(self):
        """rank id → handle; only the local worker is addressable from this process."""
        return {self.rank: self._worker}

    def num_actors(self) -> int:
        return self.world_size

    # -- execution ------------------------------------------------------------------------------------------------
    def execute_with_actor_pool(self, func: Callable, values: Sequence[Any],
                                affinity_actors: Optional[Sequence[Optional[Sequence[int]]]] = None,
                                cost: Optional[Sequence[float]] = None) -> List[Any]:
        values = list(values)
        world, rank = self.world_size, self.rank
        owner = assign_to_ranks(len(values), world, affinity_actors or None, cost)
        mine = [(i, v) for i, (v, r) in enumerate(zip(values, owner)) if r == rank]
        for _, v in mine:
            self._pool.submit(lambda a, value: a.apply(func, value), v)
        local, err = [], None
        for i, _ in mine:
            try:
                local.append((i, self._pool.get_next()))
            except Exception as e:  # noqa: BLE001 — keep the collective below matched on every rank
                err = err or e
                local.append((i, None))
        if world == 1:
            if err is not None:
                raise err
            return [r for _, r in local]
        gathered: List[Any] = [None] * world
        dist.all_gather_object(gathered, (local, repr(err) if err is not None else None), group=self._group)
        failures = [f"rank {r}: {e}" for r, (_, e) in enumerate(gathered) if e is not None]
        if failures:
            raise RuntimeError("execute_with_actor_pool failed on " + "; ".join(failures))
        out: List[Any] = [None] * len(values)
        for part, _ in gathered:
            for i, r in part:
                out[i] = r
        return out

    def foreach_actor(self, func: Callable) -> List[Any]:
        """Run ``func(worker)`` on every rank; every rank gets the list of all results (rank order)."""
        res = self._worker.apply(func)
        if self.world_size == 1:
            return [res]
        gathered: List[Any] = [None] * self.world_size
        dist.all_gather_object(gathered, res, group=self._group)
        return gathered
