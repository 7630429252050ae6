"""A group of helper worker *processes* on this node with the call surface of the reference's vendored Ray-Train
``WorkerGroup`` (``fllib/_internal/worker_group.py``: ``execute`` / ``execute_async`` / ``execute_single`` /
``execute_single_async`` / ``add_workers`` / ``remove_workers`` / ``shutdown`` / ``__len__``).

The GPU ranks of a job are started by ``blades_amd.parallel.launcher`` and talk over RCCL; this class is for the
other kind of worker the reference used Ray actors for — host-side helpers (data preparation, partitioning,
evaluation of saved checkpoints on the CPU) that should run beside the training process.  Each worker is one
``spawn``-ed Python process that executes the callables it is sent, in order; callables and arguments travel by
pickle (cloudpickle when available, so lambdas and closures work).  Workers never touch the GPU unless the callable
does, and never re-exec.
"""
import concurrent.futures as cf
import multiprocessing as mp
import os
from typing import Any, Callable, Dict, List, Optional

try:  # closures / lambdas
    import cloudpickle as _pickle
except Exception:  # pragma: no cover
    import pickle as _pickle


def _call(payload: bytes):
    func, args, kwargs = _pickle.loads(payload)
    return func(*args, **kwargs)


def _init_worker(env: Dict[str, str], index: int):
    os.environ.update(env)
    os.environ["BLADES_WORKER_INDEX"] = str(index)


class WorkerMetadata:
    def __init__(self, index: int, pid: Optional[int]):
        self.index, self.pid, self.node_ip, self.hostname = index, pid, "127.0.0.1", os.uname().nodename


class _Worker:
    def __init__(self, index: int, env: Dict[str, str]):
        self.index = index
        self.pool = cf.ProcessPoolExecutor(max_workers=1, mp_context=mp.get_context("spawn"),
                                           initializer=_init_worker, initargs=(env, index))
        self.metadata = WorkerMetadata(index, None)

    def submit(self, func: Callable, *args, **kwargs) -> cf.Future:
        return self.pool.submit(_call, _pickle.dumps((func, args, kwargs)))


class ProcessWorkerGroup:
    def __init__(self, num_workers: int = 1, env: Optional[Dict[str, str]] = None):
        if num_workers <= 0:
            raise ValueError(f"The provided `num_workers` must be greater than 0. Received num_workers={num_workers}.")
        self._env = dict(env or {})
        self._next_index = 0
        self.workers: List[_Worker] = []
        self.add_workers(num_workers)

    # -- membership -----------------------------------------------------------------------------------------------
    def add_workers(self, num_workers: int) -> None:
        for _ in range(num_workers):
            self.workers.append(_Worker(self._next_index, self._env))
            self._next_index += 1

    def remove_workers(self, worker_indexes: List[int]) -> None:
        """Stops and drops the workers at these positions of ``self.workers``."""
        drop = set(worker_indexes)
        keep = []
        for i, w in enumerate(self.workers):
            if i in drop:
                w.pool.shutdown(wait=True, cancel_futures=True)
            else:
                keep.append(w)
        self.workers = keep

    def shutdown(self, patience_s: float = 5) -> None:
        for w in self.workers:
            w.pool.shutdown(wait=patience_s > 0, cancel_futures=True)
        self.workers = []

    def __len__(self) -> int:
        return len(self.workers)

    # -- execution ------------------------------------------------------------------------------------------------
    def execute_async(self, func: Callable, *args, **kwargs) -> List[cf.Future]:
        if not self.workers:
            raise RuntimeError("There are no active workers. This worker group has most likely been shut down.")
        return [w.submit(func, *args, **kwargs) for w in self.workers]

    def execute(self, func: Callable, *args, **kwargs) -> List[Any]:
        return [f.result() for f in self.execute_async(func, *args, **kwargs)]

    def execute_single_async(self, worker_index: int, func: Callable, *args, **kwargs) -> cf.Future:
        if not 0 <= worker_index < len(self.workers):
            raise ValueError(f"The provided worker_index {worker_index} is not valid for {len(self.workers)} workers.")
        return self.workers[worker_index].submit(func, *args, **kwargs)

    def execute_single(self, worker_index: int, func: Callable, *args, **kwargs) -> Any:
        return self.execute_single_async(worker_index, func, *args, **kwargs).result()
