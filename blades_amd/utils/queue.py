"""Inter-process queue that moves tensors without copying them through a pipe.

Reference: ``fllib/utils/queue.py:8-30`` — a Ray queue actor whose ``put`` replaces a CUDA tensor by its IPC
handle (``reduce_tensor``) and whose ``get`` rebuilds it.  Here the processes are the per-GPU ranks (or helper
processes such as a data loader) of one node, so the queue is a ``torch.multiprocessing`` queue: its pickler
already ships device tensors as HIP IPC handles (dmabuf; needs ``HSA_ENABLE_IPC_MODE_LEGACY=0``, exported by
``blades_amd.parallel.launcher``) and host tensors as shared-memory segments.  The usual IPC rule applies: the
producer must keep a device tensor alive until the consumer has opened it.
"""
import queue as _pyqueue
from typing import Any, Optional

import torch.multiprocessing as mp

Empty = _pyqueue.Empty
Full = _pyqueue.Full


class Queue:
    def __init__(self, maxsize: int = 0, ctx: Optional[str] = "spawn") -> None:
        self.maxsize = maxsize
        self._q = mp.get_context(ctx).Queue(maxsize)

    def put(self, item: Any, block: bool = True, timeout: Optional[float] = None) -> None:
        self._q.put(item, block, timeout)

    def get(self, block: bool = True, timeout: Optional[float] = None) -> Any:
        return self._q.get(block, timeout)

    def put_nowait(self, item: Any) -> None:
        self.put(item, block=False)

    def get_nowait(self) -> Any:
        return self.get(block=False)

    def put_nowait_batch(self, items) -> None:
        for it in items:
            self.put_nowait(it)

    def get_nowait_batch(self, num_items: int):
        return [self.get_nowait() for _ in range(num_items)]

    def qsize(self) -> int:
        return self._q.qsize()

    size = qsize
    __len__ = qsize

    def empty(self) -> bool:
        return self._q.empty()

    def full(self) -> bool:
        return self._q.full()

    def shutdown(self) -> None:
        self._q.close()
        self._q.join_thread()
