"""Collective communication helpers over ``torch.distributed`` (RCCL on GPUs, gloo on CPU).

The reference has an *experimental*, never-wired NCCL layer (``fllib/communication``):
``Communicator`` actors that ``gather`` fixed-size packages into a pre-allocated NaN-filled
``_BufferList`` on ``SERVER_RANK`` and ``broadcast`` tensors from it, rendezvous on a hard-coded
IP, tensors handed over through a CUDA-IPC Ray queue; ``CommunicationSet.broadcast`` returns
before reaching the collective.  This module is the working equivalent for the one-process-per-GPU
runtime — no IPC queue is needed because the tensors already live in the rank that owns them:

* ``gather_async`` / ``wait`` / ``local_round``  — packages of the ranks collected on the server
  rank into the buffer list (async, like the reference's ``async_op=True``);
* ``broadcast_tensor``                            — from any root;
* ``all_to_all_rows`` / ``all_gather_rows``       — the two layouts the engine uses for the update
  matrix (row shard -> column shard, and "gather the full matrix", BASELINE config #5).

xGMI is point-to-point (7 links per GPU): all-to-all and all-gather use every link at once,
whereas a gather to one root is bound by the root's links — it is kept for API parity, the
engine itself never funnels updates through one rank.
"""
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from blades_amd.constants import SERVER_RANK
from .runtime import Runtime, get_runtime


class _BufferList:
    """``buffer_len`` pre-allocated package slots, NaN = "empty" (reference ``communicator.py:18-115``)."""

    def __init__(self, buffer_len: int, pkg_size: Sequence[int], device, dtype=torch.float32):
        self.pkg_size = tuple(pkg_size)
        self.buffers: List[torch.Tensor] = [torch.full(self.pkg_size, float("nan"), device=device, dtype=dtype)
                                            for _ in range(buffer_len)]
        self._next = 0

    def __len__(self):
        return len(self.buffers)

    def __getitem__(self, i):
        return self.buffers[i]

    def take(self, n: int) -> List[torch.Tensor]:
        """Next ``n`` slots (ring order)."""
        out = []
        for _ in range(n):
            out.append(self.buffers[self._next])
            self._next = (self._next + 1) % len(self.buffers)
        return out

    def filled(self) -> List[torch.Tensor]:
        return [b for b in self.buffers if not bool(torch.isnan(b).any())]

    def reset(self):
        for b in self.buffers:
            b.fill_(float("nan"))
        self._next = 0

    def stack(self) -> torch.Tensor:
        return torch.stack(self.filled(), dim=0) if self.filled() else torch.empty((0,) + self.pkg_size)


def all_gather_uneven_rows(U_row: torch.Tensor, rows_per_rank: Sequence[int], group=None) -> torch.Tensor:
    """all-gather of row blocks whose heights differ per rank (collectives want equal shapes: pad to
    the tallest block, gather, cut the padding)."""
    kmax = max(rows_per_rank)
    mine = U_row
    if U_row.shape[0] < kmax:
        mine = torch.zeros(kmax, U_row.shape[1], device=U_row.device, dtype=U_row.dtype)
        mine[: U_row.shape[0]] = U_row
    buf = torch.empty(len(rows_per_rank) * kmax, U_row.shape[1], device=U_row.device, dtype=U_row.dtype)
    dist.all_gather_into_tensor(buf, mine.contiguous(), group=group)
    return torch.cat([buf[i * kmax: i * kmax + k] for i, k in enumerate(rows_per_rank)], dim=0)


class Communicator:
    def __init__(self, pkg_size: Sequence[int] = (1,), buffer_len: Optional[int] = None,
                 runtime: Optional[Runtime] = None, group=None, dtype=torch.float32):
        self.rt = runtime or get_runtime()
        self.group = group
        self.pkg_size = tuple(pkg_size)
        self.dtype = dtype
        self.is_server = self.rt.rank == SERVER_RANK
        n = buffer_len if buffer_len is not None else max(self.rt.world_size - 1, 1)
        self.buffer_list = _BufferList(n, self.pkg_size, self.rt.device, dtype) if self.is_server else None
        self._pending = []
        self._dummy = torch.zeros(self.pkg_size, device=self.rt.device, dtype=dtype)

    # -- reference surface ---------------------------------------------------------------------
    def setup(self):
        return self

    @property
    def world_size(self) -> int:
        return self.rt.world_size

    def gather_async(self, tensor: Optional[torch.Tensor] = None):
        """One package from every non-server rank into the server's next buffer slots."""
        if self.rt.world_size == 1:
            if tensor is not None and self.buffer_list is not None:
                self.buffer_list.take(1)[0].copy_(tensor)
            return None
        if self.is_server:
            slots = self.buffer_list.take(self.rt.world_size - 1)
            work = dist.gather(self._dummy, [self._dummy.clone()] + slots, dst=SERVER_RANK, group=self.group,
                               async_op=True)
        else:
            if tensor is None or tuple(tensor.shape) != self.pkg_size:
                raise ValueError(f"package must have shape {self.pkg_size}")
            work = dist.gather(tensor.to(self.rt.device, self.dtype).contiguous(), dst=SERVER_RANK, group=self.group,
                               async_op=True)
        self._pending.append(work)
        return work

    def wait(self):
        for w in self._pending:
            if w is not None:
                w.wait()
        self._pending = []

    def local_round(self, tensors: Sequence[Optional[torch.Tensor]]):
        """``len(tensors)`` gathers, then wait (reference ``local_round``)."""
        for t in tensors:
            self.gather_async(t)
        self.wait()
        return self.buffer_list.filled() if self.is_server else None

    def broadcast_tensor(self, tensor: torch.Tensor, root: int = SERVER_RANK) -> torch.Tensor:
        if self.rt.world_size > 1:
            dist.broadcast(tensor, src=root, group=self.group)
        return tensor

    # -- update-matrix layouts -----------------------------------------------------------------
    def all_to_all_rows(self, U_row: torch.Tensor, rows_per_rank: Sequence[int]) -> torch.Tensor:
        """``[K_r, G*Dl]`` row shard -> ``[K, Dl]`` column shard."""
        G = self.rt.world_size
        if G == 1:
            return U_row
        Kr, Dpad = U_row.shape
        Dl = Dpad // G
        send = U_row.view(Kr, G, Dl).transpose(0, 1).contiguous()
        out = torch.empty(sum(rows_per_rank), Dl, device=U_row.device, dtype=U_row.dtype)
        dist.all_to_all_single(out.view(-1), send.view(-1), [k * Dl for k in rows_per_rank], [Kr * Dl] * G,
                               group=self.group)
        return out

    def all_gather_rows(self, U_row: torch.Tensor, rows_per_rank: Sequence[int]) -> torch.Tensor:
        if self.rt.world_size == 1:
            return U_row
        return all_gather_uneven_rows(U_row, rows_per_rank, self.group)


class CommunicationSet:
    """All ranks' communicators seen as one object (reference ``communicator_set.py``).  In the
    one-process-per-GPU runtime every rank builds its own member; ``broadcast`` really broadcasts."""

    def __init__(self, pkg_size: Sequence[int] = (1,), buffer_len: Optional[int] = None, runtime=None):
        self.communicator = Communicator(pkg_size, buffer_len, runtime)

    def broadcast(self, tensor: torch.Tensor, root: int = SERVER_RANK) -> torch.Tensor:
        return self.communicator.broadcast_tensor(tensor, root)

    def gather(self, tensor: Optional[torch.Tensor]):
        return self.communicator.local_round([tensor])
