"""Shard context: how this process' slice of the update matrix relates to the
whole, plus the (tiny) collectives aggregators and adversaries need.

Design: clients are sharded over ranks for training
(row shards ``[K_r, D]``), one all-to-all turns them into column shards
``[K, D/G]``; every coordinate-wise statistic is then local, every
norm/dot/Gram statistic is a local partial followed by a small all-reduce over
RCCL (xGMI).  With ``world_size == 1`` every collective is the identity, so
the same aggregator code serves the single-GPU, multi-GPU and CPU-test paths.
"""
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.distributed as dist

from blades_amd.constants import COL_ALIGN


def padded_width(d_total: int, world_size: int) -> int:
    """Per-rank column width: ceil(D / G) rounded up to ``COL_ALIGN``."""
    per = -(-d_total // world_size)
    return -(-per // COL_ALIGN) * COL_ALIGN


@dataclass
class ShardContext:
    d_total: int = 0  # true number of parameters D
    world_size: int = 1
    rank: int = 0
    group: Optional[object] = None
    force: bool = False  # take the distributed paths with one rank too (``Runtime.force_collectives``)
    # statistics, filled by users (bytes moved per round etc.)
    stats: dict = field(default_factory=dict)

    @property
    def d_local(self) -> int:
        return padded_width(self.d_total, self.world_size) if self.distributed else self.d_total

    @property
    def col_offset(self) -> int:
        return self.rank * self.d_local if self.distributed else 0

    @property
    def d_valid(self) -> int:
        """Number of non-padding columns in this rank's slice."""
        return max(0, min(self.d_local, self.d_total - self.col_offset))

    @property
    def distributed(self) -> bool:
        return self.world_size > 1 or self.force

    # -- collectives -----------------------------------------------------
    def all_reduce(self, t: torch.Tensor, op=dist.ReduceOp.SUM) -> torch.Tensor:
        if self.distributed:
            dist.all_reduce(t, op=op, group=self.group)
            self.stats["allreduce_bytes"] = self.stats.get("allreduce_bytes", 0) + t.numel() * t.element_size()
        return t

    def all_gather_cols(self, local: torch.Tensor) -> torch.Tensor:
        """``[.., d_local]`` slices -> ``[.., D]`` (padding stripped)."""
        if not self.distributed:
            return local
        parts = [torch.empty_like(local) for _ in range(self.world_size)]
        dist.all_gather(parts, local.contiguous(), group=self.group)
        full = torch.cat(parts, dim=-1)
        return full[..., : self.d_total]

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.distributed:
            dist.broadcast(t, src=src, group=self.group)
        return t

    def barrier(self):
        if self.distributed:
            dist.barrier(group=self.group)


def local_context(d_total: int) -> ShardContext:
    return ShardContext(d_total=d_total)


def context_from_env(d_total: int) -> ShardContext:
    """Context for the default process group (if initialised)."""
    if dist.is_available() and dist.is_initialized():
        return ShardContext(d_total=d_total, world_size=dist.get_world_size(), rank=dist.get_rank())
    return ShardContext(d_total=d_total)
