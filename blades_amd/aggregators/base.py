"""Aggregator base class.

Reference aggregators are callables ``__call__(List[Tensor[D]]) -> Tensor[D]``
(``fllib/aggregators/__init__.py``).  Here the primary entry point is
``aggregate(U, ctx)`` on the resident ``[K, D_local]`` update matrix (a column
shard when ``ctx.world_size > 1``); ``__call__`` keeps the list-of-vectors
signature by stacking and delegating, so reference-style user code works.
"""
from typing import List, Optional, Sequence, Union

import torch

from blades_amd.parallel.context import ShardContext, local_context


class Aggregator:
    #: extra info of the last call (e.g. which clients were kept) for logging
    last_info: dict
    #: ``aggregate`` rescales rows of ``U`` in place (norm clipping).  The engine's resident update matrix is meant to
    #: be consumed that way; ``__call__`` protects a caller-owned tensor by working on a copy, as the reference does
    #: by stacking its input list first (fllib/aggregators/clippedclustering.py:28, signguard.py:40).
    mutates_input: bool = False

    def __init__(self):
        self.last_info = {}

    def aggregate(self, U: torch.Tensor, ctx: ShardContext) -> torch.Tensor:
        raise NotImplementedError

    def __call__(self, inputs: Union[Sequence[torch.Tensor], torch.Tensor],
                 ctx: Optional[ShardContext] = None) -> torch.Tensor:
        if isinstance(inputs, torch.Tensor):
            U = inputs
        else:
            U = torch.stack([t.reshape(-1) for t in inputs], dim=0)
        if U.dtype != torch.float32:
            U = U.float()
        if U.dim() != 2 or U.stride(1) != 1:  # a row-strided view of a padded buffer is fine as it is
            U = U.contiguous()
        if ctx is None:
            ctx = local_context(U.shape[1])
        if self.mutates_input and U is inputs:
            U = U.clone()
        return self.aggregate(U, ctx)

    # -- checkpointing of cross-round state ------------------
    def state_dict(self) -> dict:
        return {}

    def load_state_dict(self, state: dict) -> None:
        pass

    def __repr__(self):
        return f"{type(self).__name__}()"


def global_row_stats(U, ctx):
    """(||row||^2 [K], sign counts [K,3]) reduced over all column shards."""
    from blades_amd import ops

    sq, cnt = ops.row_stats(U, ctx.d_valid if ctx.distributed else None)
    if ctx.distributed:
        pack = torch.cat([sq.double().unsqueeze(1), cnt.double()], dim=1)
        ctx.all_reduce(pack)
        sq, cnt = pack[:, 0].float(), pack[:, 1:].round().to(torch.int64)
    return sq, cnt


def global_sqnorm(U, ctx):
    from blades_amd import ops

    sq = ops.row_sqnorm(U)
    return ctx.all_reduce(sq) if ctx.distributed else sq


def global_gram(U, ctx):
    from blades_amd import ops

    G = ops.gram(U)
    return ctx.all_reduce(G) if ctx.distributed else G


def pairwise_sqdist_from_gram(G: torch.Tensor) -> torch.Tensor:
    """``d2[i,j] = g_ii + g_jj - 2 g_ij`` clamped at 0."""
    diag = G.diagonal()
    d2 = diag.unsqueeze(0) + diag.unsqueeze(1) - 2.0 * G
    d2 = d2.clamp_min_(0.0)
    d2.fill_diagonal_(0.0)
    return d2
