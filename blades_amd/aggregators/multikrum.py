"""Multi-Krum (reference ``fllib/aggregators/multikrum.py``)."""
import torch

from blades_amd import ops
from .base import Aggregator, global_gram, pairwise_sqdist_from_gram


class Multikrum(Aggregator):
    r"""Selects the ``k`` clients with the lowest Krum score and averages them.

    Score of client *i* = sum of its ``n - f - 2`` smallest neighbour terms.
    The reference squares the already squared distances a second time
    (``multikrum.py:19-21`` on values from ``:87``), i.e. it sums **4th powers**
    of Euclidean distances; ``reference_compat=True`` (default) reproduces
    that, ``False`` uses the textbook squared distances.

    Pairwise distances come from one ``U U^T`` MFMA GEMM (``ops.gram``) instead
    of the reference's O(K^2) Python double loop.
    """

    def __init__(self, num_byzantine: int, k: int = 1, reference_compat: bool = True):
        super().__init__()
        self.f = num_byzantine
        self.m = k
        self.reference_compat = reference_compat

    def aggregate(self, U, ctx):
        n = U.shape[0]
        f, m = self.f, self.m
        if n < 1:
            raise ValueError(f"Number of workers should be positive integer. Got {n}.")
        if m < 1 or m > n:
            raise ValueError(f"Number of workers for aggregation should be >=1 and <= {n}. Got {m}.")
        if 2 * f + 2 > n:
            raise ValueError(f"Too many Byzantine workers: 2 * {f} + 2 >= {n}.")
        d2 = pairwise_sqdist_from_gram(global_gram(U, ctx))
        terms = d2 * d2 if self.reference_compat else d2
        terms = terms + torch.diag(torch.full((n,), float("inf"), device=U.device))
        nn = n - f - 2
        if nn > 0:
            scores = torch.topk(terms, nn, dim=1, largest=False).values.sum(dim=1)
        else:
            scores = torch.zeros(n, device=U.device)
        # stable sort = Python's sorted() tie behaviour (lowest index first)
        order = torch.sort(scores, stable=True).indices[:m]
        mask = torch.zeros(n, device=U.device)
        mask[order] = 1.0
        self.last_info = {"selected": order}
        return ops.col_mean(U, mask)
