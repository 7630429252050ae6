"""SignGuard (reference ``fllib/aggregators/signguard.py``)."""
import numpy as np
import torch

from blades_amd import ops
from blades_amd.ops import host
from .base import Aggregator, global_row_stats


def _make_inner(agg: str):
    if agg not in ("mean", "median"):
        raise NotImplementedError(f"{agg} is not supported yet.")
    return agg


def _majority_label(labels: np.ndarray) -> int:
    """``flag = 1 if sum(labels) > num // 2 else 0`` (``signguard.py:61``)."""
    return 1 if int(labels.sum()) > len(labels) // 2 else 0


def masked_inner(U, mask, agg: str):
    return ops.col_mean(U, mask) if agg == "mean" else ops.col_median(U, mask)


def sign_features(cnt: torch.Tensor, d_total: int) -> np.ndarray:
    return (cnt.double() / float(d_total)).cpu().numpy()


class Signguard(Aggregator):
    """Norm filter ∩ sign-statistics KMeans majority, then Mean/Median.

    One fused pass gives every row's squared norm and (>0, <0, ==0) counts
    (``ops.row_stats``; partials are all-reduced when column-sharded); rows
    above the median norm ``M`` are scaled to ``M`` in place (sign counts are
    invariant to positive scaling, so they are taken in the same pass); the
    ``[K,3]`` features go to the native 2-means (``ops.host.kmeans``).
    """

    mutates_input = True

    def __init__(self, agg="mean", max_tau=1e5, linkage="average") -> None:
        super().__init__()
        assert linkage in ["average", "single"]
        self.tau = max_tau
        self.linkage = linkage
        self.l2norm_his = []
        self.agg = _make_inner(agg)

    def aggregate(self, U, ctx):
        num = U.shape[0]
        sq, cnt = global_row_stats(U, ctx)
        norms = sq.sqrt()
        M = norms.median() if num % 2 == 1 else 0.5 * (
            torch.kthvalue(norms, num // 2).values + torch.kthvalue(norms, num // 2 + 1).values)
        scale = torch.where(norms > M, M / norms, torch.ones_like(norms))
        ops.row_scale_(U, scale)
        norms = torch.minimum(norms, M)
        S1 = (norms >= 0.1 * M) & (norms <= 3.0 * M)

        labels, _ = host.kmeans(sign_features(cnt, ctx.d_total), k=2, seed=0)
        flag = _majority_label(labels)
        S2 = torch.from_numpy(labels == flag).to(U.device)
        inter = S1 & S2
        self.last_info = {"num_selected": inter.sum()}
        return masked_inner(U, inter, self.agg)
