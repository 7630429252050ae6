"""Clipped clustering (reference ``fllib/aggregators/clippedclustering.py``)."""
import numpy as np
import torch

from blades_amd import ops
from blades_amd.ops import host
from .base import Aggregator, global_gram, global_row_stats
from .signguard import _majority_label, _make_inner, masked_inner, sign_features


def cosine_distance_from_gram(G: torch.Tensor) -> np.ndarray:
    """``1 - cos`` matrix with the reference's clean-ups (``:50-52``):
    ``-inf -> 0``, ``inf -> 2``, ``nan -> 2``; zero diagonal."""
    nrm = G.diagonal().clamp_min(0).sqrt()
    # torch.cosine_similarity clamps each norm at eps = 1e-8
    denom = nrm.clamp_min(1e-8).unsqueeze(0) * nrm.clamp_min(1e-8).unsqueeze(1)
    dis = (1.0 - G / denom).double().cpu().numpy()
    dis[dis == -np.inf] = 0
    dis[dis == np.inf] = 2
    dis[np.isnan(dis)] = 2
    dis = 0.5 * (dis + dis.T)
    np.fill_diagonal(dis, 0.0)
    return dis


class Clippedclustering(Aggregator):
    """Clip to the running median norm, agglomerative 2-clustering on cosine
    distance, keep the majority cluster (optionally ∩ SignGuard's sign
    clustering), then Mean/Median.

    ``l2norm_his`` grows by K every round like the reference (``:33-35``).  The
    cosine matrix comes from one MFMA Gram GEMM on the clipped rows; the
    clustering runs in the native host library.
    """

    def __init__(self, agg="mean", signguard=False, max_tau=1e5, linkage="average") -> None:
        super().__init__()
        assert linkage in ["average", "single"]
        self.tau = max_tau
        self.signguard = signguard
        self.linkage = linkage
        self.l2norm_his = []
        self.agg = _make_inner(agg)

    def aggregate(self, U, ctx):
        num = U.shape[0]
        if self.signguard:
            sq, cnt = global_row_stats(U, ctx)
        else:
            from .base import global_sqnorm

            sq, cnt = global_sqnorm(U, ctx), None
        norms = sq.sqrt()
        self.l2norm_his.extend(norms.tolist())
        threshold = min(float(np.median(self.l2norm_his)), self.tau)
        scale = torch.where(norms > threshold, threshold / norms, torch.ones_like(norms))
        ops.row_scale_(U, scale)

        dis = cosine_distance_from_gram(global_gram(U, ctx))
        labels = host.agglomerative2(dis, self.linkage)
        flag = _majority_label(labels)
        sel = torch.from_numpy(labels == flag).to(U.device)
        if self.signguard:
            klabels, _ = host.kmeans(sign_features(cnt, ctx.d_total), k=2, seed=0)
            kflag = _majority_label(klabels)
            sel = sel & torch.from_numpy(klabels == kflag).to(U.device)
        self.last_info = {"num_selected": sel.sum()}
        return masked_inner(U, sel, self.agg)

    def state_dict(self):
        return {"l2norm_his": list(self.l2norm_his)}

    def load_state_dict(self, state):
        self.l2norm_his = list(state.get("l2norm_his", []))
