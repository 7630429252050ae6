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
    dis = torch.nan_to_num((1.0 - G / denom).double(), nan=2.0, posinf=2.0, neginf=0.0)
    dis = 0.5 * (dis + dis.T)
    dis.fill_diagonal_(0.0)
    return dis.cpu().numpy()  # the one device->host copy of the aggregator: K x K doubles for the host clustering


class NormHistory:
    """Every client norm seen so far (reference ``l2norm_his``, ``clippedclustering.py:33-35``) as a device buffer
    that doubles its capacity: appending and taking the median never leave the device."""

    def __init__(self):
        self.buf = None
        self.n = 0

    def extend(self, norms: torch.Tensor):
        k = norms.numel()
        if self.buf is None or self.buf.device != norms.device or self.n + k > self.buf.numel():
            cap = max(2 * (self.n + k), 1024)
            new = torch.empty(cap, dtype=torch.float32, device=norms.device)
            if self.n:
                new[: self.n] = self.buf[: self.n].to(norms.device)
            self.buf = new
        self.buf[self.n: self.n + k] = norms.float()
        self.n += k

    def median(self) -> torch.Tensor:
        """``numpy.median`` semantics (mean of the two middle values for an even count), 0-d device tensor."""
        v = self.buf[: self.n]
        hi = torch.kthvalue(v, self.n // 2 + 1).values
        return hi if self.n % 2 else 0.5 * (hi + torch.kthvalue(v, self.n // 2).values)

    def tolist(self):
        return [] if self.buf is None else self.buf[: self.n].tolist()

    def __len__(self):
        return self.n


class Clippedclustering(Aggregator):
    """Clip to the running median norm, agglomerative 2-clustering on cosine
    distance, keep the majority cluster (optionally ∩ SignGuard's sign
    clustering), then Mean/Median.

    ``l2norm_his`` grows by K every round like the reference (``:33-35``) and stays on the device (norms, running
    median and clip factors are computed without a host sync).  The cosine matrix comes from one MFMA Gram GEMM on the
    clipped rows and makes the aggregator's single device->host copy; the clustering itself is sequential
    (K - 1 dependent merges) and runs in the native host library (blades_amd/ops/csrc/host.cpp, nearest-neighbour
    chain, ~10 ms at K = 1000, 0.3 ms at K = 60).
    """

    mutates_input = True

    def __init__(self, agg="mean", signguard=False, max_tau=1e5, linkage="average") -> None:
        super().__init__()
        assert linkage in ["average", "single"]
        self.tau = max_tau
        self.signguard = signguard
        self.linkage = linkage
        self.l2norm_his = NormHistory()
        self.agg = _make_inner(agg)

    def aggregate(self, U, ctx):
        num = U.shape[0]
        if self.signguard:
            sq, cnt = global_row_stats(U, ctx)
        else:
            from .base import global_sqnorm

            sq, cnt = global_sqnorm(U, ctx), None
        norms = sq.sqrt()
        self.l2norm_his.extend(norms)
        threshold = self.l2norm_his.median().clamp(max=self.tau).to(norms.dtype)
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
        return {"l2norm_his": self.l2norm_his.tolist()}

    def load_state_dict(self, state):
        self.l2norm_his = NormHistory()
        his = state.get("l2norm_his", [])
        if len(his):
            self.l2norm_his.extend(torch.tensor(his, dtype=torch.float32))
