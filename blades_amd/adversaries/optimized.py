"""Adaptive (Fang), MinMax, SignGuard-attack, ClippedClustering-attack."""
import math
from typing import Dict

import numpy as np
import torch

from blades_amd import ops
from blades_amd.aggregators.base import global_gram, pairwise_sqdist_from_gram
from blades_amd.aggregators.clippedclustering import cosine_distance_from_gram
from blades_amd.ops import host
from .adversary import Adversary
from .simple import _is_signguard, half_negation_sign


def _seed(cfg) -> int:
    return int(getattr(cfg, "seed", 0) or 0)


class AdaptiveAdversary(Adversary):
    """Fang et al. attack on Median/TrimmedMean (``adaptive_adversary.py:22-67``).

    With ``s = sign(mean)``, ``mx/mn`` the benign column max/min, ``b = 2`` and
    ``u ~ U(0,1)`` per coordinate:

    ====================  ==========================================
    s = -1, mx > 0        ``mx * (1 + u (b-1))``
    s = -1, mx < 0        ``mx * (1 + u (1/b - 1))``
    s = +1, mn > 0        ``mn * (1/b + u (1 - 1/b))``
    s = +1, mn < 0        ``mn * (b + u (1 - b))``
    s = 0                 ``mean`` (= 0)
    otherwise (mx|mn = 0)  0
    ====================  ==========================================
    """

    def forge(self, algorithm, U, ctx):
        mask = algorithm.benign_mask
        mean = ops.col_mean(U, mask)
        mn, mx = ops.col_min_max(U, mask)
        # one fused kernel: Philox uniform per global coordinate + the five sign / extremum cases
        out = ops.adaptive_forge(mean, mn, mx, _seed(self.global_config), 3000 + self._round, ctx.col_offset, 2.0)
        ops.set_rows_(U, algorithm.malicious_rows, out)
        return out


class MinMaxAdversary(Adversary):
    """Shejwalkar & Houmansadr "min-max": ``mean - gamma * std`` with the largest
    ``gamma`` in [0,5] (bisection to 0.01) such that the forged vector's max
    distance to a benign update stays below the max benign pairwise distance
    (``minmax_adversary.py:37-63``).

    The bisection needs no pass over D: with ``a_i = ||mean - x_i||^2``,
    ``b_i = (mean - x_i) . dev`` and ``c = ||dev||^2`` the distance is
    ``a_i - 2 gamma b_i + gamma^2 c``.  So the whole attack is one fused
    mean/std pass, one Gram GEMM, two row reductions, and scalar math that
    stays on the device.
    """

    def __init__(self, clients, global_config: Dict = None):
        super().__init__(clients, global_config)
        self.threshold = 3.0
        self.threshold_diff = 1e-4
        self.num_byzantine = None
        self.negative_indices = None

    def forge(self, algorithm, U, ctx):
        mask = algorithm.benign_mask
        bm = mask > 0
        mean, dev = ops.col_mean_std(U, mask)
        if _is_signguard(algorithm):
            dev = dev * half_negation_sign(ctx, U.device)
        G = global_gram(U, ctx)
        d2 = pairwise_sqdist_from_gram(G)[bm][:, bm]
        threshold = d2.max().sqrt()
        a = ops.sqdist_to(U, mean)
        b = ops.row_dot(U, dev)
        scal = torch.stack([(mean * dev).sum(), (dev * dev).sum()])
        if ctx.distributed:
            ctx.all_reduce(a), ctx.all_reduce(b), ctx.all_reduce(scal)
        a, b = a[bm], scal[0] - b[bm]
        c = scal[1]
        low = torch.zeros((), device=U.device)
        high = torch.full((), 5.0, device=U.device)
        mid = (low + high) / 2
        lo_f, hi_f = 0.0, 5.0
        if U.is_cuda and ops.smallk_max(U.device) > 0 and a.numel() > 0:
            # the nine halving steps in one launch (ops/csrc/smallk.hip) instead of ten tiny kernels each
            mid = ops.minmax_bisect(a.float(), b.float(), c, threshold, 0.0, 5.0, 0.01)
            hi_f = lo_f
        while abs(hi_f - lo_f) > 0.01:  # fixed trip count: the interval halves every time
            mid = (low + high) / 2
            loss = (a - 2 * mid * b + mid * mid * c).clamp_min(0).max().sqrt()
            ok = loss < threshold
            low = torch.where(ok, mid, low)
            high = torch.where(ok, high, mid)
            hi_f = lo_f + (hi_f - lo_f) / 2
        out = mean - mid * dev
        ops.set_rows_(U, algorithm.malicious_rows, out)
        self.num_byzantine = int(algorithm.malicious_rows.numel())
        return out


class SignGuardAdversary(Adversary):
    """Noise with the benign mean's sign statistics (``signguard_adversary.py:39-67``): exactly ``pos`` coordinates
    get ``+U(0,1)``, exactly ``neg`` get ``-U(0,1)``, the rest 0, at uniformly random positions.

    The reference concatenates the three blocks and applies a CPU ``randperm(D)``.  Here every coordinate gets a
    unique random sort key (31 Philox bits : global index); the ``pos`` smallest keys are the positive positions, the
    next ``neg`` the negative ones — the same distribution with the same exact counts, as a device sort plus two
    one-pass kernels, and invariant to how D is sharded over ranks (the two thresholds are found by a bisection on
    the key value with one scalar all-reduce per step when the matrix is column-sharded)."""

    def forge(self, algorithm, U, ctx):
        mean = ops.col_mean(U, algorithm.benign_mask)
        _, cnt = ops.row_stats(mean.unsqueeze(0), ctx.d_valid if ctx.distributed else None)  # [1, 3]: > 0, < 0, == 0
        cnt = cnt[0, :2].to(torch.int64)
        if ctx.distributed:
            c = cnt.double()
            ctx.all_reduce(c)
            cnt = c.round().to(torch.int64)
        ranks = torch.stack([cnt[0], cnt[0] + cnt[1]])  # keys of global rank pos / pos + neg bound the two classes
        seed = _seed(self.global_config)
        n = U.shape[1]
        keys = ops.perm_keys(n, seed, 4000 + 2 * self._round, ctx.col_offset, ctx.col_offset + ctx.d_valid
                             if ctx.distributed else n, device=U.device)
        srt = torch.sort(keys).values
        big = torch.full((1,), 0x7FFFFFFFFFFFFFFF, dtype=torch.int64, device=U.device)
        if not ctx.distributed:
            thr = torch.cat([srt, big]).index_select(0, ranks.clamp(max=n))
        else:
            # largest t with #(keys < t) <= rank, over all shards: bisection over the 63 key bits
            lo = torch.zeros(2, dtype=torch.int64, device=U.device)
            hi = big.expand(2).clone()
            for _ in range(63):
                mid = lo + ((hi - lo) >> 1) + ((hi - lo) & 1)  # upper middle, no int64 overflow
                c = torch.searchsorted(srt, mid).double()  # keys < mid on this shard
                ctx.all_reduce(c)
                ok = c.round().to(torch.int64) <= ranks
                lo = torch.where(ok, mid, lo)
                hi = torch.where(ok, hi, mid - 1)
            thr = lo
        out = ops.signguard_forge(keys, thr, seed, 4001 + 2 * self._round, ctx.col_offset)
        ops.set_rows_(U, algorithm.malicious_rows, out)
        return out


class AttackclippedclusteringAdversary(Adversary):
    """Chain attack against ClippedClustering
    (``attackclippedclustering_adversary.py:24-97``): malicious client t gets the
    benign direction rotated a further ``theta_cross`` away from the benign mean,
    scaled by 10, so that single-linkage clustering chains the attackers onto the
    benign cluster.

    Every crafted vector is ``alpha_t * mean + beta_t * x_far`` for scalars that
    follow from norms and dot products (Gram GEMM + two row reductions), so only
    the final row writes touch D-sized data.  Two reference quirks are kept:
    ``dis_cross`` is the minimum over **all** pairs (the loop never checks the
    labels, ``:47-53``), and the arg-max index found inside the larger cluster
    is applied to the full benign set (``:73-76``).
    """

    def __init__(self, clients, global_config: Dict = None, linkage="single"):
        super().__init__(clients, global_config)
        self.linkage = linkage

    def forge(self, algorithm, U, ctx):
        bm = algorithm.benign_mask > 0
        # index-based logic below runs in *client* order so that it does not depend on how clients are
        # laid out over ranks (rows are rank-major)
        rc = algorithm.row_clients.to(U.device)
        bidx = torch.nonzero(bm).flatten()
        bidx = bidx[torch.argsort(rc[bidx])]
        mean = ops.col_mean(U, algorithm.benign_mask)
        G = global_gram(U, ctx)[bidx][:, bidx]
        dots = ops.row_dot(U, mean)
        mm = (mean * mean).sum().unsqueeze(0)
        if ctx.distributed:
            ctx.all_reduce(dots), ctx.all_reduce(mm)
        # two device->host copies in all: the scalars below and the K x K distance matrix for the host clustering
        num = int(bidx.numel())
        packed = torch.cat([dots[bidx].double(), G.diagonal().clamp_min(0).sqrt().double(), mm.sqrt().double()])
        packed = packed.cpu().numpy()
        dots, norms, mean_norm = packed[:num], packed[num: 2 * num], float(packed[2 * num])
        dis = cosine_distance_from_gram(G)
        labels = host.agglomerative2(dis, "single")
        off = dis + np.eye(num) * 10.0
        dis_cross = float(off.min()) if num > 1 else 0.0
        theta_cross = math.acos(max(-1.0, min(1.0, 1 - dis_cross))) - 0.1
        flag = 1 if labels.sum() > num // 2 else 0
        larger = np.nonzero(labels == flag)[0]
        cos2mean = dots / np.maximum(norms * mean_norm, 1e-16)
        idx_max = int(np.argmax(1 - cos2mean[larger]))
        theta = math.acos(max(-1.0, min(1.0, cos2mean[larger][idx_max])))
        far = idx_max  # (sic) index into the full benign set, see docstring
        # mal = alpha * mean + beta * x_far
        alpha, beta = 0.0, 1.0 / max(norms[far], 1e-16)
        x_far = U[bidx[far]].clone()
        mrows = algorithm.malicious_rows
        rows = mrows[torch.argsort(algorithm.malicious_row_clients.to(mrows.device))]
        coef = []
        for _ in range(int(rows.numel())):  # scalar recurrence on the host, no device work
            if theta + theta_cross >= math.pi:
                alpha, beta = -10.0, 0.0
            else:
                t = math.tan(theta)
                ca = math.cos(theta + theta_cross - 1e-4) - math.sin(theta + theta_cross - 1e-4) / t
                cb = math.cos(theta_cross - 1e-4) + math.sin(theta_cross - 1e-4) / t
                alpha, beta = 10 * (ca / mean_norm + cb * alpha), 10 * cb * beta
                theta = theta + theta_cross - 1e-4
            coef.append((alpha, beta))
        if coef:
            ab = torch.tensor(coef, dtype=U.dtype, device=U.device)
            basis = torch.stack([mean, x_far])  # [2, D]
            step = max(1, (64 << 20) // max(U.shape[1], 1))  # rows per write: <= 256 MB of temporaries
            for i in range(0, len(coef), step):
                U[rows[i: i + step]] = ab[i: i + step] @ basis


def find_orthogonal_unit_vector(v: torch.Tensor) -> torch.Tensor:
    """A unit vector orthogonal to ``v`` (reference ``signguard_adversary.py:10-23`` keeps this helper although its
    attack never calls it): Gram-Schmidt of a fixed pseudo-random direction against ``v``."""
    v = v.reshape(-1).float()
    g = torch.Generator(device="cpu").manual_seed(0)
    r = torch.randn(v.numel(), generator=g).to(v.device)
    vv = torch.dot(v, v)
    if float(vv) > 0:
        r = r - torch.dot(r, v) / vv * v
    return r / r.norm().clamp_min(1e-12)
