"""Aggregator semantics: the reference's 4x3 oracle (fllib/aggregators/tests/test_aggregators.py:25-55)
plus parity of every aggregator with a direct re-statement of the reference algorithm."""
import numpy as np
import pytest
import torch

from blades_amd.aggregators import (Centeredclipping, Clippedclustering, DnC, GeoMed, Mean, Median, Multikrum,
                                    Signguard, Trimmedmean)

RAW = torch.tensor([[1, 2, 3], [-1, 4, -1], [2.0, 2, 3], [3.0, 1.0, 3]])
INPUTS = [t.squeeze() for t in torch.split(RAW, 1)]


def test_mean_median_oracle():
    assert torch.allclose(Mean()(INPUTS), torch.tensor([1.25, 2.25, 2.0]))
    assert torch.allclose(Median()(INPUTS), torch.tensor([1.5, 2.0, 3.0]))


def test_trimmedmean_oracle_and_rounding():
    # the reference test builds Trimmedmean(2), which its own guard rejects for K=4 (stale test);
    # its expected vector is what trimming ONE value per side gives
    with pytest.raises(ValueError):
        Trimmedmean(2)(INPUTS)
    with pytest.raises(ValueError):
        Trimmedmean(1)(INPUTS)  # 1 rounds up to 2 ("round up to even", aggregators.py:31-38)
    t = Trimmedmean(1, filter_frac=0.5)  # int(0.5) = 0 -> plain mean
    assert t.num_excluded == 0
    five = INPUTS + [torch.tensor([0.0, 0.0, 0.0])]
    t = Trimmedmean(2)
    assert t.num_excluded == 2
    X = torch.stack(five)
    s = torch.sort(X, 0).values
    assert torch.allclose(t(five), s[2:3].mean(0))
    assert Trimmedmean(3).num_excluded == 4


def test_geomed_first_order_condition():
    out = GeoMed(eps=1e-8, maxiter=1000, ftol=1e-22)(INPUTS)
    cond = torch.vstack([(x - out) / torch.norm(x - out) for x in INPUTS]).sum(0)
    assert torch.allclose(cond, torch.zeros_like(cond), atol=1e-3)


def test_dnc_oracle():
    out = DnC(num_byzantine=1, sub_dim=3, num_iters=1)(INPUTS)
    # keeps the 3 rows with the smallest projection on the top singular direction -> drops [-1,4,-1]
    assert torch.allclose(out, torch.tensor([2.0, 1.6667, 3.0]), atol=1e-4)


def _ref_multikrum(U, f, m):
    n = len(U)
    d2 = torch.cdist(U, U) ** 2
    scores = []
    for i in range(n):
        s = sorted([float(d2[i, j]) ** 2 for j in range(n) if j != i])[: n - f - 2]
        scores.append(sum(s))
    idx = sorted(range(n), key=lambda i: scores[i])[:m]
    return U[idx].mean(0)


def test_multikrum_matches_reference_quirk():
    g = torch.Generator().manual_seed(0)
    U = torch.randn(12, 50, generator=g)
    U[:3] += 5.0
    for m in (1, 4):
        got = Multikrum(3, k=m)(list(U))
        assert torch.allclose(got, _ref_multikrum(U, 3, m), atol=1e-5)
    with pytest.raises(ValueError):
        Multikrum(6)(list(U))  # 2f + 2 > n


def test_centeredclipping_matches_reference_and_is_stateful():
    g = torch.Generator().manual_seed(1)
    U = torch.randn(8, 30, generator=g) * 3
    agg = Centeredclipping(tau=2.0, n_iter=3)
    mom = torch.zeros(30)
    for _ in range(2):  # two rounds: the momentum persists
        for _ in range(3):
            mom = sum((v - mom) * min(1.0, 2.0 / float(torch.norm(v - mom))) for v in U) / len(U) + mom
        got = agg(list(U))
        assert torch.allclose(got, mom, atol=1e-5)


def _features(U):
    D = U.shape[1]
    return np.array([[float((u > 0).sum()) / D, float((u < 0).sum()) / D, float((u == 0).sum()) / D] for u in U])


def test_signguard_matches_reference_with_sklearn():
    from sklearn.cluster import KMeans

    g = torch.Generator().manual_seed(2)
    U = torch.randn(20, 400, generator=g)
    U[:5] = -torch.rand(5, 400, generator=g) * 6  # sign-skewed, large attackers
    X = U.clone()
    norms = X.norm(dim=1)
    M = float(np.median(norms.numpy()))
    for i in range(20):
        if norms[i] > M:
            X[i] *= M / (norms[i] + 1e-6) if False else M / norms[i]
    n2 = X.norm(dim=1)
    S1 = {i for i in range(20) if 0.1 * M <= n2[i] <= 3.0 * M + 1e-5}
    lab = KMeans(n_clusters=2, random_state=0, n_init=10).fit(_features(X)).labels_
    flag = 1 if lab.sum() > 10 else 0
    S2 = {i for i in range(20) if lab[i] == flag}
    ref = X[sorted(S1 & S2)].mean(0)
    agg = Signguard()
    got = agg(list(U))
    assert int(agg.last_info["num_selected"]) == len(S1 & S2)
    assert torch.allclose(got, ref, atol=1e-5)
    assert not (S1 & S2) & set(range(5))  # attackers filtered


@pytest.mark.parametrize("linkage", ["average", "single"])
def test_clippedclustering_matches_sklearn(linkage):
    from sklearn.cluster import AgglomerativeClustering

    g = torch.Generator().manual_seed(3)
    base = torch.randn(200, generator=g)
    U = base + 0.3 * torch.randn(15, 200, generator=g)
    U[:4] = -base + 0.3 * torch.randn(4, 200, generator=g)  # opposite-direction attackers
    agg = Clippedclustering(linkage=linkage)
    got = agg(list(U))
    X = U.clone()
    norms = X.norm(dim=1)
    thr = float(np.median(norms.numpy()))
    for i in range(15):
        if norms[i] > thr:
            X[i] *= thr / norms[i]
    dis = np.zeros((15, 15))
    for i in range(15):
        for j in range(i + 1, 15):
            dis[i, j] = dis[j, i] = 1 - float(torch.nn.functional.cosine_similarity(X[i], X[j], dim=0))
    lab = AgglomerativeClustering(metric="precomputed", linkage=linkage, n_clusters=2).fit(dis).labels_
    flag = 1 if lab.sum() > 15 // 2 else 0
    ref = X[[i for i in range(15) if lab[i] == flag]].mean(0)
    assert torch.allclose(got, ref, atol=1e-5)
    assert len(agg.l2norm_his) == 15
    agg(list(U))
    assert len(agg.l2norm_his) == 30  # history grows by K every round (clippedclustering.py:33-35)


def test_median_variants_inner_agg():
    g = torch.Generator().manual_seed(4)
    U = torch.randn(9, 40, generator=g)
    got = Signguard(agg="median")(list(U))
    assert got.shape == (40,)
    with pytest.raises(NotImplementedError):
        Signguard(agg="foo")


def test_from_config_resolution():
    from blades_amd.utils.from_config import from_config

    assert isinstance(from_config(Mean, {"type": "Median"}), Median)
    assert isinstance(from_config(Mean, "Trimmedmean", num_byzantine=2), Trimmedmean)
    assert isinstance(from_config(Mean, {"type": "fllib.aggregators.Multikrum", "num_byzantine": 1}), Multikrum)
    assert isinstance(from_config(Mean, {"type": "Clippedclustering", "linkage": "single"}), Clippedclustering)
    assert isinstance(from_config(Mean, None), Mean)


def test_host_clustering_large_k_matches_sklearn():
    from sklearn.cluster import AgglomerativeClustering

    from blades_amd.ops import host

    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 1, (300, 8)), rng.normal(4, 1, (100, 8))])
    d = np.linalg.norm(x[:, None] - x[None], axis=-1)
    for linkage in ("average", "single", "complete"):
        a = host.agglomerative2(d, linkage)
        b = AgglomerativeClustering(metric="precomputed", linkage=linkage, n_clusters=2).fit(d).labels_
        assert (a == b).all() or (a == 1 - b).all(), linkage


def test_geomed_gram_formulation_matches_direct():
    """Weiszfeld in coefficient space (one Gram pass + O(K^2) iterations) gives the direct form's result."""
    from blades_amd.aggregators import GeoMed
    from blades_amd.parallel.context import local_context

    g = torch.Generator().manual_seed(5)
    U = torch.randn(40, 3000, generator=g) * 0.1
    U[:8] += 0.7
    ctx = local_context(3000)
    a = GeoMed(method="direct").aggregate(U, ctx)
    gm = GeoMed(method="gram")
    b = gm.aggregate(U, ctx)
    assert gm.last_info["method"] == "gram"
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)
    w = torch.rand(40, generator=g) + 0.1
    torch.testing.assert_close(GeoMed(method="direct").aggregate(U, ctx, w), GeoMed(method="gram").aggregate(U, ctx, w),
                               rtol=1e-4, atol=1e-6)
    # a row sitting exactly on the median (distance 0 -> eps clamp) behaves the same in both forms
    V = torch.cat([U[:5], U[:5].mean(0, keepdim=True)])
    torch.testing.assert_close(GeoMed(method="direct").aggregate(V, ctx), GeoMed(method="gram").aggregate(V, ctx),
                               rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("name", ["Signguard", "Clippedclustering"])
def test_call_does_not_modify_the_callers_tensor(name):
    """Norm-clipping aggregators rescale rows in place on the engine's resident matrix; ``__call__`` on a
    user-owned tensor works on a copy (the reference stacks its input list first)."""
    import blades_amd.aggregators as A

    torch.manual_seed(0)
    U = torch.randn(12, 40)
    U[3] *= 50
    keep = U.clone()
    getattr(A, name)()(U)
    assert torch.equal(U, keep)


def test_sample_distinct_is_a_draw_without_replacement():
    """ops.sample_distinct (DnC's coordinate draw; reference: np.random.choice(d, sub_dim, replace=False)): n distinct
    ids per set from a keyed permutation of [0, D); n = D gives a permutation; sets and seeds differ; roughly uniform."""
    from blades_amd import ops

    p = ops.sample_distinct(1000, 1000, 3, 7, "cpu")
    assert p.shape == (3, 1000) and p.dtype == torch.int64
    for row in p:
        assert torch.equal(row.sort().values, torch.arange(1000))
    assert not torch.equal(p[0], p[1]) and not torch.equal(p[0], torch.arange(1000))
    assert torch.equal(p, ops.sample_distinct(1000, 1000, 3, 7, "cpu"))
    assert not torch.equal(p, ops.sample_distinct(1000, 1000, 3, 8, "cpu"))
    D, n = 136074, 10000
    s = ops.sample_distinct(D, n, 5, 42, "cpu")
    assert all(r.unique().numel() == n for r in s) and int(s.min()) >= 0 and int(s.max()) < D
    hist = torch.histc(s.float(), bins=20, min=0, max=D)
    expect = 5 * n / 20
    assert float((hist - expect).abs().max()) < 5 * expect ** 0.5
    # position j of a set is not correlated with its id (a sorted or strided prefix would be)
    r = torch.corrcoef(torch.stack([torch.arange(n).double(), s[0].double()]))[0, 1]
    assert abs(float(r)) < 0.05
    # overlap of two sets is what independent draws give: n^2 / D expected common ids
    common = len(set(s[0].tolist()) & set(s[1].tolist()))
    assert abs(common - n * n / D) < 6 * (n * n / D) ** 0.5
    assert ops.sample_distinct(1, 1, 2, 0, "cpu").tolist() == [[0], [0]]
