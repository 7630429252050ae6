"""Adversaries: unit semantics on hand-made updates (MagicMock algorithm, like
blades/adversaries/tests/test_adaptive_attack.py) and integration through FedavgConfig.build()."""
import math
from unittest.mock import MagicMock

import numpy as np
import pytest
import torch

from blades_amd.adversaries import (ALIEAdversary, AdaptiveAdversary, AttackclippedclusteringAdversary,
                                    IPMAdversary, LabelFlipAdversary, MinMaxAdversary, NoiseAdversary,
                                    SignFlipAdversary, SignGuardAdversary)
from blades_amd.aggregators import Mean, Signguard
from blades_amd.parallel.context import local_context

from .simple_fixtures import simple_config


def mock_algo(U, f, aggregator=None):
    K, D = U.shape
    algo = MagicMock()
    algo.update_matrix = U
    mask = torch.ones(K)
    mask[:f] = 0
    algo.benign_mask = mask
    algo.malicious_rows = torch.arange(f)
    algo.malicious_row_clients = torch.arange(f)
    algo.row_clients = torch.arange(K)
    algo.ctx = local_context(D)
    algo.server.aggregator = aggregator or Mean()
    algo.local_results = []
    return algo


def cfg(n):
    c = MagicMock()
    c.num_clients = n
    c.seed = 7
    return c


def benign(K=10, f=3, D=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    U = torch.randn(K, D, generator=g)
    U[:f] = 0
    return U


def test_alie_matches_formula():
    U = benign()
    B = U[3:].clone()
    adv = ALIEAdversary([object()] * 3, cfg(10))
    s = 10 // 2 + 1 - 3
    z = torch.distributions.Normal(0.0, 1.0).icdf(torch.tensor((10 - 3 - s) / (10 - 3)))
    assert abs(adv.z_max - float(z)) < 1e-6
    adv.on_local_round_end(mock_algo(U, 3))
    exp = B.mean(0) + B.std(0) * z
    for r in range(3):
        assert torch.allclose(U[r], exp, atol=1e-5)
    assert torch.equal(U[3:], B)  # benign rows untouched


def test_alie_signguard_negates_first_half_of_std():
    U = benign(D=10)
    B = U[3:].clone()
    adv = ALIEAdversary([object()] * 3, cfg(10))
    adv.on_local_round_end(mock_algo(U, 3, Signguard()))
    std = B.std(0)
    std[:5] *= -1
    assert torch.allclose(U[0], B.mean(0) + std * adv.z_max, atol=1e-5)


def test_ipm():
    U = benign()
    B = U[3:].clone()
    IPMAdversary([object()] * 3, cfg(10), scale=100.0).on_local_round_end(mock_algo(U, 3))
    assert torch.allclose(U[1], -100.0 * B.mean(0), atol=1e-4)


def test_noise_is_fresh_per_client_and_round():
    U = benign(D=5000)
    adv = NoiseAdversary([object()] * 3, cfg(10), mean=0.1, std=0.1)
    a = mock_algo(U, 3)
    adv.on_local_round_end(a)
    first = U[:3].clone()
    assert not torch.allclose(first[0], first[1])
    assert abs(float(first.mean()) - 0.1) < 0.01 and abs(float(first.std()) - 0.1) < 0.01
    adv.on_local_round_end(a)
    assert not torch.allclose(U[0], first[0])


def test_adaptive_within_bounds():
    # port of the reference's bound check (test_adaptive_attack.py:30-42)
    U = torch.tensor([[0.0, 0.0, 0.0], [1.0, -2.0, 3.0], [2.0, -1.0, 1.0], [1.5, -3.0, 2.0], [0.5, -2.5, 2.5]])
    adv = AdaptiveAdversary([object()], cfg(5))
    out = adv.forge(mock_algo(U, 1), U, local_context(3))
    B = U[1:]
    mx, mn = B.max(0).values, B.min(0).values
    # mean > 0 & min > 0  ->  in [min/2, min];  mean < 0 & max < 0 -> in [max, max/2]
    assert mn[0] / 2 - 1e-6 <= out[0] <= mn[0] + 1e-6
    assert mx[1] - 1e-6 <= out[1] <= mx[1] / 2 + 1e-6
    assert mn[2] / 2 - 1e-6 <= out[2] <= mn[2] + 1e-6
    assert torch.equal(U[0], out)


def test_minmax_matches_reference_bisection():
    U = benign(K=12, f=3, D=80, seed=3)
    B = U[3:].clone()
    out = MinMaxAdversary([object()] * 3, cfg(12)).forge(mock_algo(U, 3), U, local_context(80))
    mean, dev = B.mean(0), B.std(0)
    thr = torch.cdist(B, B).max()
    low, high = 0.0, 5.0
    while abs(high - low) > 0.01:
        mid = (low + high) / 2
        loss = torch.cdist((mean - mid * dev)[None], B).max()
        if loss < thr:
            low = mid
        else:
            high = mid
    assert torch.allclose(out, mean - mid * dev, atol=1e-4)
    assert torch.allclose(U[2], out)


def test_signguard_attack_sign_statistics():
    U = benign(K=10, f=2, D=20000, seed=5)
    U[2:, :5000] = U[2:, :5000].abs()  # skew the mean's signs
    B = U[2:]
    out = SignGuardAdversary([object()] * 2, cfg(10)).forge(mock_algo(U, 2), U, local_context(20000))
    m = B.mean(0)
    # exact counts, like the reference's block construction + randperm (signguard_adversary.py:39-67); a U(0,1) draw
    # that is exactly 0 (probability 2^-24 each) would move a coordinate to the zero class
    assert abs(int((out > 0).sum()) - int((m > 0).sum())) <= 1
    assert abs(int((out < 0).sum()) - int((m < 0).sum())) <= 1
    assert float(out.abs().max()) <= 1.0
    # positions are spread over the whole vector, not the leading block
    half = out[:10000]
    assert 0.35 < float((half != 0).float().mean()) / max(float((out != 0).float().mean()), 1e-9) < 1.65


def test_attack_clippedclustering_chain():
    g = torch.Generator().manual_seed(6)
    base = torch.randn(300, generator=g)
    U = base + 0.5 * torch.randn(12, 300, generator=g)
    U[:3] = 0
    B = U[3:].clone()
    AttackclippedclusteringAdversary([object()] * 3, cfg(12)).on_local_round_end(mock_algo(U, 3))
    mean = B.mean(0)
    cos = [float(torch.nn.functional.cosine_similarity(U[r], mean, dim=0)) for r in range(3)]
    assert cos[0] > cos[1] > cos[2]  # each attacker is rotated further away from the benign mean
    assert all(abs(float(U[r].norm()) - 10.0) < 1.0 or float(U[r].norm()) > 5 for r in range(3))


# ---------------- integration through the config builder ----------------------------------------
@pytest.mark.parametrize("adv,cls", [
    ("blades.adversaries.LabelFlipAdversary", LabelFlipAdversary),
    ("blades.adversaries.SignFlipAdversary", SignFlipAdversary),
    ("blades.adversaries.IPMAdversary", IPMAdversary),
    ("blades.adversaries.ALIEAdversary", ALIEAdversary),
    ("blades.adversaries.NoiseAdversary", NoiseAdversary),
])
def test_adversary_wiring(adv, cls):
    algo = simple_config(f=1, adversary={"type": adv}).build()
    assert isinstance(algo.adversary, cls)
    clients = algo.client_manager.clients
    assert clients[0].is_malicious and not clients[1].is_malicious
    r = algo.train()
    assert isinstance(r, dict) and math.isfinite(r["train_loss"])


def test_fedsgd_equals_sgd_labelflip_zero_malicious():
    """The reference's equivalence oracle (test_labelflip_adversary.py:51-67): client lr 1.0 +
    server lr eta with Mean aggregation == one SGD step with lr eta on the mean client gradient."""
    torch.manual_seed(0)
    model = torch.nn.Linear(2, 2)
    ref = torch.nn.Linear(2, 2)
    ref.load_state_dict(model.state_dict())
    algo = simple_config(model=model, f=0, adversary={"type": "blades.adversaries.LabelFlipAdversary"},
                         server_lr=0.1).build()
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
    y = torch.tensor([0, 1, 0])
    opt = torch.optim.SGD(ref.parameters(), lr=0.1)
    for _ in range(3):
        algo.training_step()
        opt.zero_grad()
        torch.nn.functional.cross_entropy(ref(x), y).backward()  # every client holds the same 3 samples
        opt.step()
    for p, q in zip(algo.server.get_global_model().parameters(), ref.parameters()):
        assert torch.allclose(p, q, atol=1e-6)


def test_noise_attack_diverges_from_clean_sgd():
    torch.manual_seed(0)
    model = torch.nn.Linear(2, 2)
    ref = torch.nn.Linear(2, 2)
    ref.load_state_dict(model.state_dict())
    algo = simple_config(model=model, f=1, adversary={"type": "blades.adversaries.NoiseAdversary"}).build()
    algo.training_step()
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
    y = torch.tensor([0, 1, 0])
    opt = torch.optim.SGD(ref.parameters(), lr=0.1)
    torch.nn.functional.cross_entropy(ref(x), y).backward()
    opt.step()
    diffs = [float((p - q).abs().max()) for p, q in zip(algo.server.get_global_model().parameters(),
                                                        ref.parameters())]
    assert max(diffs) > 1e-3


def test_reference_style_custom_adversary_local_results():
    """A third-party adversary written against the reference API (edits algorithm.local_results)."""
    from blades_amd.adversaries import Adversary
    from blades_amd.constants import CLIENT_UPDATE

    class Ones(Adversary):
        def on_local_round_end(self, algorithm):
            b = self.get_benign_updates(algorithm)
            for result in algorithm.local_results:
                client = algorithm.client_manager.get_client_by_id(result["id"])
                if client.is_malicious:
                    result[CLIENT_UPDATE] = torch.ones_like(b[0]) * 3.0

    import tests.test_adversaries as me

    me.Ones = Ones
    algo = simple_config(f=1, adversary={"type": "tests.test_adversaries.Ones"}).build()
    assert algo.adversary.uses_local_results
    algo.training_step()
    assert torch.allclose(algo.update_matrix[0], torch.full((6,), 3.0))


def test_dp_fedavg_clip_and_noise():
    from blades_amd.algorithms.fedavg import FedavgDPConfig

    cfg = simple_config(f=0)
    d = cfg.to_dict()
    d.pop("task_spec")
    dp = FedavgDPConfig().update_from_dict(d)
    dp.global_model = torch.nn.Linear(2, 2)
    dp.dataset_config["train_bs"] = 3
    dp.dp_clip_threshold = 0.01
    dp.dp_privacy_epsilon = 1e9  # ~no noise: isolates the clip
    algo = dp.build()
    assert abs(algo.config.dp_privacy_sensitivity - 2 * 0.01 / 3) < 1e-12
    algo.training_step()
    norms = algo.update_matrix.norm(dim=1)
    assert float(norms.max()) <= 0.01 + 1e-5
    dp2 = FedavgDPConfig().update_from_dict(d)
    dp2.global_model = torch.nn.Linear(2, 2)
    dp2.dataset_config["train_bs"] = 3
    dp2.dp_privacy_epsilon = 0.01  # huge noise
    a2 = dp2.build()
    a2.training_step()
    assert float(a2.update_matrix.norm(dim=1).min()) > 1.0
