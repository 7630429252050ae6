"""ALIE, IPM, Noise, SignFlip, LabelFlip adversaries."""
from typing import Dict, Optional

import torch

from blades_amd import ops
from blades_amd.clients.callbacks import LabelFlipCallback, SignFlipCallback
from .adversary import Adversary


def _is_signguard(algorithm) -> bool:
    from blades_amd.aggregators import Signguard

    return isinstance(algorithm.server.aggregator, Signguard)


def half_negation_sign(ctx, device) -> torch.Tensor:
    """±1 per local column: −1 for global columns < D//2.

    The reference draws ``random.sample(range(D//2), D//2)`` as "negative
    indices" (``alie_adversary.py:33-39``), which is a permutation of the whole
    first half — i.e. exactly this mask."""
    col = torch.arange(ctx.d_local, device=device) + ctx.col_offset
    return torch.where(col < ctx.d_total // 2, -1.0, 1.0)


def alie_z_max(num_clients: int, num_byzantine: int) -> float:
    """``z_max = Phi^-1((n - f - s)/(n - f))``, ``s = floor(n/2) + 1 - f``
    (reference ``alie_adversary.py:18-23``)."""
    s = num_clients // 2 + 1 - num_byzantine
    cdf = (num_clients - num_byzantine - s) / (num_clients - num_byzantine)
    return float(torch.distributions.Normal(0.0, 1.0).icdf(torch.tensor(cdf, dtype=torch.float32)))


class ALIEAdversary(Adversary):
    """"A little is enough": every malicious row = ``mean + z_max * std`` of the
    benign rows (unbiased std) — one fused column pass (``ops.alie_forge_``)."""

    def __init__(self, clients, global_config: Dict = None):
        super().__init__(clients, global_config)
        self.num_clients = global_config.num_clients
        self.z_max = alie_z_max(self.num_clients, len(clients)) if len(clients) < self.num_clients else 0.0
        self.negative_indices = None

    def forge(self, algorithm, U, ctx):
        sign = half_negation_sign(ctx, U.device) if _is_signguard(algorithm) else None
        ops.alie_forge_(U, algorithm.benign_mask, algorithm.malicious_rows, self.z_max, sign)


class IPMAdversary(Adversary):
    """Inner-product manipulation: ``-scale * mean(benign)`` (``ipm_adversary.py``)."""

    def __init__(self, clients, global_config: Dict = None, scale: float = 1.0):
        super().__init__(clients, global_config)
        self._scale = scale

    def forge(self, algorithm, U, ctx):
        ops.ipm_forge_(U, algorithm.benign_mask, algorithm.malicious_rows, self._scale)


class NoiseAdversary(Adversary):
    """A fresh ``N(mean, std)`` vector per malicious client (``noise_adversary.py``).

    Drawn on the device with the counter-based generator: element (client g,
    coordinate c, round r) is a pure function of (seed, r, g, c), so the attack
    is identical whatever the number of GPUs."""

    def __init__(self, clients, global_config: Dict = None, mean: Optional[float] = 0.1,
                 std: Optional[float] = 0.1):
        super().__init__(clients, global_config)
        self._noise_mean = mean
        self._noise_std = std

    def forge(self, algorithm, U, ctx):
        rows = algorithm.malicious_rows
        pairs = torch.stack([rows.cpu(), algorithm.malicious_row_clients.cpu()], dim=1)
        seed = int(getattr(self.global_config, "seed", 0) or 0)
        ops.add_normal_rows_(U, pairs, seed, 2000 + self._round, ctx.col_offset, ctx.d_total,
                             float(self._noise_mean), float(self._noise_std), True)


class SignFlipAdversary(Adversary):
    """Malicious clients train, but ascend: ``grad -> -grad`` (``signflip_adversary.py``)."""

    def on_algorithm_start(self, algorithm):
        for client in self.clients:
            client.to_malicious(callbacks_cls=SignFlipCallback, local_training=True)


class LabelFlipAdversary(Adversary):
    """Malicious clients train on ``num_classes - 1 - y`` (``labelflip_adversary.py``)."""

    def on_algorithm_start(self, algorithm):
        num_class = self.global_config.dataset_config["num_classes"]
        for client in self.clients:
            client.to_malicious(callbacks_cls=LabelFlipCallback(num_class), local_training=True)
