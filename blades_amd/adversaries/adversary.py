"""Adversary base class + config.

Hooks follow ``blades/adversaries/adversary.py:9-36``.  Update-forging attacks
implement ``forge(algorithm, U, ctx)`` and work **in place on the resident
update matrix** ``U [K, D_local]`` (a column shard when several GPUs take part):
coordinate-wise statistics are local, norm/dot statistics are local partials
plus a small all-reduce, and the forged vector is written straight into the
malicious rows.  Third-party adversaries written against the reference API
(override ``on_local_round_end`` and edit ``algorithm.local_results``) still
work on a single rank: ``Fedavg`` materialises ``local_results`` as row views
and copies replaced tensors back.
"""
import copy
from typing import Any, Dict, List, Type

import torch

from blades_amd.constants import CLIENT_UPDATE
from blades_amd.utils.from_config import from_config


class Adversary:
    def __init__(self, clients: List, global_config=None, malicious_client_cls=None):
        self.clients = clients
        self.global_config = global_config
        self._round = 0

    # -- reference-style access ---------------------------------------------
    def get_benign_updates(self, algorithm) -> torch.Tensor:
        if getattr(algorithm, "local_results", None):
            ups = []
            for result in algorithm.local_results:
                g = result.get(CLIENT_UPDATE, None)
                client = algorithm.client_manager.get_client_by_id(result["id"])
                if g is not None and not client.is_malicious:
                    ups.append(g)
            if not ups:
                raise ValueError("No benign updates found. To use this adversary, you must have at least one "
                                 "benign client.")
            return torch.vstack(ups)
        U = algorithm.update_matrix
        return U[algorithm.benign_mask > 0]

    @property
    def uses_local_results(self) -> bool:
        return (type(self).on_local_round_end is not Adversary.on_local_round_end
                and type(self).forge is Adversary.forge)

    # -- hooks ----------------------------------------------------------------
    def on_algorithm_start(self, algorithm):
        for client in self.clients:
            client.to_malicious(local_training=False)

    def on_local_round_end(self, algorithm):
        if type(self).forge is not Adversary.forge and len(self.clients) > 0:
            nb = getattr(algorithm, "num_benign", None)  # host-side count; a MagicMock algorithm has only the mask
            if (nb if isinstance(nb, int) else float(algorithm.benign_mask.sum())) == 0:
                raise ValueError("No benign updates found. To use this adversary, you must have at least one "
                                 "benign client.")
            self.forge(algorithm, algorithm.update_matrix, algorithm.ctx)
        self._round += 1

    def forge(self, algorithm, U: torch.Tensor, ctx) -> None:
        """Overwrite the malicious rows of ``U`` in place."""

    # -- checkpoint -------------------------------------------------------------
    def state_dict(self) -> dict:
        return {"round": self._round}

    def load_state_dict(self, s: dict) -> None:
        self._round = s.get("round", 0)


class AdversaryConfig:
    def __init__(self, adversary_cls=Adversary, config=None) -> None:
        self.adversary_class: Type["Adversary"] = adversary_cls
        self.global_config: Dict[str, Any] = config

    def to_dict(self) -> dict:
        d = dict(vars(self))
        gc = d.pop("global_config")
        d = copy.deepcopy(d)
        d["global_config"] = gc  # shared, not copied (it is the frozen algorithm config)
        return d

    def build(self, clients: List) -> "Adversary":
        config_dict = self.to_dict()
        cls = config_dict.pop("adversary_class")
        return from_config(cls, config_dict, clients=clients)

    def update_from_dict(self, config_dict: Dict) -> "AdversaryConfig":
        for key, value in config_dict.items():
            setattr(self, key, value)
        return self
