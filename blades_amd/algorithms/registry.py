"""Algorithm registry (reference ``blades/algorithms/registry.py``)."""
import importlib

from blades_amd.utils.registry import register_trainable


def _import_fedavg():
    from blades_amd.algorithms.fedavg import Fedavg, FedavgConfig

    return Fedavg, FedavgConfig().to_dict()


def _import_fedavg_dp():
    from blades_amd.algorithms.fedavg import FedavgDP, FedavgDPConfig

    return FedavgDP, FedavgDPConfig().to_dict()


ALGORITHMS = {"FEDAVG": _import_fedavg, "FEDAVG_DP": _import_fedavg_dp}


def _get_algorithm_class(alg: str, return_config: bool = False):
    if alg in ALGORITHMS:
        cls, cfg = ALGORITHMS[alg]()
    elif "." in alg:
        mod, _, name = alg.rpartition(".")
        cls = getattr(importlib.import_module(mod), name)
        cfg = cls.get_default_config().to_dict()
    else:
        raise ValueError(f"Unknown algorithm {alg}.")
    return (cls, cfg) if return_config else cls


def register_all():
    for key in ALGORITHMS:
        register_trainable(key, _get_algorithm_class(key))
