"""Dataset catalog (reference ``fllib/datasets/catalog.py``).

``dataset_config`` is either ``{custom_dataset: name|class|dotted.path,
custom_dataset_config: {...}, num_clients, train_batch_size}`` or
``{type: cifar10|mnist|fashionmnist, num_clients, train_batch_size, seed, iid,
alpha}`` (+ the offline extras ``synthetic``, ``n_train``, ``n_test``,
``data_root``).
"""
from pathlib import Path

from blades_amd.constants import DEFAULT_DATA_ROOT, FLLIB_DATASET
from blades_amd.utils.from_config import from_config
from blades_amd.utils.registry import _global_registry
from .cifar10 import CIFAR10
from .fashionmnist import FASHIONMNIST
from .fldataset import FLDataset
from .mnist import MNIST

__all__ = ["CIFAR10", "MNIST", "FASHIONMNIST", "DatasetCatalog", "make_dataset"]

_BUILTIN = {"cifar10": CIFAR10, "mnist": MNIST, "fashionmnist": FASHIONMNIST}


def make_dataset(identifier: str, num_clients: int, train_batch_size: int = 32, seed: int = 0, iid: bool = True,
                 alpha: float = 0.1, **kwargs) -> FLDataset:
    if identifier not in _BUILTIN:
        raise ValueError(f"Unknown dataset: {identifier}")
    data_root = kwargs.pop("data_root", None) or (Path.home() / DEFAULT_DATA_ROOT)
    return _BUILTIN[identifier](data_root=data_root, train_bs=train_batch_size or 32, num_clients=num_clients,
                                seed=seed, iid=iid, alpha=alpha, **kwargs)


class DatasetCatalog:
    @staticmethod
    def get_dataset(dataset_config=None, **dataset_kwargs) -> FLDataset:
        custom = dataset_config.get("custom_dataset")
        if custom:
            kw = dict(dataset_kwargs, **dataset_config.get("custom_dataset_config", {}))
            if isinstance(custom, str) and "." in custom:
                return from_config(custom, None, **kw)
            cls = custom if isinstance(custom, type) else _global_registry.get(FLLIB_DATASET, custom)
            if "num_clients" in dataset_config:
                kw["num_clients"] = dataset_config["num_clients"]
            if "train_batch_size" in dataset_config:
                kw["train_bs"] = dataset_config["train_batch_size"]
            return cls(**kw)
        if dataset_config.get("type") and dataset_config.get("num_clients"):
            extra = {k: dataset_config[k] for k in ("synthetic", "n_train", "n_test", "data_root", "use_cache")
                     if k in dataset_config}
            return make_dataset(
                dataset_config["type"], dataset_config["num_clients"],
                train_batch_size=dataset_config.get("train_batch_size"), seed=dataset_config.get("seed"),
                iid=dataset_config.get("iid", True), alpha=dataset_config.get("alpha", 0.1), **extra)
        raise NotImplementedError(f"Cannot build a dataset from {dataset_config}")

    @staticmethod
    def register_custom_dataset(dataset_name: str, dataset_class: type) -> None:
        _global_registry.register(FLLIB_DATASET, dataset_name, dataset_class)
