"""CIFAR-10 federated dataset (reference ``fllib/datasets/cifar10.py``)."""
from typing import Optional

from . import raw
from .fldataset import FLDataset
from .transforms import BatchAugment, Normalize

CIFAR10_MEAN = (0.4914, 0.4822, 0.4465)
CIFAR10_STD = (0.2023, 0.1994, 0.2010)


class CIFAR10(FLDataset):
    def __init__(self, data_root: str = "./data", cache_name: str = "", iid: Optional[bool] = True,
                 alpha: Optional[float] = 0.1, num_clients: Optional[int] = 20, seed=1, train_data=None,
                 test_data=None, train_bs: Optional[int] = 64, synthetic=None, n_train=None, n_test=None,
                 use_cache: bool = True):
        train_set = test_set = None
        if train_data is None:
            train_set, test_set, self.is_synthetic = raw.load("cifar10", data_root, synthetic, 0, n_train, n_test)
            if self.is_synthetic:
                cache_name = cache_name or (f"CIFAR10syn_{n_train}_{n_test}_num_clients_{num_clients}_iid_{iid}"
                                            f"_alpha_{alpha}_train_bs_{train_bs}_seed_{seed}.obj")
        super().__init__(
            train_set, test_set, data_root=data_root, cache_name=cache_name, iid=iid, alpha=alpha,
            num_clients=num_clients, seed=seed, train_data=train_data, test_data=test_data, train_bs=train_bs,
            train_transform=BatchAugment(32, (0.75, 1.0), 0.5, CIFAR10_MEAN, CIFAR10_STD, erase_p=0.25),
            test_transform=Normalize(CIFAR10_MEAN, CIFAR10_STD), use_cache=use_cache,
        )
