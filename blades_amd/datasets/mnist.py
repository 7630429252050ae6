"""MNIST: no normalisation, no transform, images stay [N, 28, 28]
(reference ``fllib/datasets/mnist.py``)."""
from typing import Optional

from . import raw
from .fldataset import FLDataset


class MNIST(FLDataset):
    _name = "mnist"

    def __init__(self, data_root: str = "./data", cache_name: str = "", iid: Optional[bool] = True,
                 alpha: Optional[float] = 0.1, num_clients: Optional[int] = 20, seed=1, train_data=None,
                 test_data=None, train_bs: Optional[int] = 32, synthetic=None, n_train=None, n_test=None,
                 use_cache: bool = True):
        train_set = test_set = None
        if train_data is None:
            train_set, test_set, self.is_synthetic = raw.load(self._name, data_root, synthetic, 0, n_train, n_test)
            if self.is_synthetic:
                cache_name = cache_name or (f"{type(self).__name__}syn_{n_train}_{n_test}_num_clients_{num_clients}"
                                            f"_iid_{iid}_alpha_{alpha}_train_bs_{train_bs}_seed_{seed}.obj")
        super().__init__(train_set, test_set, data_root=data_root, cache_name=cache_name, iid=iid, alpha=alpha,
                         num_clients=num_clients, seed=seed, train_data=train_data, test_data=test_data,
                         train_bs=train_bs, use_cache=use_cache)
