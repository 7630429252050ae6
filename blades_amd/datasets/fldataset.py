"""Federated dataset: partition + cache + per-client loaders.

Behavioural contract of ``fllib/datasets/fldataset.py``:

* images are scaled by 1/255; the **test set is split evenly and unshuffled**
  (``np.array_split``); the IID train split is ``np.array_split`` of the
  **unshuffled** train set; the non-IID split is per-class Dirichlet(alpha) with
  the "cap at N/num_clients" rule, retried until every client has >= 10 samples
  (native partitioner, ``ops/csrc/host.cpp``);
* partitions are cached in ``<data_root>/<Class>_num_clients_…_seed_….obj`` as
  five consecutive pickles (meta, train ids, train dict, test ids, test dict)
  and regenerated when the meta differs;
* the train loader of a client is an **infinite generator** of
  ``(Tensor[bs,…], LongTensor[bs])`` that reshuffles every epoch and applies the
  train transform to the whole batch; the test loader yields batches of 128.

New here: ``device_store(ids, device)`` uploads the partitions of a set of
clients once (concatenated, HBM-resident) for the batched engine, which then
draws per-client batches with a device-side gather (``engine/sampler.py``).
"""
import copy
import logging
import os
import pickle
import random
from pathlib import Path
from typing import Dict, Generator, List, Optional, Tuple

import numpy as np
import torch

from blades_amd.constants import DEFAULT_DATA_ROOT

logger = logging.getLogger(__name__)


def set_random_seed(seed_value=0, use_cuda=False):
    np.random.seed(seed_value)
    random.seed(seed_value)
    torch.manual_seed(seed_value)
    os.environ["PYTHONHASHSEED"] = str(seed_value)
    if use_cuda and torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed_value)


class CustomTensorDataset(torch.utils.data.Dataset):
    def __init__(self, data_X, data_y, transform_list=None):
        self.tensors = (data_X, data_y)
        self.transforms = transform_list

    def __getitem__(self, index):
        x = self.tensors[0][index]
        if self.transforms:
            x = self.transforms(x)
        return x, self.tensors[1][index]

    def __len__(self):
        return self.tensors[1].size(0)


class _TestLoader:
    """Sequential batches of a client's test shard (batch-level transform)."""

    def __init__(self, x: torch.Tensor, y: torch.Tensor, bs: int, transform=None):
        self.x, self.y, self.bs, self.transform = x, y, bs, transform
        self.dataset = CustomTensorDataset(x, y, transform)

    def __iter__(self):
        for i in range(0, len(self.y), self.bs):
            xb = self.x[i: i + self.bs]
            if self.transform:
                xb = self.transform(xb)
            yield xb, self.y[i: i + self.bs]

    def __len__(self):
        return -(-len(self.y) // self.bs)


class DeviceStore:
    """HBM-resident partitions of a set of clients (concatenated)."""

    def __init__(self, ids, xs, ys, device, dtype=torch.float32):
        self.ids = list(ids)
        self.lengths = np.array([len(y) for y in ys], dtype=np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.lengths)[:-1]]).astype(np.int64)
        if len(xs):
            self.x = torch.from_numpy(np.concatenate(xs, axis=0)).to(device=device, dtype=dtype)
            self.y = torch.from_numpy(np.concatenate(ys, axis=0).astype(np.int64)).to(device)
        else:
            self.x = torch.empty(0, device=device)
            self.y = torch.empty(0, dtype=torch.int64, device=device)
        self.device = device


class FLDataset:
    def __init__(
        self,
        train_set=None,
        test_set=None,
        data_root: str = "",
        cache_name: str = "",
        iid: Optional[bool] = True,
        alpha: Optional[float] = 0.1,
        num_clients: Optional[int] = 20,
        num_classes: Optional[int] = 10,
        seed=1,
        train_data=None,
        test_data=None,
        train_bs: Optional[int] = 32,
        test_bs: Optional[int] = 128,
        train_transform=None,
        test_transform=None,
        is_image=True,
        use_cache: bool = True,
        reference_compat_rng: bool = True,
    ):
        self.train_transform = train_transform
        self.test_transform = test_transform
        self.num_classes = num_classes
        self.train_bs = train_bs
        self.test_bs = test_bs
        self.is_image = is_image
        self.reference_compat_rng = reference_compat_rng
        if train_data:
            self.train_data = train_data
            self.test_data = test_data
            self._preprocess()
            return

        if data_root == "":
            data_root = Path.home() / DEFAULT_DATA_ROOT
        meta_info = {
            "num_clients": num_clients,
            "data_root": str(data_root),
            "train_bs": train_bs,
            "iid": iid,
            "alpha": alpha,
            "seed": seed,
            "is_image": is_image,
            "n_train": int(len(train_set[1])),
        }
        loaded = None
        if use_cache:
            os.makedirs(data_root, exist_ok=True)
            if cache_name == "":
                cache_name = (f"{self.__class__.__name__}_num_clients_{num_clients}_iid_{iid}"
                              f"_alpha_{alpha}_train_bs_{train_bs}_seed_{seed}.obj")
            self._data_path = os.path.join(data_root, cache_name)
            if os.path.exists(self._data_path):
                try:
                    with open(self._data_path, "rb") as f:  # file written by this framework
                        if pickle.load(f) == meta_info:
                            loaded = [pickle.load(f) for _ in range(4)]
                        else:
                            logger.warning("data-partition arguments differ from the cache; regenerating.")
                except Exception:  # corrupt cache -> regenerate
                    loaded = None
        if loaded is None:
            loaded = list(self._generate_datasets(train_set, test_set, iid=iid, alpha=alpha,
                                                  num_clients=num_clients, seed=seed))
            if use_cache:
                with open(self._data_path, "wb") as f:
                    pickle.dump(meta_info, f)
                    for obj in loaded:
                        pickle.dump(obj, f)
        train_clients, self.train_data, test_clients, self.test_data = loaded
        assert sorted(train_clients) == sorted(test_clients)
        self._preprocess()

    def __reduce__(self):
        return (_rebuild, (self.train_data, self.test_data, self.train_bs, self.test_bs, self.train_transform,
                           self.test_transform, self.num_classes))

    # ------------------------------------------------------------------
    def _generate_datasets(self, train_set, test_set, *, iid=True, alpha=0.1, num_clients=20, seed=1):
        x_test, y_test = test_set
        x_train, y_train = train_set
        y_train = np.asarray(y_train).reshape(-1)
        y_test = np.asarray(y_test).reshape(-1)
        if self.is_image:
            x_train = x_train.astype("float32") / 255.0
            x_test = x_test.astype("float32") / 255.0
        ids = [str(i) for i in range(num_clients)]
        x_test_splits = np.array_split(x_test, num_clients)
        y_test_splits = np.array_split(y_test, num_clients)
        if iid:
            x_train_splits = np.array_split(x_train, num_clients)
            y_train_splits = np.array_split(y_train, num_clients)
        else:
            from blades_amd.ops import host

            owner = host.dirichlet_partition(y_train, self.num_classes, num_clients, alpha, seed, min_size=10)
            rng = np.random.default_rng(seed)
            x_train_splits, y_train_splits = [], []
            for j in range(num_clients):
                idx = np.nonzero(owner == j)[0]
                rng.shuffle(idx)
                x_train_splits.append(x_train[idx])
                y_train_splits.append(y_train[idx])
        train_dataset = {i: {"x": x_train_splits[k], "y": y_train_splits[k].flatten()} for k, i in enumerate(ids)}
        test_dataset = {i: {"x": x_test_splits[k], "y": y_test_splits[k].flatten()} for k, i in enumerate(ids)}
        return ids, train_dataset, ids, test_dataset

    def _train_generator(self, data, labels, batch_size, seed=0) -> Generator[Tuple[torch.Tensor, torch.Tensor], None, None]:
        """Infinite shuffled batch stream (reference ``fldataset.py:230-251``).

        With ``reference_compat_rng`` the first ``next()`` re-seeds the *global*
        numpy/python/torch RNGs to ``seed`` exactly like the reference; otherwise
        a private generator is used."""
        if self.reference_compat_rng:
            set_random_seed(seed)
            perm = lambda n: np.random.permutation(n)  # noqa: E731
        else:
            rng = np.random.default_rng(seed)
            perm = lambda n: rng.permutation(n)  # noqa: E731
        i = 0
        idx = perm(len(labels))
        data, labels = data[idx], labels[idx]
        while True:
            if i * batch_size >= len(labels):
                i = 0
                idx = perm(len(labels))
                data, labels = data[idx], labels[idx]
            X = torch.as_tensor(data[i * batch_size: (i + 1) * batch_size], dtype=torch.float32)
            y = labels[i * batch_size: (i + 1) * batch_size]
            i += 1
            if self.train_transform:
                X = self.train_transform(X)
            yield X, torch.as_tensor(y, dtype=torch.long)

    def _preprocess(self):
        self._train_dls: Dict[str, Generator] = {}
        self._test_sets: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        for u_id in self.train_data.keys():
            self._train_dls[u_id] = self._train_generator(
                np.asarray(self.train_data[u_id]["x"]), np.asarray(self.train_data[u_id]["y"]), self.train_bs)
            self._test_sets[u_id] = (
                torch.as_tensor(np.asarray(self.test_data[u_id]["x"]), dtype=torch.float32),
                torch.as_tensor(np.asarray(self.test_data[u_id]["y"]), dtype=torch.long),
            )

    # ------------------------------------------------------------------
    @property
    def client_ids(self) -> List[str]:
        return list(self._train_dls.keys())

    def subset(self, u_ids: List[str]) -> "FLDataset":
        u_ids = set(u_ids)
        sub = copy.copy(self)
        sub.train_data = {k: v for k, v in self.train_data.items() if k in u_ids}
        sub.test_data = {k: v for k, v in self.test_data.items() if k in u_ids}
        sub._preprocess()
        return sub

    def split(self, n: int) -> List["FLDataset"]:
        """Randomly splits the clients into ``n`` disjoint near-equal subsets."""
        if n <= 0:
            raise ValueError(f"The number of splits {n} is not positive.")
        keys = list(self._train_dls.keys())
        random.shuffle(keys)
        size = len(keys) // n
        groups = [keys[i * size: (i + 1) * size] for i in range(n)]
        for i in range(len(keys) % n):
            groups[i].append(keys[-(i + 1)])
        return [self.subset(g) for g in groups]

    def get_train_loader(self, u_id: str) -> Generator:
        return self._train_dls[u_id]

    def get_test_loader(self, u_id):
        x, y = self._test_sets[u_id]
        return _TestLoader(x, y, self.test_bs, self.test_transform)

    def train_size(self, u_id) -> int:
        return len(self.train_data[u_id]["y"])

    def test_size(self, u_id) -> int:
        return len(self.test_data[u_id]["y"])

    def device_store(self, ids, device, split="train", dtype=torch.float32) -> DeviceStore:
        src = self.train_data if split == "train" else self.test_data
        xs = [np.asarray(src[i]["x"], dtype=np.float32) for i in ids]
        ys = [np.asarray(src[i]["y"]) for i in ids]
        return DeviceStore(ids, xs, ys, device, dtype)


def _rebuild(train_data, test_data, train_bs, test_bs, train_transform, test_transform, num_classes):
    return FLDataset(train_data=train_data, test_data=test_data, train_bs=train_bs, test_bs=test_bs,
                     train_transform=train_transform, test_transform=test_transform, num_classes=num_classes)
