"""Tiny fixtures in the spirit of blades/adversaries/tests/simple_dataset.py: a 3-sample,
2-feature dataset and a Linear model, registered through the catalogs."""
import numpy as np
import torch

from blades_amd.datasets import DatasetCatalog, FLDataset


class SimpleDataset(FLDataset):
    def __init__(self, cache_name: str = "", iid=True, alpha=0.1, num_clients=3, seed=1, train_data=None,
                 test_data=None, train_bs=3, num_classes=2):
        features = np.array([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]] * num_clients, dtype=np.float32)
        labels = np.array([0, 1, 0] * num_clients)
        super().__init__((features, labels), (features.copy(), labels.copy()), cache_name=cache_name, iid=iid,
                         alpha=alpha, num_clients=num_clients, seed=seed, train_data=train_data,
                         test_data=test_data, train_bs=train_bs, is_image=False, num_classes=num_classes,
                         use_cache=False)


DatasetCatalog.register_custom_dataset("simple", SimpleDataset)


def simple_config(model=None, f=0, adversary=None, agg="Mean", client_lr=1.0, server_lr=0.1, num_clients=3, **kw):
    from blades_amd.algorithms.fedavg import FedavgConfig

    cfg = (FedavgConfig()
           .resources(num_remote_workers=0, num_gpus_per_worker=0)
           .data(dataset_config={"custom_dataset": "simple", "num_clients": num_clients, "train_batch_size": 3,
                                 "num_classes": 2})
           .training(global_model=model if model is not None else torch.nn.Linear(2, 2),
                     server_config={"lr": server_lr, "aggregator": {"type": agg}})
           .client(client_config={"lr": client_lr})
           .adversary(num_malicious_clients=f, adversary_config=adversary or {})
           .evaluation(evaluation_interval=1)
           .engine(device="cpu", amp_dtype=None, **kw))
    return cfg
