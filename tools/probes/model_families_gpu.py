"""All four model families through one FL round loop on the GPU (ALIE + Median), FedSGD and 2 local steps."""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
os.environ.setdefault("BLADES_SYNTHETIC_DATA", "1")
import torch
from blades_amd.algorithms.fedavg import FedavgConfig

for ds, model in (("mnist", "mlp"), ("fashionmnist", "cnn"), ("cifar10", "resnet"), ("cifar10", "cct")):
    for nb in (1, 2):
        root = tempfile.mkdtemp()
        cfg = (FedavgConfig().resources(num_gpus_for_driver=1)
               .data(dataset_config={"type": ds, "num_clients": 20, "num_classes": 10, "train_batch_size": 32,
                                     "n_train": 20 * 64, "n_test": 20 * 16, "data_root": root, "use_cache": False})
               .training(global_model=model, server_config={"aggregator": {"type": "Median"}, "optimizer": {"lr": 0.1}})
               .client(client_config={"lr": 0.5, "num_batch_per_round": nb})
               .adversary(num_malicious_clients=4, adversary_config={"type": "blades.adversaries.ALIEAdversary"})
               .evaluation(evaluation_interval=3))
        algo = cfg.build()
        t0 = time.perf_counter()
        for _ in range(3):
            r = algo.train()
        torch.cuda.synchronize()
        e = algo.engine
        path = "fused" if (nb == 1 and e.fused is not None) else "grouped" if (nb > 1 and e.grouped is not None) else "vmap"
        print(f"{ds:13s} {model:7s} nb={nb} path={path:8s} loss={r['train_loss']:.4f} acc={r.get('acc_top_1')} "
              f"{time.perf_counter() - t0:.1f}s", flush=True)
