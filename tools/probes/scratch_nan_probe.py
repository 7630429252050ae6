"""Poison the wgrad split-K scratch with NaN before every call: any slab the kernel's reduce reads without a block
having written it shows up as NaN in the update rows (eager path, tiny chunks = large split factors)."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
os.environ["BLADES_SYNTHETIC_DATA"] = "1"
os.environ["BLADES_HIP_GRAPH"] = "0"
import torch
from blades_amd import ops
from blades_amd.algorithms.fedavg import FedavgConfig

h = ops._h()
orig = h._scratch
def poisoned(n, device):
    buf = orig(n, device)
    buf.fill_(float("nan"))
    return buf
h._scratch = poisoned
for chunk, K, bs in ((2, 8, 8), (3, 10, 8), (1, 4, 4), (5, 12, 16)):
    cfg = (FedavgConfig().resources(num_gpus_for_driver=1)
           .data(dataset_config={"type": "cifar10", "num_clients": K, "num_classes": 10, "train_batch_size": bs,
                                 "n_train": K * 20, "n_test": K * 8, "data_root": tempfile.mkdtemp(), "use_cache": False})
           .training(global_model="resnet", server_config={"aggregator": {"type": "Median"}, "optimizer": {"lr": 0.0}})
           .client(client_config={"lr": 1.0})
           .adversary(num_malicious_clients=2, adversary_config={"type": "blades.adversaries.ALIEAdversary"})
           .evaluation(evaluation_interval=0).engine(clients_per_chunk=chunk))
    cfg.seed = 5
    algo = cfg.build()
    for r in range(3):
        algo.train()
        U = algo.update_matrix[:, : algo.engine.D]
        bad = (~torch.isfinite(U)).sum(dim=0).nonzero().flatten()
        print(f"chunk={chunk} K={K} bs={bs} round {r}: non-finite columns {bad.numel()}"
              + (f" first {bad[:5].tolist()} last {bad[-3:].tolist()}" if bad.numel() else ""), flush=True)
