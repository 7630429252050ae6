"""Multi-chunk HIP-graph rounds with one memory pool per graph: do the sporadic non-finite rows persist?"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
os.environ["BLADES_SYNTHETIC_DATA"] = "1"
os.environ["BLADES_HIP_GRAPH"] = "1"
import torch
from blades_amd.algorithms.fedavg import FedavgConfig
for trial in range(6):
    cfg = (FedavgConfig().resources(num_gpus_for_driver=1)
           .data(dataset_config={"type": "cifar10", "num_clients": 8, "num_classes": 10, "train_batch_size": 8,
                                 "n_train": 160, "n_test": 64, "data_root": tempfile.mkdtemp(), "use_cache": False})
           .training(global_model="resnet", server_config={"aggregator": {"type": "Median"}, "optimizer": {"lr": 0.0}})
           .client(client_config={"lr": 1.0})
           .adversary(num_malicious_clients=2, adversary_config={"type": "blades.adversaries.ALIEAdversary"})
           .evaluation(evaluation_interval=0).engine(clients_per_chunk=2, amp_dtype=(sys.argv[1] if len(sys.argv) > 1 else "fp32")))
    cfg.seed = 5
    algo = cfg.build()
    bad_rounds = []
    for r in range(8):
        algo.train()
        U = algo.update_matrix[:, : algo.engine.D]
        if not bool(torch.isfinite(U).all()):
            rows = (~torch.isfinite(U)).any(dim=1).nonzero().flatten().tolist()
            bad_rounds.append((r, rows))
    print(f"trial {trial}: replays {algo.engine.graph_replays} non-finite rounds {bad_rounds}", flush=True)
