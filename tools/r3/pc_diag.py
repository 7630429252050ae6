"""Does the per-client-weight fused executor stay on for a model at bench scale?  python tools/r3/pc_diag.py cnn fashionmnist"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("BLADES_SYNTHETIC_DATA", "1")
import torch
from blades_amd.algorithms.fedavg import FedavgConfig
model, ds = sys.argv[1], sys.argv[2]
K = 60
cfg = (FedavgConfig().resources(num_gpus_for_driver=1)
       .data(dataset_config={"type": ds, "num_clients": K, "num_classes": 10, "train_batch_size": 64, "synthetic": True,
                             "n_train": K * 128, "n_test": K * 8, "data_root": tempfile.mkdtemp(), "use_cache": False})
       .training(global_model=model, server_config={"aggregator": {"type": "Median"}, "optimizer": {"lr": 0.1}})
       .client(client_config={"lr": 0.1, "momentum": 0.0, "num_batch_per_round": 4})
       .adversary(num_malicious_clients=12, adversary_config={"type": "blades.adversaries.ALIEAdversary"})
       .evaluation(evaluation_interval=0).engine(amp_dtype="fp32"))
algo = cfg.build()
for i in range(3):
    r = algo.train()
    torch.cuda.synchronize()
    print(i, "fused_pc", algo.engine.fused_pc, "error", getattr(algo.engine, "fused_pc_error", None), "loss", r.get("train_loss"), flush=True)
