"""A few training + evaluation rounds of the north-star shape at reduced client count, printing every round:
python tools/probes/e2e_eval_probe.py [clients=100] [rounds=3]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("BLADES_SYNTHETIC_DATA", "1")
from blades_amd.algorithms.fedavg import FedavgConfig

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
root = tempfile.mkdtemp()
cfg = (FedavgConfig()
       .resources(num_gpus_for_driver=1)
       .data(dataset_config={"type": "cifar10", "num_clients": K, "num_classes": 10, "train_batch_size": 64,
                             "n_train": K * 128, "n_test": K * 10, "data_root": root, "use_cache": False,
                             "synthetic": True})
       .training(global_model="resnet18", server_config={"aggregator": {"type": "Median"}, "optimizer": {"lr": 0.1}})
       .client(client_config={"lr": 1.0})
       .adversary(num_malicious_clients=K // 5, adversary_config={"type": "blades.adversaries.ALIEAdversary"})
       .evaluation(evaluation_interval=1))
cfg.seed = 1
algo = cfg.build()
print("built", flush=True)
for r in range(R):
    t0 = time.time()
    res = algo.train()
    print(f"round {r}: {time.time() - t0:.2f}s loss {res['train_loss']:.4f} acc {res.get('acc_top_1')} ce {res.get('ce_loss')}", flush=True)
