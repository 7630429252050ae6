import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from blades_amd.algorithms.fedavg import FedavgConfig
import tempfile
root = tempfile.mkdtemp()
K=1000
cfg = (FedavgConfig().resources(num_gpus_for_driver=1)
       .data(dataset_config={"type": "cifar10", "num_clients": K, "num_classes": 10, "train_batch_size": 8, "synthetic": True,
                             "n_train": K*16, "n_test": K*2, "data_root": root, "use_cache": False})
       .training(global_model="resnet18", server_config={"aggregator": {"type": "Mean"}, "optimizer": {"lr": 0.1}})
       .client(client_config={"lr": 1.0}).adversary(num_malicious_clients=200, adversary_config={"type": "blades.adversaries.ALIEAdversary"})
       .evaluation(evaluation_interval=0).engine(clients_per_chunk=200, amp_dtype="bf16"))
algo = cfg.build()
U = algo.engine.U_row
print("U", U.shape, U.stride(), U.data_ptr() % 256)
U.normal_()
from blades_amd import ops
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
ctx = algo.ctx
print("col_mean", t(lambda: ops.col_mean(U)))
print("agg.aggregate", t(lambda: algo.server.aggregator.aggregate(U, ctx)))
agg = algo.server.aggregator.aggregate(U, ctx)
print("isfinite", t(lambda: torch.isfinite(agg).all()))
print("apply_aggregate", t(lambda: algo.server.apply_aggregate(agg)))
print("server.step", t(lambda: algo.server.step(U, {"timestep": 1})))
