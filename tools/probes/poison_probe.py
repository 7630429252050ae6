"""Root-cause probe for the round-1 "non-finite rows under HIP graphs that share a memory pool" symptom.

Runs multi-chunk ResNet-10 rounds (a) eagerly, (b) as HIP graphs with private pools, (c) as HIP graphs sharing one
pool — each with BLADES_POISON=1, i.e. every kernel workspace / output buffer pre-filled with NaN, so a kernel that
reads memory nothing wrote gives a non-finite row deterministically, whatever the allocator handed out.

    BLADES_POISON=1 python tools/probes/poison_probe.py [amp]
"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
os.environ["BLADES_SYNTHETIC_DATA"] = "1"
import torch

amp = sys.argv[1] if len(sys.argv) > 1 else "fp32"


def run(graph, shared, rounds=8, K=10, chunk=3):
    os.environ["BLADES_HIP_GRAPH"] = "1" if graph else "0"
    os.environ["BLADES_GRAPH_SHARED_POOL"] = "1" if shared else "0"
    from blades_amd.algorithms.fedavg import FedavgConfig
    cfg = (FedavgConfig().resources(num_gpus_for_driver=1)
           .data(dataset_config={"type": "cifar10", "num_clients": K, "num_classes": 10, "train_batch_size": 8,
                                 "n_train": K * 24, "n_test": 64, "data_root": tempfile.mkdtemp(), "use_cache": False})
           .training(global_model="resnet", server_config={"aggregator": {"type": "Median"}, "optimizer": {"lr": 0.0}})
           .client(client_config={"lr": 1.0})
           .adversary(num_malicious_clients=2, adversary_config={"type": "blades.adversaries.ALIEAdversary"})
           .evaluation(evaluation_interval=0).engine(clients_per_chunk=chunk, amp_dtype=amp))
    cfg.seed = 5
    algo = cfg.build()
    algo.engine.hip_graphs = "1" if graph else "0"
    out, bad = [], []
    for r in range(rounds):
        algo.train()
        U = algo.update_matrix[:, : algo.engine.D]
        if not bool(torch.isfinite(U).all()):
            bad.append((r, (~torch.isfinite(U)).any(dim=1).nonzero().flatten().tolist()))
        out.append(U.clone())
    return out, bad, algo.engine.graph_replays


print("poison:", os.environ.get("BLADES_POISON"), "amp:", amp, flush=True)
ref, bad, _ = run(False, False)
print(f"eager: non-finite rounds {bad}", flush=True)
for shared in (False, True):
    got, bad, rep = run(True, shared)
    same = [bool(torch.equal(a, b)) for a, b in zip(ref, got)]
    print(f"graphs shared_pool={shared}: replays {rep} non-finite rounds {bad} bitwise==eager per round {same}", flush=True)
