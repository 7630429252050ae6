"""cProfile of the host side of a round (where does a launch-bound round spend its CPU time?).

    python tools/r3/host_profile.py --model cnn --dataset fashionmnist --local-steps 20 [--rounds 5]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("BLADES_SYNTHETIC_DATA", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="cnn")
    ap.add_argument("--dataset", default="fashionmnist")
    ap.add_argument("--clients", type=int, default=60)
    ap.add_argument("--local-steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    import torch
    from blades_amd.algorithms.fedavg import FedavgConfig
    from blades_amd.parallel.runtime import init_runtime, shutdown_runtime

    init_runtime(use_gpu=True)
    K, f = a.clients, int(round(0.2 * a.clients))
    cfg = (FedavgConfig().resources(num_gpus_for_driver=1)
           .data(dataset_config={"type": a.dataset, "num_clients": K, "num_classes": 10, "train_batch_size": 64,
                                 "synthetic": True, "n_train": K * 128, "n_test": K * 8,
                                 "data_root": tempfile.mkdtemp(), "use_cache": False})
           .training(global_model=a.model, server_config={"aggregator": {"type": "Median"},
                                                          "optimizer": {"lr": 0.1, "momentum": 0.0}})
           .client(client_config={"lr": 1.0, "momentum": 0.0, "num_batch_per_round": a.local_steps})
           .adversary(num_malicious_clients=f, adversary_config={"type": "blades.adversaries.ALIEAdversary"})
           .evaluation(evaluation_interval=0).engine(clients_per_chunk=32, amp_dtype="fp32"))
    algo = cfg.build()
    for _ in range(5):
        algo.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.rounds):
        algo.train()
    torch.cuda.synchronize()
    pr.disable()
    dt = (time.perf_counter() - t0) / a.rounds * 1e3
    print(f"{a.model} {K}x{a.local_steps}: {dt:.2f} ms per round under cProfile; graph replays {algo.engine.graph_replays}")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(a.top)
    print(s.getvalue())
    shutdown_runtime()


if __name__ == "__main__":
    main()
