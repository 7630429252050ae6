"""Worker of test_distributed.py: runs a few FL rounds and (rank 0) saves theta."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BLADES_SYNTHETIC_DATA", "1")

import torch  # noqa: E402

from blades_amd.algorithms.fedavg import FedavgConfig  # noqa: E402
from blades_amd.parallel.runtime import init_runtime, shutdown_runtime  # noqa: E402


def main():
    spec = json.loads(sys.argv[1])
    out = sys.argv[2]
    rt = init_runtime(use_gpu=False)
    cfg = (FedavgConfig()
           .data(dataset_config={"type": "mnist", "num_clients": spec["K"], "num_classes": 10, "train_batch_size": 8,
                                 "n_train": spec["K"] * 24, "n_test": spec["K"] * 6, "data_root": spec["root"],
                                 "use_cache": False, "synthetic": True})
           .training(global_model=spec.get("model", "mlp"),
                     server_config={"aggregator": spec["agg"], "optimizer": {"lr": 0.5, "momentum": spec.get("smom", 0.0)}})
           .client(client_config={"lr": 0.5, "momentum": spec.get("cmom", 0.0), "num_batch_per_round": spec.get("nb", 1)})
           .adversary(num_malicious_clients=spec["f"], adversary_config=spec["adv"])
           .evaluation(evaluation_interval=spec["rounds"])
           .engine(clients_per_chunk=3, amp_dtype=None, device="cpu", update_layout=spec.get("layout", "alltoall")))
    cfg.seed = 11
    algo = cfg.build()
    # dropout stays on: the engine's masks are keyed by (seed, client, round, layer), not by the rank's RNG
    if spec.get("resume"):
        algo.load_checkpoint(spec["resume"])
    last = None
    for _ in range(spec["rounds"]):
        last = algo.train()
    if spec.get("ckpt") and rt.is_main:
        algo.save_checkpoint(spec["ckpt"])
    elif spec.get("ckpt"):
        algo.save_checkpoint(spec["ckpt"])
    if rt.is_main:
        torch.save({"theta": algo.server.theta[: algo.engine.D].clone(), "train_loss": last["train_loss"],
                    "acc": last.get("acc_top_1"), "ce": last.get("ce_loss"),
                    "comm_bytes": last.get("comm_bytes")}, out)
    shutdown_runtime()


if __name__ == "__main__":
    main()
