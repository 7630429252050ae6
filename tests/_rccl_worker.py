"""Worker of test_rccl_single_rank.py: one rank, every aggregator/adversary pair, a few rounds each, with or
without ``BLADES_FORCE_COLLECTIVES`` (a one-rank process group whose collectives really run: RCCL on the GPU,
gloo on the CPU).  Saves theta / loss / accuracy / comm bytes per case."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BLADES_SYNTHETIC_DATA", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from blades_amd.algorithms.fedavg import FedavgConfig  # noqa: E402
from blades_amd.parallel.runtime import init_runtime, shutdown_runtime  # noqa: E402


def main():
    spec = json.loads(sys.argv[1])
    out = sys.argv[2]
    gpu = spec["device"] == "cuda"
    rt = init_runtime(use_gpu=gpu)
    forced = os.environ.get("BLADES_FORCE_COLLECTIVES") == "1"
    assert rt.distributed == forced
    if forced:
        assert dist.is_initialized() and dist.get_backend() == ("nccl" if gpu else "gloo")
    res = {"backend": rt.backend}
    if forced and gpu:  # (gloo has no send-to-self)
        # the point-to-point pattern of the update exchange (engine/trainer.py::_exchange_upto): several batches in
        # flight, row slabs of a resident matrix as buffers — to this rank itself, the only peer there is
        src = torch.arange(6 * 40, dtype=torch.float32, device=rt.device).view(6, 40)
        dst = torch.zeros_like(src)
        works = []
        for a, b in ((0, 2), (2, 6)):
            works += dist.batch_isend_irecv([dist.P2POp(dist.isend, src[a:b], 0), dist.P2POp(dist.irecv, dst[a:b], 0)])
        for w in works:
            w.wait()
        assert torch.equal(src, dst)
        from blades_amd.parallel.communicator import all_gather_uneven_rows

        assert torch.equal(all_gather_uneven_rows(src, [6]), src)
        res["p2p_self"] = True
    for i, (agg, adv, layout) in enumerate(spec["cases"]):
        cfg = (FedavgConfig()
               .data(dataset_config={"type": "mnist", "num_clients": spec["K"], "num_classes": 10,
                                     "train_batch_size": 8, "n_train": spec["K"] * 24, "n_test": spec["K"] * 6,
                                     "data_root": spec["root"], "use_cache": False, "synthetic": True})
               .training(global_model=spec.get("model", "mlp"),
                         server_config={"aggregator": agg, "optimizer": {"lr": 0.5, "momentum": 0.9}})
               .client(client_config={"lr": 0.5, "momentum": 0.0, "num_batch_per_round": spec.get("nb", 1)})
               .adversary(num_malicious_clients=spec["f"], adversary_config=adv)
               .evaluation(evaluation_interval=spec["rounds"])
               .engine(clients_per_chunk=3, amp_dtype=None, device=spec["device"], update_layout=layout))
        cfg.seed = 11
        algo = cfg.build()
        last = None
        for _ in range(spec["rounds"]):
            last = algo.train()
        ck = os.path.join(spec["root"], f"ck{i}_{int(forced)}")
        algo.save_checkpoint(ck)
        algo2 = cfg.build()
        algo2.load_checkpoint(ck)
        algo2.train()
        res[str(i)] = {"theta": algo.server.theta[: algo.engine.D].float().cpu().clone(),
                       "theta_resumed": algo2.server.theta[: algo2.engine.D].float().cpu().clone(),
                       "train_loss": float(last["train_loss"]), "acc": float(last.get("acc_top_1")),
                       "comm_bytes": dict(last.get("comm_bytes") or {})}
        print(f"case {i} done", flush=True)
    torch.save(res, out)
    shutdown_runtime()


if __name__ == "__main__":
    main()
