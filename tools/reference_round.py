"""Measured stand-in for the missing published throughput ("no rounds/sec anywhere").

A plain-PyTorch restatement of what the reference does in one FL round, on one GPU, without Ray:
clients train *one after the other* on a single model instance (`worker.py:66-74`,
`fllib/clients/client.py:46-74`: save weights -> one SGD step on a batch -> pseudo-gradient = new - old ->
restore), the K flattened updates are stacked, ALIE forges the malicious rows
(`alie_adversary.py:28-41`), Median aggregates (`aggregators.py:12-16`), and the server takes an SGD step
(`server.py:109-130`).  fp32, like the reference.  Ray's RPC/serialisation overhead is *not* modelled, so
this is an optimistic baseline for the reference.

usage: python tools/reference_round.py [--clients 1000] [--rounds 2] [--model resnet18] [--batch 64]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.nn.functional as F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", type=int, default=1000)
    ap.add_argument("--malicious-frac", type=float, default=0.2)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="resnet18")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    from blades_amd.models.catalog import ModelCatalog
    from scipy.stats import norm

    dev = torch.device("cpu" if a.cpu else "cuda:0")
    torch.backends.cudnn.benchmark = True
    model = ModelCatalog.get_model(a.model).to(dev)
    K, f = a.clients, int(round(a.malicious_frac * a.clients))
    params = [p for p in model.parameters()]
    opt = torch.optim.SGD(params, lr=1.0)
    server_opt = torch.optim.SGD(params, lr=0.1)
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(K, a.batch, 3, 32, 32, device=dev, generator=g)
    Y = torch.randint(0, 10, (K, a.batch), device=dev, generator=g)
    s = K // 2 + 1 - f
    z = float(norm.ppf((K - f - s) / (K - f)))

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def one_round():
        updates = []
        model.train()
        for c in range(f, K):  # the adversary's clients do not train under ALIE
            saved = [p.detach().clone() for p in params]
            opt.zero_grad()
            loss = F.cross_entropy(model(X[c]), Y[c]).clamp(0, 1e6)
            loss.backward()
            opt.step()
            loss.item()  # the reference syncs on every batch (task.py:170-186)
            updates.append(torch.cat([(p.detach() - q).view(-1) for p, q in zip(params, saved)]))
            with torch.no_grad():
                for p, q in zip(params, saved):
                    p.copy_(q)
        benign = torch.vstack(updates)
        forged = benign.mean(0) + z * benign.std(0)
        U = torch.stack([forged] * f + updates, dim=0)
        agg = (U.median(dim=0)[0] - (-U).median(dim=0)[0]) / 2
        off = 0
        for p in params:
            p.grad = -agg[off: off + p.numel()].view_as(p)
            off += p.numel()
        server_opt.step()

    for _ in range(a.warmup):
        one_round()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.rounds):
        one_round()
    sync()
    dt = (time.perf_counter() - t0) / a.rounds
    print(json.dumps({"what": "sequential per-client PyTorch restatement of the reference round (fp32, no Ray)",
                      "model": a.model, "num_clients": K, "num_malicious": f, "batch": a.batch,
                      "s_per_round": round(dt, 4), "rounds_per_sec": round(1.0 / dt, 5),
                      "device": torch.cuda.get_device_name(0) if dev.type == "cuda" else "cpu"}))


if __name__ == "__main__":
    main()
