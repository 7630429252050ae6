"""Time every aggregator and every update-forging adversary on a resident update matrix.

usage: python tools/bench_aggregators.py [K=1000] [model=resnet18|resnet|cnn|mlp]
One GPU, column shard = full width.  Prints ms per call (median of 3 after one warm-up call).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from blades_amd.parallel.context import local_context

DIMS = {"resnet18": 11_173_962, "resnet": 4_903_242, "cnn": 1_475_146, "mlp": 136_074}


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    model = sys.argv[2] if len(sys.argv) > 2 else "resnet18"
    D = DIMS[model]
    f = K // 5
    dev = torch.device("cuda:0")
    from blades_amd import aggregators as A

    ctx = local_context(D)
    g = torch.Generator(device=dev).manual_seed(0)
    Dpad = (D + 255) // 256 * 256  # the engine's row stride (float4-aligned rows)
    U = (torch.randn(K, Dpad, device=dev, generator=g) * 0.01)[:, :D]
    U[:f] += 0.05  # a shifted malicious cluster so the filters have something to find
    specs = [("Mean", {}), ("Median", {}), ("Trimmedmean", {"num_byzantine": f}), ("GeoMed", {}),
             ("DnC", {"num_byzantine": f}), ("Multikrum", {"num_byzantine": f, "k": K - f}),
             ("Centeredclipping", {}), ("Clippedclustering", {}), ("Signguard", {})]
    print(f"K={K} D={D} ({model}), update matrix {K * D * 4 / 1e9:.1f} GB")
    for name, kw in specs:
        agg = getattr(A, name)(**kw)
        W = U
        if name in ("Clippedclustering", "Signguard"):  # these clip rows in place
            W = torch.empty(K, Dpad, device=dev)[:, :D].copy_(U)
        try:
            ms = timeit(lambda: agg.aggregate(W, ctx))
            print(f"{name:18s} {ms:10.2f} ms", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"{name:18s} failed: {type(e).__name__}: {e}", flush=True)
        del W
    # the update-forging adversaries on the same matrix (rows [0, f) are the malicious clients' rows)
    from unittest.mock import MagicMock

    from blades_amd import adversaries as ADV

    gcfg = MagicMock()
    gcfg.num_clients, gcfg.seed = K, 7
    algo = MagicMock()
    algo.update_matrix = U
    mask = torch.ones(K, device=dev)
    mask[:f] = 0
    algo.benign_mask = mask
    algo.malicious_rows = torch.arange(f, device=dev)
    algo.malicious_row_clients = torch.arange(f)
    algo.row_clients = torch.arange(K)
    algo.ctx = ctx
    algo.server.aggregator = A.Mean()
    algo.local_results = []
    for name, kw in [("ALIEAdversary", {}), ("IPMAdversary", {"scale": 0.1}), ("NoiseAdversary", {}),
                     ("AdaptiveAdversary", {}), ("MinMaxAdversary", {}), ("SignGuardAdversary", {}),
                     ("AttackclippedclusteringAdversary", {})]:
        try:
            adv = getattr(ADV, name)([object()] * f, gcfg, **kw)
            ms = timeit(lambda: adv.on_local_round_end(algo))
            print(f"{name:34s} {ms:10.2f} ms", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"{name:34s} failed: {type(e).__name__}: {e}", flush=True)


if __name__ == "__main__":
    main()
