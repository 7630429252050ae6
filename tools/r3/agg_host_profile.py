"""cProfile of one aggregator call on a resident matrix (small K: where do the milliseconds go?).

    python tools/r3/agg_host_profile.py DnC 60 resnet
"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from blades_amd import aggregators as A
from blades_amd.parallel.context import local_context

name, K, model = sys.argv[1], int(sys.argv[2]), sys.argv[3]
D = {"resnet18": 11_173_962, "resnet": 4_903_242, "cnn": 1_475_146, "mlp": 136_074}[model]
dev = torch.device("cuda:0")
f = K // 5
U = torch.randn(K, (D + 255) // 256 * 256, device=dev)[:, :D] * 0.01
U[:f] += 0.05
kw = {"DnC": {"num_byzantine": f}, "Multikrum": {"num_byzantine": f, "k": K - f}, "Trimmedmean": {"num_byzantine": f}}.get(name, {})
agg = getattr(A, name)(**kw)
ctx = local_context(D)
for _ in range(3):
    agg.aggregate(U, ctx)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    agg.aggregate(U, ctx)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue())
