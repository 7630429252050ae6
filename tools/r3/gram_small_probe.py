import sys, time, os
sys.path.insert(0, '/root/repo')
import torch
from blades_amd import ops
h = ops._h()
dev = torch.device('cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for K, D in ((60, 4_903_242), (60, 1_475_146), (100, 4_903_242), (128, 4_903_242), (200, 4_903_242), (300, 1_475_146)):
    U = torch.randn(K, (D + 255) // 256 * 256, device=dev)[:, :D]
    w = torch.rand(K, device=dev)
    m = torch.randn(D, device=dev)
    print(f"K={K} D={D} ({K * D * 4 / 1e9:.2f} GB): gram {timeit(lambda: h.gram(U)):.3f} ms, weighted_col_sum {timeit(lambda: ops.weighted_col_sum(U, w)):.3f} ms, "
          f"sqdist_to {timeit(lambda: ops.sqdist_to(U, m)):.3f} ms, torch.mv {timeit(lambda: torch.mv(U, m)):.3f} ms, row copy ideal {K * D * 4 / 5e9:.3f} ms", flush=True)
