"""Micro-benchmark of the update-matrix kernels at headline scale (ResNet-18 D, K clients).

usage: python tools/bench_ops.py [K ...]     (default 100 500 1000)
Prints ms/call and the effective HBM read bandwidth of each op.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from blades_amd import ops

D = 11_173_962
DPAD = (D + 255) // 256 * 256


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    Ks = [int(a) for a in sys.argv[1:]] or [100, 500, 1000]
    dev = torch.device("cuda:0")
    for K in Ks:
        U = torch.randn(K, DPAD, device=dev)
        gb = K * DPAD * 4 / 1e9
        f = K // 5
        rows = {
            "col_median": lambda: ops.col_median(U),
            "col_trimmed_mean": lambda: ops.col_trimmed_mean(U, f),
            "col_mean_std": lambda: ops.col_mean_std(U),
            "row_sqnorm": lambda: ops.row_sqnorm(U),
            "gram": lambda: ops.gram(U),
        }
        for name, fn in rows.items():
            ms = timeit(fn)
            print(f"K={K:5d} {name:18s} {ms:9.3f} ms   {gb / ms * 1e3 / 1e3:6.2f} TB/s read", flush=True)
        del U
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
