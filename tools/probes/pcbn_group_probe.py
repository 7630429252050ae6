"""Does running per-client BatchNorm in small client groups (reduce pass, then apply pass while the group's
tensors are still in the memory-side cache) beat one launch over the whole chunk?  Prints ms per variant."""
import sys
import time

import torch

from blades_amd import ops


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    h = ops._h()
    dev = torch.device("cuda")
    Kc, B = 200, 64
    for C, H in ((64, 32), (128, 16), (256, 8), (512, 4)):
        mk = lambda: torch.randn(Kc * B, C, H, H, device=dev, dtype=torch.bfloat16).contiguous(
            memory_format=torch.channels_last)
        x, gy = mk(), mk()
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        U = torch.zeros(Kc, 2 * C, device=dev)
        coef = torch.ones(Kc, device=dev)
        y, mean, invstd, gate = h.pcbn_forward(x, None, gamma, beta, Kc, 1e-5, True, want_mask=True)
        gate8 = gate.reshape(Kc, -1)
        mean, invstd = mean.reshape(Kc, C), invstd.reshape(Kc, C)

        def run(G):
            def f():
                for c0 in range(0, Kc, G):
                    c1 = min(Kc, c0 + G)
                    xs, gs = x[c0 * B: c1 * B], gy[c0 * B: c1 * B]
                    h.pcbn_forward(xs, None, gamma, beta, c1 - c0, 1e-5, True, want_mask=True)
            def b():
                for c0 in range(0, Kc, G):
                    c1 = min(Kc, c0 + G)
                    xs, gs = x[c0 * B: c1 * B], gy[c0 * B: c1 * B]
                    h.pcbn_backward(xs, gs, gate8[c0:c1].reshape(-1), gamma, mean[c0:c1].contiguous(),
                                    invstd[c0:c1].contiguous(), c1 - c0, coef[c0:c1], U[c0:c1], 0, C, False, False)
            return timed(f), timed(b)

        mb = x.numel() * 2 / 1e6
        line = [f"C={C} H={H} tensor={mb:.0f}MB"]
        for G in (200, 50, 20, 10, 5, 2):
            tf, tb = run(G)
            line.append(f"G={G}: fwd {tf:.3f} bwd {tb:.3f}")
        print(" | ".join(line), flush=True)


if __name__ == "__main__":
    sys.exit(main())
