"""Per-client Linear forward / data gradient: pc_linear.hip against the plane-split + conv_igemm route and torch.bmm.

    python tools/r3/bench_pc_linear.py [Kc=48] [R=64]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from blades_amd import ops


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    Kc = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    h = ops._h()
    dev = torch.device("cuda")
    for In, Out in ((2304, 600), (600, 10), (784, 128), (128, 256), (256, 10)):
        D = In * Out + Out
        W = torch.randn(Kc, (D + 7) // 8 * 8, device=dev) / In ** 0.5
        w = W.as_strided((Kc, Out, In), (W.stride(0), In, 1), 0)
        b = W.as_strided((Kc, Out), (W.stride(0), 1), In * Out)
        x = torch.randn(Kc * R, In, device=dev)
        gy = torch.randn(Kc * R, Out, device=dev)
        res = {}
        for name, rows in (("direct", 1 << 20), ("igemm", 0)):
            h._PC_LINEAR_MAX_ROWS = rows
            res[name] = (timeit(lambda: h.linear_forward(x, w, b, 3)), timeit(lambda: h.linear_dgrad(gy, w, 3)))
        wc = w.contiguous()
        res["bmm"] = (timeit(lambda: torch.baddbmm(b.unsqueeze(1), x.view(Kc, R, In), wc.transpose(1, 2))),
                      timeit(lambda: torch.bmm(gy.view(Kc, R, Out), wc)))
        mb = Kc * In * Out * 4 / 1e6
        print(f"Kc={Kc} R={R} {In}->{Out} ({mb:.0f} MB of weights): " + "  ".join(
            f"{k} fwd {v[0]:.0f} us dgrad {v[1]:.0f} us" for k, v in res.items())
            + f"   [direct fwd {mb / res['direct'][0]:.2f} TB/s]", flush=True)


if __name__ == "__main__":
    main()
