"""GPU check of ops/csrc/conv_igemm.hip: forward + data gradient against an fp64 CPU convolution, then timing.

    python tools/probes/conv_igemm_check.py [--time] [--n 12800]
"""
import argparse
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from blades_amd.ops import _hip as h  # noqa: E402

SHAPES = [
    # N, Ci, H, W, Co, k, stride, pad
    (4, 64, 32, 32, 64, 3, 1, 1),
    (3, 64, 32, 32, 128, 3, 2, 1),
    (3, 64, 32, 32, 128, 1, 2, 0),
    (5, 128, 16, 16, 128, 3, 1, 1),
    (5, 128, 16, 16, 256, 3, 2, 1),
    (9, 256, 8, 8, 256, 3, 1, 1),
    (9, 256, 8, 8, 512, 3, 2, 1),
    (9, 256, 8, 8, 512, 1, 2, 0),
    (37, 512, 4, 4, 512, 3, 1, 1),
    (6, 3, 32, 32, 64, 3, 1, 1),
    (7, 1, 28, 28, 32, 3, 1, 1),
    (7, 32, 14, 14, 64, 3, 1, 0),
    (2, 48, 13, 11, 40, 3, 1, 1),
    (2, 20, 9, 9, 24, 5, 1, 2),
    (2, 16, 15, 15, 16, 3, 2, 1),
    (2, 16, 12, 12, 16, 3, 3, 0),
]


def nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


def check(npl, verbose=True):
    dev = torch.device("cuda")
    worst = 0.0
    for (N, Ci, H, W, Co, k, s, p) in SHAPES:
        g = torch.Generator().manual_seed(N * 1000 + Ci)
        x = torch.randn(N, Ci, H, W, generator=g, dtype=torch.float64)
        w = torch.randn(Co, Ci, k, k, generator=g, dtype=torch.float64) / (Ci * k * k) ** 0.5
        b = torch.randn(Co, generator=g, dtype=torch.float64)
        x32, w32, b32 = x.float(), w.float(), b.float()
        xr = x32.double().requires_grad_(True)
        wr = w32.double()
        y_ref = F.conv2d(xr, wr, b32.double(), s, p)
        gy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64).float()
        dx_ref, = torch.autograd.grad(y_ref, xr, gy.double())
        dt = torch.bfloat16 if npl == 1 else torch.float32
        if npl == 1:  # the oracle sees the same rounded inputs
            xr = x32.bfloat16().double().requires_grad_(True)
            y_ref = F.conv2d(xr, wr.float().bfloat16().double(), b32.double(), s, p)
            dx_ref, = torch.autograd.grad(y_ref, xr, gy.bfloat16().double())
        xd = nhwc(x32.to(dev).to(dt))
        gyd = nhwc(gy.to(dev).to(dt))
        wd = w32.to(dev)
        y = h.conv_forward(xd, wd, b32.to(dev), s, p, npl)
        dx = h.conv_dgrad(gyd, wd, (H, W), s, p, npl)
        torch.cuda.synchronize()
        ey = (y.double().cpu() - y_ref.detach()).abs().max().item() / y_ref.abs().max().item()
        ed = (dx.double().cpu() - dx_ref).abs().max().item() / dx_ref.abs().max().item()
        # what plain fp32 arithmetic achieves on the same problem
        e32 = (F.conv2d(x32, w32, b32, s, p).double() - y_ref.detach()).abs().max().item() / y_ref.abs().max().item()
        eg = float('nan')
        if npl != 1:  # MIOpen's fp32 convolution on the same GPU
            yg = F.conv2d(xd, nhwc(wd), b32.to(dev), s, p)
            eg = (yg.double().cpu() - y_ref.detach()).abs().max().item() / y_ref.abs().max().item()
        worst = max(worst, ey, ed)
        if verbose:
            print(f"npl={npl} N={N} Ci={Ci} {H}x{W} Co={Co} k={k} s={s} p={p}: fwd {ey:.2e} dgrad {ed:.2e} "
                  f"(cpu fp32 conv {e32:.2e}, MIOpen fp32 conv {eg:.2e})", flush=True)
    return worst


def timing(npl, n):
    dev = torch.device("cuda")
    dt = torch.bfloat16 if npl == 1 else torch.float32
    layers = [(64, 32, 64, 3, 1, 1), (128, 16, 128, 3, 1, 1), (256, 8, 256, 3, 1, 1), (512, 4, 512, 3, 1, 1),
              (64, 32, 128, 3, 2, 1), (128, 16, 256, 3, 2, 1), (256, 8, 512, 3, 2, 1), (3, 32, 64, 3, 1, 1),
              (64, 32, 128, 1, 2, 0)]
    for (Ci, HW, Co, k, s, p) in layers:
        x = nhwc(torch.randn(n, Ci, HW, HW, device=dev).to(dt))
        w = torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5
        Ho = (HW + 2 * p - k) // s + 1
        gy = nhwc(torch.randn(n, Co, Ho, Ho, device=dev).to(dt))
        flops = 2.0 * n * Ho * Ho * Co * Ci * k * k
        for name, fn in (("fwd", lambda: h.conv_forward(x, w, None, s, p, npl)),
                         ("dgrad", lambda: h.conv_dgrad(gy, w, (HW, HW), s, p, npl))):
            if name == "dgrad" and Ci == 3:
                continue
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            dt_ms = (time.perf_counter() - t0) / reps * 1e3
            print(f"npl={npl} {name:5s} Ci={Ci:3d} {HW:2d}x{HW:<2d} Co={Co:3d} k={k} s={s}: {dt_ms:8.3f} ms  "
                  f"{flops / dt_ms / 1e9:7.1f} TFLOP/s effective ({flops * (6 if npl == 3 else 3 if npl == 2 else 1) / dt_ms / 1e9:7.1f} bf16-MFMA)",
                  flush=True)
        del x, gy


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--n", type=int, default=12800)
    ap.add_argument("--npl", type=int, nargs="*", default=[3, 2, 1])
    a = ap.parse_args()
    for npl in a.npl:
        wst = check(npl)
        print(f"npl={npl}: worst relative error {wst:.3e}", flush=True)
    if a.time:
        for npl in a.npl:
            timing(npl, a.n)
