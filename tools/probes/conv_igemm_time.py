"""Timing of conv_igemm on chosen layers: python tools/probes/conv_igemm_time.py --npl 3 --n 12800 [--layers 0 1]"""
import argparse, sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blades_amd.ops import _hip as h

LAYERS = [(64, 32, 64, 3, 1, 1), (128, 16, 128, 3, 1, 1), (256, 8, 256, 3, 1, 1), (512, 4, 512, 3, 1, 1),
          (64, 32, 128, 3, 2, 1), (128, 16, 256, 3, 2, 1), (256, 8, 512, 3, 2, 1), (3, 32, 64, 3, 1, 1),
          (64, 32, 128, 1, 2, 0)]
ap = argparse.ArgumentParser()
ap.add_argument("--npl", type=int, default=3)
ap.add_argument("--n", type=int, default=12800)
ap.add_argument("--layers", type=int, nargs="*", default=list(range(len(LAYERS))))
ap.add_argument("--dgrad", action="store_true")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda")
dt = torch.bfloat16 if a.npl == 1 else torch.float32
for li in a.layers:
    Ci, HW, Co, k, s, p = LAYERS[li]
    x = torch.randn(a.n, Ci, HW, HW, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5
    Ho = (HW + 2 * p - k) // s + 1
    gy = torch.randn(a.n, Co, Ho, Ho, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    flops = 2.0 * a.n * Ho * Ho * Co * Ci * k * k
    fns = [("fwd", lambda: h.conv_forward(x, w, None, s, p, a.npl))]
    if a.dgrad and Ci != 3:
        fns.append(("dgrad", lambda: h.conv_dgrad(gy, w, (HW, HW), s, p, a.npl)))
    for name, fn in fns:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        mult = {3: 6, 2: 3, 1: 1}[a.npl]
        print(f"npl={a.npl} {name:5s} Ci={Ci:3d} {HW:2d}x{HW:<2d} Co={Co:3d} k={k} s={s}: {ms:8.3f} ms "
              f"{flops / ms / 1e9:7.1f} TF eff ({flops * mult / ms / 1e9:7.1f} bf16-MFMA TF)", flush=True)
