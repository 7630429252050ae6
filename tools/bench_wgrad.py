"""Micro-benchmark of the per-client conv wgrad kernel on ResNet-18/CIFAR layer shapes.
usage: python tools/bench_wgrad.py [Kc=100] [B=64] [npl=1]   (npl 3 / 2: fp32 activations split into bf16 planes)"""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from blades_amd import ops

Kc = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
npl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dt = torch.bfloat16 if npl == 1 else torch.float32
h = ops._h()
dev = torch.device("cuda")
shapes = [(64, 64, 32, 3, 1, 1), (64, 128, 32, 3, 2, 1), (128, 128, 16, 3, 1, 1), (128, 256, 16, 3, 2, 1),
          (256, 256, 8, 3, 1, 1), (256, 512, 8, 3, 2, 1), (512, 512, 4, 3, 1, 1), (64, 128, 32, 1, 2, 0),
          (128, 256, 16, 1, 2, 0), (256, 512, 8, 1, 2, 0)]
tot = 0.0
for (ci, co, hw, ks, st, pad) in shapes:
    ho = (hw + 2 * pad - ks) // st + 1
    x = torch.randn(Kc * B, ci, hw, hw, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(Kc * B, co, ho, ho, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    n = co * ci * ks * ks
    U = torch.zeros(Kc, n, device=dev)
    coef = torch.ones(Kc, device=dev)
    if not h.conv_wgrad_(x, gy, Kc, U, 0, coef, False, ks, st, pad, npl):
        print(f"ci{ci:4d} co{co:4d} hw{hw:3d} k{ks} s{st}: does not fit (npl={npl})", flush=True)
        continue
    for _ in range(2):
        h.conv_wgrad_(x, gy, Kc, U, 0, coef, False, ks, st, pad, npl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 10
    e0.record()
    for _ in range(it):
        h.conv_wgrad_(x, gy, Kc, U, 0, coef, False, ks, st, pad, npl)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    fl = 2.0 * Kc * B * ho * ho * co * ci * ks * ks
    tot += ms
    print(f"ci{ci:4d} co{co:4d} hw{hw:3d} k{ks} s{st}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s ({fl * {1: 1, 2: 3, 3: 6}[npl] / ms / 1e9:7.1f} bf16-MFMA)", flush=True)
print(f"sum {tot:.2f} ms (ResNet-18 has 4/1/3/1/3/1/3 + 3 shortcut of these)")
