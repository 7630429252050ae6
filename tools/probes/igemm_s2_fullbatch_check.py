"""Strided forward convolutions (parity-class segments) at engine-sized batches against fp64: python tools/probes/igemm_s2_fullbatch_check.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blades_amd.ops import _hip as h

dev = torch.device("cuda")
torch.manual_seed(0)
for (N, Ci, HW, Co, k, s, p) in [(384, 64, 32, 128, 3, 2, 1), (384, 128, 16, 256, 3, 2, 1), (384, 256, 8, 512, 3, 2, 1),
                                 (100, 64, 32, 128, 3, 2, 1), (37, 256, 8, 512, 3, 2, 1), (50, 16, 12, 16, 3, 3, 0),
                                 (33, 32, 28, 64, 5, 2, 2)]:
    x = torch.randn(N, Ci, HW, HW, device=dev).relu().contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5
    y = h.conv_forward(x, w, None, s, p, 3)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, s, p)
    err = (y.double() - ref).abs()
    per_img = err.flatten(1).max(dim=1).values
    print(f"N={N} Ci={Ci} {HW}x{HW} Co={Co} k={k} s={s}: max err {float(err.max()):.3e} (|y| max {float(ref.abs().max()):.2f}); "
          f"worst image {int(per_img.argmax())}, first {float(per_img[0]):.2e}, last {float(per_img[-1]):.2e}, "
          f"median {float(per_img.median()):.2e}", flush=True)
