"""One forward + backward of the per-client BatchNorm kernels on the ResNet-18 layer-1 shape (200 clients x 64
samples, 64 channels, 32x32) — small enough to run under `rocprofv3 --pmc` (FETCH_SIZE / WRITE_SIZE in separate
passes) to compare the bytes the kernels move with the bytes their tensors hold."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from blades_amd import ops

h = ops._h()
Kc, B, C, H = 200, 64, 64, 32
dev = torch.device("cuda")
mk = lambda: torch.randn(Kc * B, C, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
x, r, gy = mk(), mk(), mk()
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
U = torch.zeros(Kc, 2 * C, device=dev)
coef = torch.ones(Kc, device=dev)
for _ in range(2):
    y, mean, invstd, gate = h.pcbn_forward(x, r, gamma, beta, Kc, 1e-5, True, want_mask=True)
    dx, gres = h.pcbn_backward(x, gy, gate, gamma, mean, invstd, Kc, coef, U, 0, C, False, True)
torch.cuda.synchronize()
print(f"tensor bytes: {x.numel() * 2 / 1e6:.1f} MB each (x, residual, y, gy, dx, gres); mask {gate.numel() / 1e6:.1f} MB")
