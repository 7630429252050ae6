"""A few launches of the per-client Linear kernels (pc_linear, and torch.bmm for comparison) for a rocprofv3 --pmc pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from blades_amd import ops

h = ops._h()
dev = torch.device("cuda")
Kc, R, In, Out = 48, 64, 2304, 600
W = torch.randn(Kc, In * Out + Out + 8, device=dev) / In ** 0.5
w = W.as_strided((Kc, Out, In), (W.stride(0), In, 1), 0)
b = W.as_strided((Kc, Out), (W.stride(0), 1), In * Out)
x = torch.randn(Kc * R, In, device=dev)
gy = torch.randn(Kc * R, Out, device=dev)
wc = w.contiguous()
for _ in range(3):
    h.linear_forward(x, w, b, 3)
    h.linear_dgrad(gy, w, 3)
    torch.baddbmm(b.unsqueeze(1), x.view(Kc, R, In), wc.transpose(1, 2))
    torch.bmm(gy.view(Kc, R, Out), wc)
torch.cuda.synchronize()
