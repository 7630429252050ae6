"""The kernels the fp32 headline round runs, on its shapes at bench scale (a 200-client chunk x batch 64), two launches
each, for `rocprofv3 --pmc` passes (tools/r3/profile_pmc3.sh): conv_igemm forward / data gradient (stride 1 and the
strided segments), conv_wgrad8 (stride 1), conv_wgrad8m (stride 2 parity classes), gram on bf16 planes.

usage: python tools/r3/kernels_pmc3.py [Kc=200] [B=64]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from blades_amd import ops

Kc = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
h = ops._h()
dev = torch.device("cuda")
cl = torch.channels_last
npl = 3
# (Ci, H, Co, stride)
for (ci, hw, co, s) in ((64, 32, 64, 1), (128, 16, 128, 1), (256, 8, 256, 1), (512, 4, 512, 1),
                        (64, 32, 128, 2), (128, 16, 256, 2), (256, 8, 512, 2)):
    ho = hw // s
    x = torch.randn(Kc * B, ci, hw, hw, device=dev).contiguous(memory_format=cl)
    gy = torch.randn(Kc * B, co, ho, ho, device=dev).contiguous(memory_format=cl)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    U = torch.zeros(Kc, co * ci * 9, device=dev)
    coef = torch.ones(Kc, device=dev)
    for _ in range(2):
        assert h.conv_wgrad_(x, gy, Kc, U, 0, coef, False, 3, s, 1, npl)
        y = h.conv_forward(x, w, None, s, 1, npl)
        dx = h.conv_dgrad(gy, w, (hw, hw), s, 1, npl)
    torch.cuda.synchronize()
    del x, gy, U, y, dx
Ug = torch.randn(1000, 1_400_000, device=dev)
for _ in range(2):
    G = ops.gram(Ug)
torch.cuda.synchronize()
print("done", flush=True)
