"""The strided layers of the headline round (fp32 planes) for `rocprofv3 --pmc` passes (tools/profile_pmc_strided.sh):
conv_igemm forward as parity-class segments, conv_igemm data gradient (four class launches), conv_wgrad8m.
usage: python tools/probes/kernels_pmc_strided.py [Kc=20] [B=64]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from blades_amd import ops

Kc = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
h = ops._h()
dev = torch.device("cuda")
cl = torch.channels_last
for (ci, co, hw) in ((64, 128, 32), (128, 256, 16), (256, 512, 8)):
    x = torch.randn(Kc * B, ci, hw, hw, device=dev).contiguous(memory_format=cl)
    gy = torch.randn(Kc * B, co, hw // 2, hw // 2, device=dev).contiguous(memory_format=cl)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    U = torch.zeros(Kc, co * ci * 9, device=dev)
    coef = torch.ones(Kc, device=dev)
    for _ in range(2):
        assert h.conv_wgrad_(x, gy, Kc, U, 0, coef, False, 3, 2, 1, 3)
        y, _ = h.conv_forward(x, w, None, 2, 1, 3, stats_B=B)
        dx = h.conv_dgrad(gy, w, (hw, hw), 2, 1, 3)
    torch.cuda.synchronize()
    del x, gy, U
print("done", flush=True)
