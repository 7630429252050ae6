"""The shipping hot kernels on headline-round shapes, a few launches each, for `rocprofv3 --pmc` passes
(tools/profile_pmc.sh): conv_wgrad (per-client weight gradient), conv_igemm (forward / data gradient), gram.

usage: python tools/probes/kernels_pmc.py [Kc=50] [B=64]
Shapes: ResNet-18 / CIFAR-10 layer1 (64->64, 32x32), layer2 (128->128, 16x16), layer3 (256->256, 8x8), fp32 planes
(npl=3, the headline dtype) and bf16 (npl=1); gram on a 1000 x 1.4 M slab (an eighth of the headline update matrix)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from blades_amd import ops

Kc = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
only = sys.argv[3] if len(sys.argv) > 3 else ""  # "wgrad": only the weight-gradient kernel (ablation passes)
h = ops._h()
dev = torch.device("cuda")
cl = torch.channels_last
for npl, dt in ((3, torch.float32), (1, torch.bfloat16)):
    for (c, hw) in ((64, 32), (128, 16), (256, 8)):
        x = torch.randn(Kc * B, c, hw, hw, device=dev).to(dt).contiguous(memory_format=cl)
        gy = torch.randn(Kc * B, c, hw, hw, device=dev).to(dt).contiguous(memory_format=cl)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.05
        U = torch.zeros(Kc, c * c * 9, device=dev)
        coef = torch.ones(Kc, device=dev)
        for _ in range(2):
            assert h.conv_wgrad_(x, gy, Kc, U, 0, coef, False, 3, 1, 1, npl)
            if only != "wgrad":
                y = h.conv_forward(x, w, None, 1, 1, npl)
                dx = h.conv_dgrad(gy, w, (hw, hw), 1, 1, npl)
        torch.cuda.synchronize()
        del x, gy, U
if only == "wgrad":
    print("done", flush=True)
    sys.exit(0)
Ug = torch.randn(1000, 1_400_000, device=dev)
for _ in range(2):
    G = ops.gram(Ug)
torch.cuda.synchronize()
print("done", flush=True)
