"""Where does the 128-channel parity-class weight-gradient kernel differ from the reference?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from blades_amd.ops import _hip as h

dev = torch.device("cuda")
Kc, B, Cin, Cout, H = int(sys.argv[1]), 8, 64, 128, 32
g = torch.Generator().manual_seed(2)
x = torch.randn(Kc * B, Cin, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
gy = torch.randn(Kc * B, Cout, H // 2, H // 2, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
n = Cout * Cin * 9
U = torch.full((Kc, n), 7.0, device=dev)
coef = torch.ones(Kc, device=dev)
assert h.conv_wgrad_(x, gy, Kc, U, 0, coef, False, 3, 2, 1, 3)
torch.cuda.synchronize()
w = torch.zeros(Cout, Cin, 3, 3, device=dev)
ref = torch.stack([torch.ops.aten.convolution_backward(gy[c * B:(c + 1) * B], x[c * B:(c + 1) * B], w, None, [2, 2], [1, 1],
                                                        [1, 1], False, [0, 0], 1, [False, True, False])[1] for c in range(Kc)])
got = U.view(Kc, Cout, Cin, 3, 3)
bad = ~torch.isfinite(got) | ((got - ref).abs() > 1e-3 * ref.abs().max())
print("bad total", int(bad.sum()), "of", bad.numel(), " nan", int((~torch.isfinite(got)).sum()), " untouched(7.0)", int((got == 7.0).sum()))
print("bad per tap", bad.sum(dim=(0, 1, 2)).tolist())
print("bad per client", bad.sum(dim=(1, 2, 3, 4)).tolist())
print("bad per co block of 16", bad.sum(dim=(0, 2, 3, 4)).view(-1, 16).sum(1).tolist())
print("bad per ci block of 16", bad.sum(dim=(0, 1, 3, 4)).view(-1, 16).sum(1).tolist())
