"""Stand-alone timing of the per-client BatchNorm kernels (fp32 activations) on the ResNet-18 shapes, 40 clients x 64
samples: bytes the tensors hold / time = achieved bandwidth.  python tools/probes/pcbn_time.py [Kc]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from blades_amd.ops import _hip as h

Kc = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = 64
dev = torch.device("cuda")
for C, H in [(64, 32), (128, 16), (256, 8), (512, 4)]:
    mk = lambda: torch.randn(Kc * B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    x, r, gy = mk(), mk(), mk()
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    U = torch.zeros(Kc, 2 * C, device=dev)
    coef = torch.ones(Kc, device=dev)
    nbytes = x.numel() * 4
    for res, want_gres, tag in [(None, False, "plain"), (r, True, "residual")]:
        def fwd():
            return h.pcbn_forward(x, res, gamma, beta, Kc, 1e-5, True, want_mask=True)
        y, mean, invstd, gate = fwd()
        def bwd():
            return h.pcbn_backward(x, gy, gate, gamma, mean, invstd, Kc, coef, U, 0, C, False, want_gres)
        for name, fn, traffic in [("fwd", fwd, (2 + (res is not None)) * nbytes + nbytes + nbytes / 32),
                                  ("bwd", bwd, 2 * 2 * nbytes + nbytes + 2 * nbytes / 32 + want_gres * nbytes)]:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            print(f"C={C:3d} {H:2d}x{H:<2d} {tag:8s} {name}: {ms:7.3f} ms  tensor {nbytes / 1e6:6.1f} MB  "
                  f"two-pass traffic {traffic / 1e6:7.1f} MB -> {traffic / ms / 1e6:6.2f} TB/s", flush=True)
