"""The CNN's first layer as a row GEMM: [Kc * 64 * 784 rows, 16] x [32, 16]^T through pc_linear.hip against conv_igemm's 1x1 path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from blades_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


h = ops._h()
dev = torch.device("cuda")
Kc, B, HW, wd, Co = 48, 64, 28, 16, 32
M = Kc * B * HW * HW
col = torch.randn(M, wd, device=dev)
W = torch.randn(Kc, Co * wd + Co + 8, device=dev)
w = W.as_strided((Kc, Co, wd), (W.stride(0), wd, 1), 0)
b = W.as_strided((Kc, Co), (W.stride(0), 1), Co * wd)
h._PC_LINEAR_MAX_ROWS = 1 << 30
t_pc = timeit(lambda: h.linear_forward(col, w, b, 3))
h._PC_LINEAR_MAX_ROWS = 0
t_ig = timeit(lambda: h.linear_forward(col, w, b, 3))
col4 = col.view(Kc * B, HW, HW, wd).permute(0, 3, 1, 2)
w4 = w.as_strided((Kc, Co, wd, 1, 1), (W.stride(0), wd, 1, 1, 1))
t_conv = timeit(lambda: h.conv_forward(col4, w4, b, 1, 0, 3))
ws = w[0].contiguous()
h._PC_LINEAR_MAX_ROWS = 1 << 30
t_pc1 = timeit(lambda: h.linear_forward(col, ws.unsqueeze(0), b[:1], 3))
t_ig1 = timeit(lambda: h.conv_forward(col4, ws.view(Co, wd, 1, 1), b[0].contiguous(), 1, 0, 3))
print(f"per-client weights: pc_linear {t_pc:.0f} us, linear via igemm {t_ig:.0f} us, conv_forward 1x1 {t_conv:.0f} us; "
      f"shared weights: pc_linear {t_pc1:.0f} us, conv_forward {t_ig1:.0f} us   (ideal at 5 TB/s: {(M * wd * 4 + M * Co * 4) / 5e6:.0f} us)")
