"""Does MIOpen's exhaustive tuning (MIOPEN_FIND_ENFORCE=search) beat its default choice for the slowest ResNet-18
layer shape (3x3, 64->64 channels, 32x32, NHWC bf16, 200 clients x 64 samples)?  Run once per setting."""
import os, sys, time, torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
ci = co = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 32
x = torch.randn(N, ci, hw, hw, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, 3, 3, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
t0 = time.perf_counter()
y = F.conv2d(x, w, None, 1, 1)
torch.cuda.synchronize()
print(f"first call (find{'+tune' if os.environ.get('MIOPEN_FIND_ENFORCE') else ''}): {time.perf_counter() - t0:.1f}s", flush=True)
for _ in range(3):
    F.conv2d(x, w, None, 1, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    F.conv2d(x, w, None, 1, 1)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
fl = 2 * N * hw * hw * co * ci * 9
print(f"ENFORCE={os.environ.get('MIOPEN_FIND_ENFORCE')} N={N} c={ci} hw={hw}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
