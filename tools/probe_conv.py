"""Probe MIOpen / hipBLASLt performance on ResNet-18 CIFAR shapes (bf16).

Run on the GPU box: python tools/probe_conv.py > gpurun_out/probe_conv.log
"""
import time, torch
import torch.nn.functional as F

dev = "cuda"
torch.backends.cudnn.benchmark = True

def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

print(torch.__version__, torch.cuda.get_device_name(0), flush=True)
x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
y = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
ms = timeit(lambda: x @ y)
print(f"gemm 8192^3 bf16: {ms:.3f} ms  {2*8192**3/ms/1e9:.1f} TFLOP/s", flush=True)
del x, y

KC = 32  # clients in the probe
B = 64
N = KC * B
shapes = [(64, 64, 32, 1), (64, 128, 32, 2), (128, 128, 16, 1), (128, 256, 16, 2),
          (256, 256, 8, 1), (256, 512, 8, 2), (512, 512, 4, 1)]
for fmt_name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
    for (ci, co, hw, s) in shapes:
        x = torch.randn(N, ci, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=fmt)
        w = torch.randn(co, ci, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=fmt)
        yy = F.conv2d(x, w, None, s, 1)
        gy = torch.randn_like(yy)
        flops = 2 * N * (hw // s) ** 2 * co * ci * 9
        try:
            t_f = timeit(lambda: F.conv2d(x, w, None, s, 1))
            t_d = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
            t_w = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
            print(f"{fmt_name} conv3x3 ci{ci} co{co} hw{hw} s{s} N{N}: fwd {t_f:.3f} ms ({flops/t_f/1e9:.0f} TF) dgrad {t_d:.3f} ms ({flops/t_d/1e9:.0f} TF) wgrad {t_w:.3f} ms ({flops/t_w/1e9:.0f} TF)", flush=True)
        except Exception as e:
            print("ERR", fmt_name, ci, co, hw, s, repr(e)[:200], flush=True)
        # grouped (per-client) wgrad: input [B, KC*ci, hw, hw], groups=KC
        try:
            xg = torch.randn(B, KC * ci, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=fmt)
            wg = torch.randn(KC * co, ci, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=fmt)
            yg = F.conv2d(xg, wg, None, s, 1, 1, KC)
            gyg = torch.randn_like(yg)
            t_gw = timeit(lambda: torch.ops.aten.convolution_backward(gyg, xg, wg, None, [s, s], [1, 1], [1, 1], False, [0, 0], KC, [False, True, False]), iters=3, warm=1)
            t_gf = timeit(lambda: F.conv2d(xg, wg, None, s, 1, 1, KC), iters=3, warm=1)
            print(f"   grouped g={KC}: fwd {t_gf:.3f} ms ({flops/t_gf/1e9:.0f} TF) wgrad {t_gw:.3f} ms ({flops/t_gw/1e9:.0f} TF)", flush=True)
        except Exception as e:
            print("ERR grouped", fmt_name, ci, co, hw, s, repr(e)[:200], flush=True)
# batchnorm on per-client view
x = torch.randn(B, KC * 64, 32, 32, device=dev, dtype=torch.bfloat16)
g = torch.ones(KC * 64, device=dev); b = torch.zeros(KC * 64, device=dev)
t = timeit(lambda: F.batch_norm(x, None, None, g, b, True))
print(f"bn per-client view [64,{KC*64},32,32] bf16: {t:.3f} ms ({x.numel()*2*2/t/1e6:.0f} GB/s)", flush=True)
# bmm
a = torch.randn(KC, 512, 64, device=dev, dtype=torch.bfloat16); bb = torch.randn(KC, 64, 10, device=dev, dtype=torch.bfloat16)
print("bmm small", timeit(lambda: torch.bmm(a, bb)), flush=True)
