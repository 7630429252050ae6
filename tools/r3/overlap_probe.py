"""Do the HBM-bound BatchNorm kernels run UNDER a convolution kernel of another stream?
python tools/r3/overlap_probe.py   (BLADES_IGEMM_RF=1 for the 222-register kernel; default v1 = 247 registers)
A = conv_forward (64-channel 32x32 layer, 6400 images) x 4, B = pcbn_forward (apply pass dominated) x 12 on another stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from blades_amd.ops import _hip as h

dev = torch.device("cuda")
cl = torch.channels_last
N, C, HW = 6400, 64, 32
x = torch.randn(N, C, HW, HW, device=dev).contiguous(memory_format=cl)
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
xb = torch.randn(N, C, HW, HW, device=dev).contiguous(memory_format=cl)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
gy = torch.randn(N, C, HW, HW, device=dev).contiguous(memory_format=cl)
U = torch.zeros(100, C * C * 9, device=dev)
coef = torch.ones(100, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "igemm"

def A():
    for _ in range(4):
        if which == "igemm":
            h.conv_forward(x, w, None, 1, 1, 3)
        else:
            h.conv_wgrad_(x, gy, 100, U, 0, coef, False, 3, 1, 1, 3)

def B():
    for _ in range(12):
        h.pcbn_forward(xb, None, gamma, beta, 100, 1e-5, True, want_mask=True)

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fa, fb):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if fa:
        with torch.cuda.stream(s1): fa()
    if fb:
        with torch.cuda.stream(s2): fb()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for _ in range(2): timed(A, B)
ta = min(timed(A, None) for _ in range(3)); tb = min(timed(None, B) for _ in range(3)); tab = min(timed(A, B) for _ in range(3))
print(f"{which} RF={os.environ.get('BLADES_IGEMM_RF', 'hybrid')}: A alone {ta:.2f} ms, B (BatchNorm) alone {tb:.2f} ms, both on two streams {tab:.2f} ms "
      f"(sum {ta + tb:.2f}, max {max(ta, tb):.2f}) -> {100 * (ta + tb - tab) / min(ta, tb):.0f} % of the shorter one hidden")
