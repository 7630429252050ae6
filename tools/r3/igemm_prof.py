"""Where does a conv_igemm workgroup's time go?  BLADES_IGEMM_PROF=1 python tools/r3/igemm_prof.py [--n 2048]
Runs the PROF build of conv_igemm_kernel<3,4,true> (s_memtime stamps at phase boundaries, summed over all waves) on the
ResNet-18 3x3 stride-1 layers and prints cycles per wave and share per phase."""
import argparse, ctypes, os, sys, time
os.environ.setdefault("BLADES_IGEMM_PROF", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blades_amd.ops import _hip as h

LAYERS = [(64, 32, 64), (128, 16, 128), (256, 8, 256), (512, 4, 512)]
NAMES = ["prologue", "load issue", "frag reads + MFMA", "barrier 1", "commit (B only)", "commit (A+B)", "barrier 2", "epilogue"]
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=2048); ap.add_argument("--dgrad", action="store_true")
a = ap.parse_args()
prof = h._lib.blades_conv_igemm_prof; prof.argtypes = [ctypes.c_void_p]; prof.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 16)()
dev = torch.device("cuda")
for Ci, HW, Co in LAYERS:
    x = torch.randn(a.n, Ci, HW, HW, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, 3, 3, device=dev) / (Ci * 9) ** 0.5
    fn = (lambda: h.conv_dgrad(x, w, (HW, HW), 1, 1, 3)) if a.dgrad else (lambda: h.conv_forward(x, w, None, 1, 1, 3))
    for _ in range(3): fn()
    assert prof(buf) == 0, "profiling build not active"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 5
    for _ in range(reps): fn()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    prof(buf)
    v = list(buf)[:9]; waves = max(v[8], 1); tot = max(sum(v[:8]), 1)
    steps = 9 * Ci // 32
    print(f"Ci={Ci} {HW}x{HW} Co={Co} n={a.n}: {ms:.3f} ms/launch (PROF build), {waves // reps} waves/launch, {steps} steps/tile, "
          f"{tot / waves:.0f} cycles/wave (MFMA issue floor {steps * 96 * 16})")
    for nm, c in zip(NAMES, v[:8]):
        print(f"    {nm:20s} {c / waves:10.0f} cyc/wave  {100.0 * c / tot:5.1f} %")
