"""How do the two conv_igemm workgroups of a CU interleave?  BLADES_IGEMM_PROF=2 python tools/r3/igemm_trace.py [--n 12800]
Uses the light trace build (four wall-clock stamps per workgroup, none in the K loop): per layer the mean prologue /
loop / epilogue time of a tile, the shader clock inside the loop, the share of a CU slot's time spent in each phase, and
for a CU in the middle of the chip how much of the wall time has 0 / 1 / 2 workgroups inside their K loop."""
import argparse, ctypes, os, sys, time
os.environ.setdefault("BLADES_IGEMM_PROF", "2")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blades_amd.ops import _hip as h

LAYERS = [(64, 32, 64), (128, 16, 128), (256, 8, 256), (512, 4, 512)]
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=12800); ap.add_argument("--dgrad", action="store_true")
a = ap.parse_args()
trace = h._lib.blades_conv_igemm_trace; trace.argtypes = [ctypes.c_void_p, ctypes.c_long]; trace.restype = ctypes.c_int
dev = torch.device("cuda")
for Ci, HW, Co in LAYERS:
    x = torch.randn(a.n, Ci, HW, HW, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, 3, 3, device=dev) / (Ci * 9) ** 0.5
    fn = (lambda: h.conv_dgrad(x, w, (HW, HW), 1, 1, 3)) if a.dgrad else (lambda: h.conv_forward(x, w, None, 1, 1, 3))
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    tiles = (a.n * HW * HW // 256) * (Co // 64)
    tiles = min(tiles, 1 << 20)
    buf = np.zeros((tiles, 8), dtype=np.uint64)
    assert trace(buf.ctypes.data_as(ctypes.c_void_p), tiles) == 0, "trace build not active"
    t = buf[:, :4].astype(np.int64)
    ok = t[:, 0] > 0
    t, cyc, hw, xcc = t[ok], buf[ok, 4].astype(np.int64), buf[ok, 5].astype(np.int64), buf[ok, 6].astype(np.int64)
    base = t[:, 0].min()
    us = (t - base) / 100.0  # s_memrealtime: 100 MHz
    pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    clk = cyc / np.maximum(loop, 1e-3) / 1e3  # GHz
    span = us[:, 3].max()
    print(f"Ci={Ci} {HW}x{HW} Co={Co} n={a.n}: {ms:.3f} ms/launch (trace build), {len(t)} tiles, device span {span / 1e3:.3f} ms")
    print(f"    per tile: prologue {pro.mean():6.2f} us  loop {loop.mean():6.2f} us  epilogue {epi.mean():6.2f} us   "
          f"(p10/p50/p90 loop {np.percentile(loop, 10):.1f}/{np.percentile(loop, 50):.1f}/{np.percentile(loop, 90):.1f}, "
          f"prologue {np.percentile(pro, 10):.1f}/{np.percentile(pro, 50):.1f}/{np.percentile(pro, 90):.1f}, "
          f"epilogue {np.percentile(epi, 10):.1f}/{np.percentile(epi, 50):.1f}/{np.percentile(epi, 90):.1f})")
    print(f"    shader clock inside the loop: {np.median(clk):.2f} GHz;  MFMA issue floor of a tile's loop at that clock: "
          f"{9 * Ci // 32 * 96 * 16 / np.median(clk) / 1e3:.2f} us per wave")
    # CU identity = (xcc, se, sh, cu) from HW_ID bits [15:8]
    cu = (xcc & 0xF) * 256 + ((hw >> 8) & 0xFF)
    ids, cnt = np.unique(cu, return_counts=True)
    print(f"    {len(ids)} CUs seen, tiles per CU {cnt.min()}..{cnt.max()}")
    # overlap census over all CUs: wall time with k workgroups inside the loop / inside any phase
    tot = np.zeros(4); res = np.zeros(4)
    for c in ids:
        m = cu == c
        ev = []
        for s, e in zip(us[m, 1], us[m, 2]): ev += [(s, 1, 0), (e, -1, 0)]
        for s, e in zip(us[m, 0], us[m, 3]): ev += [(s, 0, 1), (e, 0, -1)]
        ev.sort()
        k = r = 0; last = 0.0
        for tt, dk, dr in ev:
            tot[min(k, 3)] += tt - last; res[min(r, 3)] += tt - last
            last = tt; k += dk; r += dr
    print("    CU wall time with 0 / 1 / 2 workgroups in the K loop: " + " / ".join(f"{100 * v / tot.sum():.1f} %" for v in tot[:3]))
    print("    CU wall time with 0 / 1 / 2 workgroups resident:      " + " / ".join(f"{100 * v / res.sum():.1f} %" for v in res[:3]))
    c = ids[len(ids) // 2]
    m = np.where(cu == c)[0]
    m = m[np.argsort(us[m, 0])][20:32]
    print("    one CU, 12 consecutive tiles (us): start | loop start | loop end | end")
    for i in m:
        print(f"      {us[i, 0]:9.2f} {us[i, 1]:9.2f} {us[i, 2]:9.2f} {us[i, 3]:9.2f}   wave slot {hw[i] & 0xF} simd {(hw[i] >> 4) & 3}")
