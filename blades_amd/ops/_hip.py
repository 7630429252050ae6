"""ctypes bindings of ``libblades_hip.so`` — the hand-written gfx950 kernels.

Kernels are launched on PyTorch's *current* HIP stream with raw device
pointers, so they order correctly with surrounding torch ops and are captured
by ``torch.cuda.graphs``.  Shapes/strides/dtypes are validated here on the host
before every launch: an out-of-bounds access in a kernel can take the whole
node down, so nothing reaches a kernel unchecked.

Importing this module without the built library raises — GPU tensors never
fall back to a PyTorch implementation.
"""
import ctypes
from ctypes import c_float, c_int, c_long, c_uint, c_ulonglong, c_void_p
from typing import Optional

import torch

from . import build as _build

if not _build.HIP_LIB.exists():
    raise ImportError(
        f"{_build.HIP_LIB} is missing: build it with `python -m blades_amd.ops.build` "
        "(hipcc cross-compiles gfx950 without a GPU). GPU tensors have no other implementation."
    )

# BLADES_HIP_LIB: another build with the same ABI (same-box A/B runs of kernel changes, tools/ab_lib.sh)
_lib = ctypes.CDLL(__import__("os").environ.get("BLADES_HIP_LIB") or str(_build.HIP_LIB))

_P = c_void_p


def _sig(name, argtypes, restype=None):
    fn = getattr(_lib, name)
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


def _opt_sig(name, argtypes, restype=None):
    """A symbol newer than the oldest library ``BLADES_HIP_LIB`` may point at (same-box A/B runs): None if absent."""
    return _sig(name, argtypes, restype) if hasattr(_lib, name) else None


_weighted_col_sum = _sig("blades_weighted_col_sum", [_P, c_long, c_int, c_long, _P, _P, _P])
_col_stats = _sig("blades_col_stats", [_P, c_long, c_int, c_long, _P, _P, _P, _P, _P, c_int, c_float, c_float, _P, _P,
                                       c_int, _P, _P])
_set_rows = _sig("blades_set_rows", [_P, c_long, c_long, _P, c_int, _P, _P])
_row_scale = _sig("blades_row_scale", [_P, c_long, c_int, c_long, _P, _P])
_col_order_stat = _sig("blades_col_order_stat", [_P, c_long, c_int, c_long, _P, c_int, c_int, c_int, _P, _P], c_int)
_row_chunks = _sig("blades_row_chunks", [c_long], c_int)
_row_reduce = _sig("blades_row_reduce", [_P, c_long, c_int, c_long, c_long, _P, c_int, _P, _P, _P, _P, _P])
_server_step = _sig("blades_server_step", [_P, _P, _P, c_long, c_float, c_float, c_int, _P])
_client_sgd = _sig("blades_client_sgd_step", [_P, _P, _P, c_int, c_long, c_long, c_long, c_long, c_float, c_float,
                                              c_int, _P, _P])
_philox_uniform = _sig("blades_philox_uniform", [_P, c_long, c_ulonglong, c_uint, c_ulonglong, _P])
_philox_normal = _sig("blades_philox_normal", [_P, c_long, c_ulonglong, c_uint, c_ulonglong, c_float, c_float, _P])
_normal_rows = _sig("blades_normal_rows", [_P, c_long, c_long, _P, c_int, c_long, c_long, c_ulonglong, c_uint,
                                           c_float, c_float, c_int, _P])
_gram_plan = _sig("blades_gram_plan", [c_int, c_long, _P, _P, _P, _P])
_gram = _sig("blades_gram", [_P, c_long, c_int, c_long, _P, _P, _P])
_augment = _sig("blades_augment_batch", [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P], c_int)


_pcbn_nchunk = _sig("blades_pcbn_nchunk", [c_int, c_int, c_int], c_int)
_pcbn_fwd = _sig("blades_pcbn_forward", [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P, _P, _P, _P, _P, _P, _P,
                                         c_int, c_int, c_int, _P], c_int)
_pcbn_bwd = _sig("blades_pcbn_backward", [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_long, c_long, c_long,
                                          c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, _P], c_int)


_pcbn_bwd2 = _opt_sig("blades_pcbn_backward2", [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_long, c_long,
                                                 c_long, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P], c_int)


def _affine(t, Kc: int, C: int, device, name: str):
    """Affine vector of a per-client normalisation: [C] shared by all clients (stride 0) or [Kc, C] per client."""
    if t is None:
        return None, 0
    t = t.detach()
    if t.dim() == 2:
        if tuple(t.shape) != (Kc, C):
            raise ValueError(f"{name}: expected [{Kc}, {C}], got {tuple(t.shape)}")
        return t.to(device=device, dtype=torch.float32).contiguous(), C
    return _vec(t, C, device, name=name), 0


# Workspaces and outputs of the kernels come from here.  BLADES_POISON=1 fills them with NaN / 0xFF first: a kernel
# that reads a byte nothing wrote then shows up as a non-finite result instead of depending on what the allocator
# handed out (tools/probes/poison_probe.py).
import os as _os

_POISON = _os.environ.get("BLADES_POISON") == "1"


def _poison(t: torch.Tensor) -> torch.Tensor:
    if _POISON and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        else:
            t.view(torch.uint8).fill_(0xFF)
    return t


def _empty(*a, **k):
    return _poison(torch.empty(*a, **k))


def _empty_like(*a, **k):
    return _poison(torch.empty_like(*a, **k))


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk_matrix(U: torch.Tensor):
    if not (U.is_cuda and U.dtype == torch.float32 and U.dim() == 2 and U.stride(1) == 1):
        raise ValueError(f"expected a CUDA fp32 [K, D] matrix with unit column stride, got {U.dtype} {tuple(U.shape)} "
                         f"strides {U.stride()}")
    K, Dl = U.shape
    ld = U.stride(0) if K > 1 else max(U.stride(0), Dl)
    if ld < Dl:
        raise ValueError("row stride smaller than the row length")
    return K, Dl, ld


def _vec(t: Optional[torch.Tensor], n: int, device, dtype=torch.float32, name="vector") -> Optional[torch.Tensor]:
    if t is None:
        return None
    t = t.to(device=device, dtype=dtype).contiguous()
    if t.numel() != n:
        raise ValueError(f"{name}: expected {n} elements, got {t.numel()}")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else c_void_p(0)


_rows_checked = {}  # id(tensor) -> (weakref, version, K, validated device copy)


def _rows(rows: torch.Tensor, K: int, device) -> torch.Tensor:
    """Row-index vector for a kernel, range-checked on the host.  A device tensor is checked once (that costs
    a device->host copy, i.e. a stream sync) and remembered while it is alive and unmodified: the per-round
    callers (the adversaries' ``malicious_rows``) pass the same tensor every round."""
    if rows.is_cuda:
        hit = _rows_checked.get(id(rows))
        if hit is not None and hit[0]() is rows and hit[1] == rows._version and hit[2] == K:
            return hit[3]
    src = rows
    rows = rows.to(device="cpu", dtype=torch.int64).flatten()
    if rows.numel() and (int(rows.min()) < 0 or int(rows.max()) >= K):
        raise IndexError("row index out of range")
    out = rows.to(device).contiguous()
    if src.is_cuda:
        import weakref

        if len(_rows_checked) > 64:
            _rows_checked.clear()
        _rows_checked[id(src)] = (weakref.ref(src), src._version, K, out)
    return out


# ---------------------------------------------------------------------------
def weighted_col_sum(U, w):
    K, Dl, ld = _chk_matrix(U)
    w = _vec(w, K, U.device, name="w")
    out = _empty(Dl, device=U.device, dtype=torch.float32)
    _weighted_col_sum(_ptr(U), ld, K, Dl, _ptr(w), _ptr(out), _stream())
    return out


def _col_stats_call(U, mask, mode, a=1.0, z=0.0, sign=None, mal_rows=None):
    K, Dl, ld = _chk_matrix(U)
    dev = U.device
    mask = _vec(mask, K, dev, name="mask")
    want_std, want_mm, forge = bool(mode & 1), bool(mode & 2), bool(mode & 4)
    mean = _empty(Dl, device=dev)
    std = _empty(Dl, device=dev) if want_std else None
    mn = _empty(Dl, device=dev) if want_mm else None
    mx = _empty(Dl, device=dev) if want_mm else None
    vec = _empty(Dl, device=dev) if forge else None
    sign = _vec(sign, Dl, dev, name="sign")
    n_mal = 0
    if forge and mal_rows is not None and mal_rows.numel():
        mal_rows = _rows(mal_rows, K, dev)
        n_mal = mal_rows.numel()
    else:
        mal_rows = None
    _col_stats(_ptr(U), ld, K, Dl, _ptr(mask), _ptr(mean), _ptr(std), _ptr(mn), _ptr(mx), mode, a, z, _ptr(sign),
               _ptr(mal_rows), n_mal, _ptr(vec), _stream())
    return mean, std, mn, mx, vec


def col_mean_std(U, mask=None):
    mean, std, _, _, _ = _col_stats_call(U, mask, 1)
    return mean, std


def col_min_max(U, mask=None):
    _, _, mn, mx, _ = _col_stats_call(U, mask, 2)
    return mn, mx


def alie_forge_(U, benign_mask, mal_rows, z, sign=None):
    return _col_stats_call(U, benign_mask, 5, 1.0, z, sign, mal_rows)[4]


def ipm_forge_(U, benign_mask, mal_rows, scale):
    return _col_stats_call(U, benign_mask, 4, -scale, 0.0, None, mal_rows)[4]


def set_rows_(U, rows, v):
    K, Dl, ld = _chk_matrix(U)
    rows = _rows(rows, K, U.device)
    v = _vec(v, Dl, U.device, name="v")
    _set_rows(_ptr(U), ld, Dl, _ptr(rows), rows.numel(), _ptr(v), _stream())
    return U


def row_scale_(U, s):
    K, Dl, ld = _chk_matrix(U)
    s = _vec(s, K, U.device, name="s")
    _row_scale(_ptr(U), ld, K, Dl, _ptr(s), _stream())
    return U


def _order_stat_fill(mask, K: int, mode: int):
    """Pad table for the order-statistic kernel: 0 = live row, otherwise the value sorted in its place.

    Median mode alternates +inf / -inf over the dead slots (rows masked out and the rows up to the next
    power of two) so the two medians always sit in the middle of the sorted block; trimmed-mean mode pads
    with +inf only.  The live-row count rides in the extra last slot so no host sync is needed.
    """
    N = 64
    while N < K:
        N *= 2
    dev = mask.device
    dead = torch.ones(N, dtype=torch.bool, device=dev)
    dead[:K] = mask == 0
    inf = torch.full((), float("inf"), device=dev)
    if mode == 0:
        pad = torch.where(torch.cumsum(dead, 0) % 2 == 1, inf, -inf)
    else:
        pad = inf.expand(N)
    fill = _empty(N + 1, device=dev)
    fill[:N] = torch.where(dead, pad, torch.zeros((), device=dev))
    fill[N] = (K - dead[:K].sum()).to(torch.float32)
    return fill


def _col_order_stat_wide(U, mask, mode: int, b: int):
    """More than 2048 rows (beyond the in-LDS selection network of orderstat.hip): the same statistics from a
    device radix sort, one column slab at a time so the sorted copy stays under ~1 GB.  Masked rows sort to the
    top as +inf and the statistic indexes the first ``n`` sorted rows (``n`` stays a device scalar: no sync)."""
    K, Dl = U.shape
    out = _empty(Dl, device=U.device)
    live = None if mask is None else (mask > 0)
    n = torch.full((), K, device=U.device, dtype=torch.long) if live is None else live.sum()
    idx = torch.arange(K, device=U.device).unsqueeze(1)
    step = max(1, (256 << 20) // K)
    for c0 in range(0, Dl, step):
        blk = U[:, c0: c0 + step].float()
        if live is not None:
            blk = torch.where(live.unsqueeze(1), blk, torch.full((), float("inf"), device=U.device))
        srt = torch.sort(blk, dim=0).values
        if mode == 0:  # mean of lower and upper median
            lo, hi = (n - 1) // 2, n // 2
            res = 0.5 * (srt.index_select(0, lo.reshape(1)) + srt.index_select(0, hi.reshape(1))).squeeze(0)
        else:  # mean of sorted rows [b, n - b)
            keep = (idx >= b) & (idx < n - b)
            res = torch.where(keep, srt, torch.zeros((), device=U.device)).sum(0) / (n - 2 * b).clamp_min(1)
        out[c0: c0 + step] = res
    return out


def col_order_stat(U, mask, mode: int, b: int):
    K, Dl, ld = _chk_matrix(U)
    mask = _vec(mask, K, U.device, name="mask")
    if K > 2048:
        if mode == 1 and mask is None and not 0 <= 2 * b < K:
            raise ValueError(f"trimmed mean needs 0 <= 2*b < K, got b={b}, K={K}")
        return _col_order_stat_wide(U, mask, mode, b)
    if mode == 1 and mask is None and not 0 <= 2 * b < K:
        raise ValueError(f"trimmed mean needs 0 <= 2*b < K, got b={b}, K={K}")
    fill = None if mask is None else _order_stat_fill(mask, K, mode)
    out = _empty(Dl, device=U.device)
    rc = _col_order_stat(_ptr(U), ld, K, Dl, _ptr(fill), K, mode, b, _ptr(out), _stream())
    if rc != 0:
        raise ValueError(f"col_order_stat rejected the call (code {rc}): K={K}, ld={ld}, Dl={Dl}, b={b}")
    return out


def _row_reduce_call(U, kind, m=None, valid_cols=None, counts=False):
    K, Dl, ld = _chk_matrix(U)
    dev = U.device
    m = _vec(m, Dl, dev, name="m")
    nch = _row_chunks(Dl)
    if nch > 65535:  # (grid.y of the row kernels; 2^29 columns per rank)
        raise ValueError("row reductions: more than 2^29 columns per rank")
    part = _empty(K * nch, device=dev)
    cpart = _empty(K * nch * 3, device=dev, dtype=torch.int32) if counts else None
    out = _empty(K, device=dev)
    cnt = _empty(K, 3, device=dev, dtype=torch.int64) if counts else None
    _row_reduce(_ptr(U), ld, K, Dl, Dl if valid_cols is None else int(valid_cols), _ptr(m), kind, _ptr(part),
                _ptr(cpart), _ptr(out), _ptr(cnt), _stream())
    return out, cnt


def row_stats(U, valid_cols, counts=True):
    return _row_reduce_call(U, 0, None, valid_cols, counts)


def sqdist_to(U, m):
    return _row_reduce_call(U, 1, m)[0]


def row_dot(U, v):
    return _row_reduce_call(U, 2, v)[0]


def gram(U):
    K, Dl, ld = _chk_matrix(U)
    tile, ntile, nsplit, cps = c_int(), c_int(), c_int(), c_long()
    _gram_plan(K, Dl, ctypes.byref(tile), ctypes.byref(ntile), ctypes.byref(nsplit), ctypes.byref(cps))
    Kp = tile.value * ntile.value
    slabs = _empty(max(nsplit.value, 1) * Kp * Kp, device=U.device)
    out = _empty(K, K, device=U.device)
    _gram(_ptr(U), ld, K, Dl, _ptr(slabs), _ptr(out), _stream())
    return out


_gram_batched = _opt_sig("blades_gram_batched", [_P, c_long, c_long, c_int, c_int, c_long, _P, _P, _P], c_int)
_gather_center = _opt_sig("blades_gather_center", [_P, c_long, c_int, c_long, _P, c_long, c_long, _P, _P], c_int)


def gram_batched(S):
    """``S[b] S[b]^T`` for a contiguous fp32 stack ``[B, K, n]`` with K <= 128 (one-tile kernel of gram.hip, a grid row
    per matrix): ``[B, K, K]``."""
    if not (S.is_cuda and S.dtype == torch.float32 and S.dim() == 3 and S.is_contiguous() and 1 <= S.shape[1] <= 128
            and S.shape[0] >= 1):
        raise ValueError("gram_batched: expected a contiguous CUDA fp32 [B, K <= 128, n] stack")
    nb, K, n = (int(v) for v in S.shape)
    tile, ntile, nsplit, cps = c_int(), c_int(), c_int(), c_long()
    _gram_plan(K, n, ctypes.byref(tile), ctypes.byref(ntile), ctypes.byref(nsplit), ctypes.byref(cps))
    Kp = tile.value * ntile.value
    slabs = _empty(nb * max(nsplit.value, 1) * Kp * Kp, device=S.device)
    out = _empty(nb, K, K, device=S.device)
    if _gram_batched(_ptr(S), n, K * n, nb, K, n, _ptr(slabs), _ptr(out), _stream()) != 0:
        raise ValueError("gram_batched: rejected")
    return out


_sample_distinct = _opt_sig("blades_sample_distinct", [c_long, c_long, c_long, ctypes.c_ulonglong, _P, _P], c_int)


def sample_distinct(D: int, n: int, iters: int, seed: int, device):
    """``iters`` sets of ``n`` distinct column ids in ``[0, D)`` (colgather.hip: keyed Feistel permutation with cycle
    walking; ops/reference.py::sample_distinct gives the same numbers): int64 ``[iters, n]`` on ``device``."""
    out = torch.empty(int(iters), int(n), dtype=torch.int64, device=device)
    if _sample_distinct(int(D), int(n), int(iters), int(seed) & ((1 << 64) - 1), _ptr(out), _stream()) != 0:
        raise ValueError("sample_distinct: need 1 <= n <= D <= 2^40")
    return out


def gather_center_cols(U, idx):
    """``out[i, k, j] = U[k, idx[i, j]] - mean_k U[k, idx[i, j]]`` (colgather.hip): the centred column sub-samples of
    DnC, ``[iters, K, n]`` fp32 contiguous.  ``idx``: int64 ``[iters, n]`` on the device."""
    K, Dl, ld = _chk_matrix(U)
    if not (idx.is_cuda and idx.dtype == torch.int64 and idx.dim() == 2 and idx.numel() > 0):
        raise ValueError("gather_center_cols: idx must be a non-empty CUDA int64 [iters, n] tensor")
    idx = idx.contiguous()
    iters, n = int(idx.shape[0]), int(idx.shape[1])
    out = _empty(iters, K, n, device=U.device)
    if _gather_center(_ptr(U), ld, K, Dl, _ptr(idx), iters, n, _ptr(out), _stream()) != 0:
        raise ValueError("gather_center_cols: rejected")
    return out


_smallk_max = _opt_sig("blades_smallk_max", [], c_int)
_power_top = _opt_sig("blades_power_top", [_P, c_int, c_int, c_int, _P, _P], c_int)
_weiszfeld = _opt_sig("blades_weiszfeld", [_P, _P, c_int, c_int, ctypes.c_double, ctypes.c_double, _P, _P, _P], c_int)
_cclip = _opt_sig("blades_cclip", [_P, _P, _P, c_int, c_int, ctypes.c_double, _P, _P], c_int)
_minmax_bisect = _opt_sig("blades_minmax_bisect", [_P, _P, c_int, _P, _P, c_float, c_float, c_float, _P, _P], c_int)
SMALLK_MAX = int(_smallk_max()) if _smallk_max is not None else 0


def minmax_bisect(a, b, c, thr, lo: float, hi: float, tol: float):
    """Bisection of the min-max attack in one launch: the last midpoint gamma tried in [lo, hi] (halving until the
    interval is <= tol) under ``sqrt(max_i max(a_i - 2 gamma b_i + gamma^2 c, 0)) < thr``.  0-d fp32 tensor."""
    n = int(a.numel())
    a = _vec(a, n, a.device, name="a")
    b = _vec(b, n, a.device, name="b")
    c = c.reshape(1).to(device=a.device, dtype=torch.float32)
    thr = thr.reshape(1).to(device=a.device, dtype=torch.float32)
    out = _empty(1, device=a.device)
    if _minmax_bisect(_ptr(a), _ptr(b), n, _ptr(c), _ptr(thr), float(lo), float(hi), float(tol), _ptr(out), _stream()) != 0:
        raise ValueError("minmax_bisect: rejected")
    return out.reshape(())


def cclip(G, p, mm, n_iter: int, tau: float):
    """Centred-clipping iterations in coefficient space (K <= SMALLK_MAX): fp64 ``[1 + K]`` = (c0, c) with
    ``m_n = c0 m_0 + c^T X``, from ``G = X X^T``, ``p = X m_0`` and ``mm = |m_0|^2`` (a one-element tensor)."""
    K = int(G.shape[0])
    if not (G.is_cuda and G.dtype == torch.float32 and G.dim() == 2 and G.shape[1] == K and 1 <= K <= SMALLK_MAX):
        raise ValueError(f"cclip: expected a CUDA fp32 [K, K] matrix with K <= {SMALLK_MAX}")
    G = G.contiguous()
    p = _vec(p, K, G.device, name="p")
    mm = mm.reshape(1).to(device=G.device, dtype=torch.float32)
    out = torch.empty(1 + K, dtype=torch.float64, device=G.device)
    if _cclip(_ptr(G), _ptr(p), _ptr(mm), K, int(n_iter), float(tau), _ptr(out), _stream()) != 0:
        raise ValueError("cclip: rejected")
    return out


def power_top(G, steps: int):
    """Top eigenvector of a symmetric PSD fp32 ``[K, K]`` matrix (or of each matrix of a ``[B, K, K]`` stack),
    K <= SMALLK_MAX, by ``steps`` normalised power steps in one launch (ops/csrc/smallk.hip)."""
    K = int(G.shape[-1])
    if not (G.is_cuda and G.dtype == torch.float32 and G.dim() in (2, 3) and G.shape[-2] == K and 1 <= K <= SMALLK_MAX
            and G.shape[0] >= 1):
        raise ValueError(f"power_top: expected a CUDA fp32 [K, K] or [B, K, K] matrix with K <= {SMALLK_MAX}")
    G = G.contiguous()
    nb = 1 if G.dim() == 2 else int(G.shape[0])
    out = _empty(K, device=G.device) if G.dim() == 2 else _empty(nb, K, device=G.device)
    if _power_top(_ptr(G), K, nb, int(steps), _ptr(out), _stream()) != 0:
        raise ValueError("power_top: rejected")
    return out


def weiszfeld(G, w0, maxiter: int, eps: float, ftol: float):
    """Weiszfeld iterations of the geometric median in coefficient space on the Gram matrix ``G`` (fp32 ``[K, K]``,
    K <= SMALLK_MAX), fp64 arithmetic, one launch.  Returns (alpha fp32 ``[K]``, iterations run as an int32 tensor)."""
    K = int(G.shape[0])
    if not (G.is_cuda and G.dtype == torch.float32 and G.dim() == 2 and G.shape[1] == K and 1 <= K <= SMALLK_MAX):
        raise ValueError(f"weiszfeld: expected a CUDA fp32 [K, K] matrix with K <= {SMALLK_MAX}")
    G = G.contiguous()
    w0 = _vec(w0, K, G.device, name="w0")
    alpha = _empty(K, device=G.device)
    iters = torch.empty(1, dtype=torch.int32, device=G.device)
    if _weiszfeld(_ptr(G), _ptr(w0), K, int(maxiter), float(eps), float(ftol), _ptr(alpha), _ptr(iters), _stream()) != 0:
        raise ValueError("weiszfeld: rejected")
    return alpha, iters


def server_step_(theta, buf, agg, lr, momentum, first):
    n = theta.numel()
    for t in (theta, agg) + ((buf,) if momentum != 0.0 else ()):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
            raise ValueError("server_step_: need contiguous fp32 CUDA vectors of equal length")
    _server_step(_ptr(theta), _ptr(buf if momentum != 0.0 else theta), _ptr(agg), n, lr, momentum, int(first),
                 _stream())
    return theta


_nonfinite_flag = _opt_sig("blades_nonfinite_flag", [_P, c_long, _P, _P], None)
_server_step_guarded = _opt_sig("blades_server_step_guarded", [_P, _P, _P, c_long, c_float, c_float, c_int, _P, _P, _P], None)


def nonfinite_flag(x):
    """fp32 ``[1]`` tensor: 1 when ``x`` (contiguous fp32) holds an inf / nan, else 0 — decided on the device."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise ValueError("nonfinite_flag: need a contiguous fp32 CUDA tensor")
    flag = torch.empty(1, device=x.device, dtype=torch.float32)
    _nonfinite_flag(_ptr(x), x.numel(), _ptr(flag), _stream())
    return flag


def server_step_guarded_(theta, buf, agg, lr, momentum, first, bad, skipped):
    """``server_step_`` that treats the aggregate as zero when ``bad[0] > 0`` and then adds 1 to ``skipped`` (fp32
    device scalars): the non-finite guard without host sync and without the eight ATen launches of its tensor form."""
    n = theta.numel()
    for t in (theta, agg) + ((buf,) if momentum != 0.0 else ()):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
            raise ValueError("server_step_guarded_: need contiguous fp32 CUDA vectors of equal length")
    for t in (bad, skipped):
        if not (t.is_cuda and t.dtype == torch.float32 and t.numel() == 1):
            raise ValueError("server_step_guarded_: bad / skipped are fp32 CUDA scalars")
    _server_step_guarded(_ptr(theta), _ptr(buf if momentum != 0.0 else theta), _ptr(agg), n, lr, momentum, int(first),
                         _ptr(bad), _ptr(skipped), _stream())
    return theta


def client_sgd_step_(W, G, M, lr, momentum, first, gscale=None):
    Kc, D, ldw = _chk_matrix(W)
    Kg, Dg, ldg = _chk_matrix(G)
    if (Kg, Dg) != (Kc, D):
        raise ValueError("client_sgd_step_: W and G shapes differ")
    ldm = 0
    if momentum != 0.0:
        Km, Dm, ldm = _chk_matrix(M)
        if (Km, Dm) != (Kc, D):
            raise ValueError("client_sgd_step_: momentum buffer shape differs")
    gscale = _vec(gscale, Kc, W.device, name="gscale")
    _client_sgd(_ptr(W), _ptr(G), _ptr(M) if momentum != 0.0 else c_void_p(0), Kc, D, ldw, ldg, ldm, lr, momentum,
                int(first), _ptr(gscale), _stream())
    return W


def philox_uniform(n, seed, stream, offset, device):
    out = _empty(n, device=device, dtype=torch.float32)
    _philox_uniform(_ptr(out), n, seed & (2 ** 64 - 1), stream & 0xFFFFFFFF, offset, _stream())
    return out


def philox_normal(n, seed, stream, offset, mean, std, device):
    out = _empty(n, device=device, dtype=torch.float32)
    _philox_normal(_ptr(out), n, seed & (2 ** 64 - 1), stream & 0xFFFFFFFF, offset, mean, std, _stream())
    return out


_adaptive_forge = _opt_sig("blades_adaptive_forge", [_P, _P, _P, _P, c_long, c_ulonglong, c_uint, c_ulonglong, c_float, _P])
_perm_keys = _opt_sig("blades_perm_keys", [_P, c_long, c_ulonglong, c_uint, c_ulonglong, c_ulonglong, _P])
_signguard_forge = _opt_sig("blades_signguard_forge", [_P, _P, _P, c_long, c_ulonglong, c_uint, c_ulonglong, _P])


def adaptive_forge(mean, mn, mx, seed, stream, offset, b=2.0):
    n = mean.numel()
    for t in (mean, mn, mx):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
            raise ValueError("adaptive_forge: need contiguous fp32 CUDA vectors of equal length")
    out = _empty(n, device=mean.device, dtype=torch.float32)
    _adaptive_forge(_ptr(mean), _ptr(mn), _ptr(mx), _ptr(out), n, seed & (2 ** 64 - 1), stream & 0xFFFFFFFF, offset,
                    float(b), _stream())
    return out


def perm_keys(n, seed, stream, offset, d_valid_end, device):
    out = _empty(n, device=device, dtype=torch.int64)
    _perm_keys(_ptr(out), n, seed & (2 ** 64 - 1), stream & 0xFFFFFFFF, offset, int(d_valid_end), _stream())
    return out


def signguard_forge(keys, thresholds, seed, stream, offset):
    if not (keys.is_cuda and keys.dtype == torch.int64 and keys.is_contiguous() and thresholds.is_cuda
            and thresholds.dtype == torch.int64 and thresholds.numel() == 2 and thresholds.is_contiguous()):
        raise ValueError("signguard_forge: keys int64 [n], thresholds int64 [2], both contiguous CUDA tensors")
    out = _empty(keys.numel(), device=keys.device, dtype=torch.float32)
    _signguard_forge(_ptr(keys), _ptr(thresholds), _ptr(out), keys.numel(), seed & (2 ** 64 - 1), stream & 0xFFFFFFFF,
                     offset, _stream())
    return out


_client_dropout = _sig("blades_client_dropout_mult", [_P, _P, c_int, c_long, _P, c_uint, c_ulonglong, c_float, _P])


def client_dropout_mult(uids, n: int, seed: int, tag, layer: int, p: float):
    if not (uids.is_cuda and uids.dtype == torch.int64 and uids.dim() == 1 and uids.is_contiguous()):
        raise ValueError("client_dropout_mult: uids must be a contiguous CUDA int64 vector")
    if not (tag.is_cuda and tag.dtype == torch.int64 and tag.numel() == 1):
        raise ValueError("client_dropout_mult: tag must be a 1-element CUDA int64 tensor")
    if not 0.0 <= p < 1.0:
        raise ValueError("client_dropout_mult: need 0 <= p < 1")
    out = _empty(uids.numel(), n, device=uids.device, dtype=torch.float32)
    _client_dropout(_ptr(out), _ptr(uids), uids.numel(), n, _ptr(tag), (2 + 256 * layer) & 0xFFFFFFFF,
                    seed & (2 ** 64 - 1), p, _stream())
    return out


_shuffle_keys = _sig("blades_shuffle_keys", [_P, _P, _P, _P, c_int, c_int, c_ulonglong, _P])


def shuffle_keys(uids, epochs, lens, nmax: int, seed: int):
    """[rows, nmax] float64 sort keys of the batch sampler (see rng.hip shuffle_keys_kernel)."""
    rows = uids.numel()
    for t in (uids, epochs, lens):
        if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.numel() == rows):
            raise ValueError("shuffle_keys: uids / epochs / lens must be contiguous CUDA int64 vectors of one length")
    keys = _empty(rows, nmax, device=uids.device, dtype=torch.float64)
    _shuffle_keys(_ptr(keys), _ptr(uids), _ptr(epochs), _ptr(lens), rows, nmax, seed & (2 ** 64 - 1), _stream())
    return keys


def add_normal_rows_(U, rows, seed, stream, col_offset, d_total, mean, std, overwrite):
    K, Dl, ld = _chk_matrix(U)
    rows = rows.to(dtype=torch.int64, device="cpu").reshape(-1, 2)
    if rows.numel() and (int(rows[:, 0].min()) < 0 or int(rows[:, 0].max()) >= K):
        raise IndexError("row index out of range")
    rd = rows.to(U.device).contiguous()
    _normal_rows(_ptr(U), ld, Dl, _ptr(rd), rows.shape[0], col_offset, d_total, seed & (2 ** 64 - 1),
                 stream & 0xFFFFFFFF, mean, std, int(overwrite), _stream())
    return U


def augment_batch(x, params, size, mean, std, validate: bool = True):
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 5):
        raise ValueError("augment_batch: x must be CUDA fp32 [n, B, C, H, W]")
    x = x.contiguous()
    n, B, C, H, W = x.shape
    if params.shape != (n, 9):
        raise ValueError("augment_batch: params must be [n, 9]")
    # validated on the host copy: a table that already lives on the host (the engine's case) costs no device sync
    pc = params.detach().float().cpu() if validate else None
    params = params.to(device=x.device, dtype=torch.float32).contiguous()
    # crop window must lie inside the image (checked on the host: see module docstring)
    if n and validate and (float((pc[:, 0] + pc[:, 2]).max()) > H or float((pc[:, 1] + pc[:, 2]).max()) > W
              or float(pc[:, :3].min()) < 0 or float(pc[:, 2].min()) < 1):
        raise ValueError("augment_batch: crop window out of bounds")
    out = _empty(n, B, C, size, size, device=x.device, dtype=torch.float32)
    m = (c_float * 4)(*(list(mean) + [0.0] * 4)[:4])
    s = (c_float * 4)(*(list(std) + [1.0] * 4)[:4])
    rc = _augment(_ptr(x), _ptr(params), _ptr(out), n, B, C, H, W, size, m, s, _stream())
    if rc != 0:
        raise ValueError("augment_batch supports up to 4 channels")
    return out


# ---------------------------------------------------------------------------
_ACT_DTYPES = (torch.bfloat16, torch.float32)


def _chk_nhwc_bf16(t: torch.Tensor, name: str, dtype=torch.bfloat16):
    if not (t.is_cuda and t.dtype == dtype and t.dim() == 4
            and t.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError(f"{name}: expected a CUDA {dtype} channels-last 4-D tensor, got {t.dtype} {tuple(t.shape)} "
                         f"strides {t.stride()}")


def pcbn_supported(x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype in _ACT_DTYPES and x.dim() == 4 and x.shape[1] % 8 == 0 and x.shape[1] <= 2048
            and 256 % (x.shape[1] // 8) == 0
            and x.is_contiguous(memory_format=torch.channels_last))


def pcbn_forward(x, res, gamma, beta, Kc: int, eps: float, relu: bool, want_mask: bool = False, pre_stats=None):
    """``pre_stats``: ``(partial, nchunk)`` from ``conv_forward(..., stats_B=...)`` — the convolution that produced
    ``x`` already summed it per (client, tile, channel), so the statistics pass over ``x`` is skipped."""
    if x.dtype not in _ACT_DTYPES:
        raise ValueError(f"pcbn_forward: unsupported dtype {x.dtype}")
    _chk_nhwc_bf16(x, "x", x.dtype)
    N, C, H, W = x.shape
    if N % Kc != 0:
        raise ValueError("batch is not a multiple of the number of clients")
    P = (N // Kc) * H * W
    if res is not None:
        _chk_nhwc_bf16(res, "res", x.dtype)
        if res.shape != x.shape:
            raise ValueError("residual shape mismatch")
    dev = x.device
    gamma, gstride = _affine(gamma, Kc, C, dev, "gamma")
    beta, bstride = _affine(beta, Kc, C, dev, "beta")
    if gamma is not None and beta is not None and gstride != bstride:
        raise ValueError("pcbn_forward: gamma and beta must both be shared or both per client")
    pre_nch = 0
    if pre_stats is not None:
        partial, pre_nch = pre_stats
        if not (partial.is_cuda and partial.dtype == torch.float32 and partial.is_contiguous()
                and pre_nch > 0 and partial.numel() == Kc * pre_nch * C * 2):
            raise ValueError("pcbn_forward: pre_stats does not match [Kc, nchunk, C, 2]")
    else:
        nch = _pcbn_nchunk(P, C, x.element_size())
        partial = _empty(Kc * nch * C * 2, device=dev)
    stats = _empty(4, Kc, C, device=dev)
    y = _empty_like(x)
    # ReLU gate for the backward pass: 1 bit per element (8 channels of a pixel = 1 byte) instead of y itself
    mask = _empty(x.numel() // 8, dtype=torch.uint8, device=dev) if (relu and want_mask) else None
    rc = _pcbn_fwd(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), Kc, P, C, eps, int(relu), _ptr(partial),
                   _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), _ptr(y), _ptr(mask), x.element_size(),
                   gstride, int(pre_nch), _stream())
    if rc != 0:
        raise ValueError("pcbn_forward: unsupported shape")
    if want_mask:
        return y, stats[0], stats[1], mask
    return y, stats[0], stats[1]


def pcbn_backward(x, gy, relu_mask, gamma, mean, invstd, Kc: int, coef, U_block, off_gamma: int, off_beta: int,
                  accumulate: bool, want_gres: bool, pre_stats=None):
    """``relu_mask``: the uint8 bitmask returned by ``pcbn_forward(..., want_mask=True)`` (``None`` = no ReLU).
    ``pre_stats = (partial, nchunk)``: the raw backward sums of ``gy`` written by the data-gradient kernel that
    produced it (``conv_dgrad(bn_bwd=...)``): the reduction pass over ``gy`` and ``x`` is skipped."""
    if x.dtype not in _ACT_DTYPES:
        raise ValueError(f"pcbn_backward: unsupported dtype {x.dtype}")
    _chk_nhwc_bf16(x, "x", x.dtype)
    _chk_nhwc_bf16(gy, "gy", x.dtype)
    if gy.shape != x.shape:
        raise ValueError("gy shape mismatch")
    if relu_mask is not None:
        if not (relu_mask.is_cuda and relu_mask.dtype == torch.uint8 and relu_mask.is_contiguous()
                and relu_mask.numel() * 8 == x.numel()):
            raise ValueError("relu_mask must be the contiguous uint8 bitmask of pcbn_forward (numel = x.numel() / 8)")
    N, C, H, W = x.shape
    P = (N // Kc) * H * W
    dev = x.device
    gamma, gstride = _affine(gamma, Kc, C, dev, "gamma")
    ld = 0
    if U_block is not None:
        K, Dl, ld = _chk_matrix(U_block)
        if K != Kc or off_gamma + C > Dl or off_beta + C > Dl or off_gamma < 0 or off_beta < 0:
            raise ValueError("pcbn_backward: update-matrix block does not match")
        coef = _vec(coef, Kc, dev, name="coef")
    if mean.numel() != Kc * C or invstd.numel() != Kc * C:
        raise ValueError("pcbn_backward: stats shape mismatch")
    pre_nch = 0
    if pre_stats is not None and _pcbn_bwd2 is not None and x.dtype == torch.float32:
        partial, pre_nch = pre_stats
        if partial.numel() != Kc * pre_nch * C * 2:
            raise ValueError("pcbn_backward: pre_stats does not match (Kc, nchunk, C)")
    else:
        nch = _pcbn_nchunk(P, C, x.element_size())
        partial = _empty(Kc * nch * C * 2, device=dev)
    k = _empty(3, Kc, C, device=dev)
    dx = _empty_like(x)
    gres = _empty_like(x) if want_gres else None
    if _pcbn_bwd2 is not None:
        rc = _pcbn_bwd2(_ptr(x), _ptr(gy), _ptr(relu_mask), _ptr(gamma), _ptr(mean), _ptr(invstd), Kc, P, C, _ptr(coef),
                        _ptr(U_block), ld, off_gamma, off_beta, int(accumulate), _ptr(partial), _ptr(k[0]), _ptr(k[1]),
                        _ptr(k[2]), _ptr(dx), _ptr(gres), x.element_size(), gstride, int(pre_nch), _stream())
    else:
        rc = _pcbn_bwd(_ptr(x), _ptr(gy), _ptr(relu_mask), _ptr(gamma), _ptr(mean), _ptr(invstd), Kc, P, C, _ptr(coef),
                       _ptr(U_block), ld, off_gamma, off_beta, int(accumulate), _ptr(partial), _ptr(k[0]), _ptr(k[1]),
                       _ptr(k[2]), _ptr(dx), _ptr(gres), x.element_size(), gstride, _stream())
    if rc != 0:
        raise ValueError("pcbn_backward: unsupported shape")
    return dx, gres


# ---------------------------------------------------------------------------
# per-client conv weight gradient
_conv_wgrad = _sig("blades_conv_wgrad", [_P, _P, _P, c_long, c_long, _P, c_int] + [c_int] * 11 + [_P] + [c_int] * 5
                   + [_P, c_int, c_int, c_int, c_int, c_int, c_long, c_int, c_int, c_int, _P], c_int)
_wgrad_plans = {}


def _wgrad_plan(B, H, W, Ho, Wo, ks, stride, pad, device):
    """Per-step staging table of the wgrad kernel (see ops/csrc/conv_wgrad.hip)."""
    import numpy as np

    key = (B, H, W, Ho, Wo, ks, stride, pad, str(device))
    if key in _wgrad_plans:
        return _wgrad_plans[key]
    qpr = -(-Wo // 4)
    Q = B * Ho * qpr
    S = -(-Q // 8)
    q = np.arange(S * 8)
    valid = q < Q
    qq = np.minimum(q, Q - 1)
    b = qq // (Ho * qpr)
    oh = (qq // qpr) % Ho
    ow0 = (qq % qpr) * 4
    ih0 = oh * stride - pad
    gi0 = b * H + ih0
    gy_base = np.where(valid, (b * Ho + oh) * Wo + ow0, -1)
    lo_row = b * H + np.clip(ih0, 0, H - 1)
    hi_row = b * H + np.clip(ih0 + ks - 1, 0, H - 1)
    big = np.iinfo(np.int64).max
    glo = np.where(valid, lo_row, big).reshape(S, 8).min(axis=1)
    ghi = np.where(valid, hi_row, -1).reshape(S, 8).max(axis=1)
    # per step: 8 quads x 4 ints, then row_lo, row_hi, window_lo; one extra all-zero row behind the last step
    # (the kernel's lanes that have nothing to fetch read 16 zero bytes from there)
    table = np.zeros((S + 1, 48), dtype=np.int32)
    # Rows a step brings in: everything it needs that is not resident yet, and — when that is less than one
    # staging pass (32 pixels) — the next rows in sequence as well.  The look-ahead evens out the image
    # boundaries (the last step of an image needs no new row, the first step of the next one needs two or
    # three), so stride-1 layers never need more than one pass per step.
    rpp = max(1, 32 // W)
    last_row = B * H - 1
    lo_a, hi_a = np.zeros(S, dtype=np.int64), np.zeros(S, dtype=np.int64)
    prev = -1
    max_new = 0
    for s in range(S):
        lo = max(prev + 1, int(glo[s])) if s == 0 else prev + 1
        hi = max(int(ghi[s]), min(prev + rpp, last_row))
        lo_a[s], hi_a[s] = lo, hi
        if s > 0:  # step 0 always starts a block: its rows come in through the synchronous window load
            max_new = max(max_new, (hi - lo + 1) * W)
        prev = max(prev, hi)
    window = 1
    for s in range(S):
        table[s, 32], table[s, 33] = lo_a[s], hi_a[s]
        table[s, 34] = int(glo[s])
        nxt = int(hi_a[s + 1]) if s + 1 < S else int(hi_a[s])
        window = max(window, nxt - int(glo[s]) + 1)
    table[:S, 0:32:4] = gy_base.reshape(S, 8)
    table[:S, 1:32:4] = gi0.reshape(S, 8)
    table[:S, 2:32:4] = ih0.reshape(S, 8)
    table[:S, 3:32:4] = ow0.reshape(S, 8)
    nlines = 1
    while nlines < window:
        nlines *= 2
    # slots per line: padded width (+3 so the last quad of a ragged row stays inside the line),
    # made = 4 (mod 8) so lines sit 128 B (mod 256 B) apart -> adjacent-row quads do not bank-conflict
    need = W + 2 * pad + (3 if (W % 4 or _os.environ.get("BLADES_WGRAD_LINEPAD") == "3") else 0)
    if stride == 1 and ks == 3:
        # the stride-1 kernels read a row's tap columns as two quads per output quad: slots ow0 .. ow0 + 7
        need = max(need, 4 * ((Wo + 3) // 4) + 4)
    if stride == 2:
        need = 2 * (-(-(W + 2 * pad) // 2) + 2)
    wslots = need
    while wslots % 8 != 4:
        wslots += 1
    passes = max(1, -(-max_new // 32))
    # Is the new-row range arithmetic?  hi[s] = min((s+1)*R + off, last), lo[s] = min(s*R + off, last) + 1 for s > 0.
    rows_r, rows_off = 0, 0
    if S > 1:
        R = int(hi_a[1] - hi_a[0]) if S > 1 else 0
        off = int(hi_a[0]) - R
        if R > 0:
            sidx = np.arange(S)
            hi_f = np.minimum((sidx + 1) * R + off, last_row)
            lo_f = np.minimum(sidx * R + off, last_row) + 1
            if np.array_equal(hi_f[1:], hi_a[1:]) and np.array_equal(lo_f[1:], lo_a[1:]):
                rows_r, rows_off = R, off
    plan = dict(table=torch.from_numpy(table).to(device), nsteps=S, nlines=nlines, wslots=wslots, passes=passes,
                rows_r=rows_r, rows_off=rows_off)
    _wgrad_plans[key] = plan
    return plan


_scratch_buf = {}


def _scratch(n: int, device) -> torch.Tensor:
    """Grow-only fp32 workspace per (device, stream): reuse is stream-ordered, so chunks that run on different
    streams must not share a slab."""
    if torch.cuda.is_current_stream_capturing():
        # inside a HIP-graph capture the slab is an ordinary temporary of that graph's memory pool: no graph ever
        # reads or writes memory that a module-global handed over from another graph's pool
        return _empty(int(n), device=device, dtype=torch.float32)
    key = (str(device), int(torch.cuda.current_stream().cuda_stream))
    buf = _scratch_buf.get(key)
    if buf is None or buf.numel() < n:
        buf = _empty(int(n * 1.25) + 1024, device=device, dtype=torch.float32)
        _scratch_buf[key] = buf
    return buf


def conv_wgrad_supported(x, gy, ks, stride, pad, dilation) -> bool:
    if not (x.is_cuda and x.dtype in _ACT_DTYPES and gy.dtype == x.dtype and x.dim() == 4):
        return False
    Cin, Cout = x.shape[1], gy.shape[1]
    if (Cin % 16 and Cin > 16) or Cout % 16 or ks[0] != ks[1] or ks[0] not in (1, 3) or stride[0] != stride[1] \
            or stride[0] not in (1, 2) or pad[0] != pad[1] or tuple(dilation) != (1, 1):
        return False
    if stride[0] == 2 and (x.shape[3] + 2 * pad[0]) % 2:
        return False
    return True


def conv_wgrad_(x, gy, Kc: int, U_block, off: int, coef, accumulate: bool, ks: int, stride: int, pad: int,
                npl: int = 0, cin_store: int = 0):
    """Writes coef[c] * dW_c into U_block[c, off : off + Cout*Cin*ks*ks] for every client c.
    bf16 activations run as they are; fp32 activations are split in-kernel into ``npl`` bf16 planes
    (3 = fp32 grade, 2 = "bf16x3").  ``cin_store`` < Cin: ``x`` carries zero-padded channels; only the first
    ``cin_store`` input channels are stored (row layout ``[Cout][cin_store][ks][ks]``)."""
    if x.dtype == torch.bfloat16:
        npl = 1
    elif npl not in (2, 3):
        npl = 3
    _chk_nhwc_bf16(x, "x", x.dtype)
    _chk_nhwc_bf16(gy, "gy", x.dtype)
    N, Cin, H, W = x.shape
    N2, Cout, Ho, Wo = gy.shape
    if N != N2 or N % Kc:
        raise ValueError("conv_wgrad_: batch mismatch")
    B = N // Kc
    if Ho != (H + 2 * pad - ks) // stride + 1 or Wo != (W + 2 * pad - ks) // stride + 1:
        raise ValueError("conv_wgrad_: output size does not match the convolution geometry")
    K, Dl, ld = _chk_matrix(U_block)
    if cin_store and not (0 < cin_store <= Cin and Cin % 16 == 0):
        raise ValueError("conv_wgrad_: cin_store needs an input padded to a multiple of 16 channels")
    cin_store = cin_store or Cin
    numel = Cout * cin_store * ks * ks
    if K != Kc or off < 0 or off + numel > Dl:
        raise ValueError("conv_wgrad_: update-matrix block does not match")
    coef = _vec(coef, Kc, x.device, name="coef")
    if Cin % 16:
        # the stem (3 input channels): zero-pad to one 16-channel MFMA block; only the real channels are stored
        xp = x.new_zeros((N, H, W, 16)).permute(0, 3, 1, 2)  # channels-last storage
        xp[:, :Cin] = x
        x, Cin = xp, 16
    x_client_stride = B * H * W * Cin
    rmul, cmul = W, 1
    if npl > 1 and ks == 3 and stride == 2 and pad == 1 and H % 2 == 0 and W % 2 == 0:
        # fp32 planes: the split images of a strided layer's input window do not fit in LDS.  The layer is posed as
        # four stride-1 problems, one per parity class x[:, p::2, q::2] of the input (addressed in place), in one
        # grid; each class computes only the taps it owns (4 + 2 + 2 + 1 = 9) — see blades_conv_wgrad.
        plan = _wgrad_plan(B, Ho, Wo, Ho, Wo, 3, 1, 1, x.device)
        if plan["passes"] > 1:
            return False
        # 128 output channels per workgroup where they exist (conv_wgrad8m_kernel): a class has a quarter of a
        # stride-1 layer's matrix work per staged step, the wider tile doubles it (BLADES_WGRAD_S2_TMT=4: 64)
        tmt = 8 if _os.environ.get("BLADES_WGRAD_S2_TMT", "8") == "8" else 4
        while tmt > 1 and Cout % (tmt * 16):
            tmt //= 2
        blocks = 4 * Kc * (Cout // (tmt * 16)) * (-(-Cin // 64))
        slots, bw = int(_os.environ.get("BLADES_WGRAD_SLOTS", 256)), 4e12
        t_step = (3.2e-6 if npl == 3 else 1.8e-6) * 0.45  # 9/4 taps per block on average + per-step overhead
        if tmt == 8:
            t_step *= 1.4
        best, split = None, 1
        for sp in range(1, max(1, min(16, plan["nsteps"] // 16)) + 1):
            waves = -(-blocks * sp // slots)
            est = waves / sp * plan["nsteps"] * t_step + (0.0 if sp == 1 else (sp + 1) * Kc * numel * 4 / bw)
            if best is None or est < best * 0.97:
                best, split = est, sp
        scratch = _scratch(split * Kc * numel, x.device) if split > 1 else None
        rc = _conv_wgrad(_ptr(x), _ptr(gy), _ptr(U_block), ld, off, _ptr(coef), int(accumulate), Kc, B, Ho, Wo, Cin, Ho,
                         Wo, Cout, 3, 1, 1, _ptr(plan["table"]), plan["nsteps"], plan["nlines"], plan["wslots"], tmt,
                         split, _ptr(scratch), 2 * W, 2, plan["passes"], plan["rows_r"], plan["rows_off"],
                         x_client_stride, cin_store, npl, 1, _stream())
        if rc == 3:
            return False
        if rc != 0:
            raise ValueError(f"conv_wgrad_: strided layer as parity classes rejected (rc={rc})")
        return True
    if ks == 1 and pad == 0 and stride > 1:
        # a strided 1x1 conv only ever touches x[:, ::s, ::s]: run it as a stride-1 1x1 conv on that
        # virtual sub-image (addressed in place, nothing is copied)
        rmul, cmul = stride * W, stride
        H, W, stride = Ho, Wo, 1
    plan = _wgrad_plan(B, H, W, Ho, Wo, ks, stride, pad, x.device)
    if plan["passes"] > 4:
        return False
    tmt = 4
    n_ci_tiles = -(-Cin // 64)
    while tmt > 1 and Cout % (tmt * 16):
        tmt //= 2
    # 256 CUs x 2 resident blocks = 512 slots.  A grid that is not a multiple of that leaves a ragged last
    # wave of blocks (600 blocks = one full wave + one 17 %-full wave that takes just as long), and a small
    # grid does not fill the chip at all: split the pixel reduction over `split` blocks per tile so the grid
    # comes out near a multiple of the slot count.  Partial sums go through a scratch slab and are added in
    # fixed order, so the result stays deterministic.  Cost model: waves(split) / split of the full-K block
    # time plus the slab's write + read traffic.
    blocks = Kc * (Cout // (tmt * 16)) * n_ci_tiles
    slots, t_step, bw = 512, 1.5e-6, 4e12
    if npl > 1:  # split planes: one resident block per CU (the plane images fill LDS), 3 / 6 MFMA products per step
        slots, t_step = 256, (3.2e-6 if npl == 3 else 1.8e-6)
        slots = int(_os.environ.get("BLADES_WGRAD_SLOTS", slots))
    best, split = None, 1
    for sp in range(1, max(1, min(16, plan["nsteps"] // 16)) + 1):
        waves = -(-blocks * sp // slots)
        est = waves / sp * plan["nsteps"] * t_step + (0.0 if sp == 1 else (sp + 1) * Kc * numel * 4 / bw)
        if best is None or est < best * 0.97:  # prefer the smaller split unless the gain is real
            best, split = est, sp
    scratch = None
    if split > 1:
        scratch = _scratch(split * Kc * numel, x.device)
    rc = _conv_wgrad(_ptr(x), _ptr(gy), _ptr(U_block), ld, off, _ptr(coef), int(accumulate), Kc, B, H, W, Cin, Ho,
                     Wo, Cout, ks, stride, pad, _ptr(plan["table"]), plan["nsteps"], plan["nlines"], plan["wslots"],
                     tmt, split, _ptr(scratch), rmul, cmul, plan["passes"], plan["rows_r"], plan["rows_off"], x_client_stride,
                     cin_store, npl, 0, _stream())
    if rc == 3:  # staging does not fit in LDS
        return False
    if rc != 0:
        raise ValueError(f"conv_wgrad_: unsupported configuration (rc={rc})")
    return True


# ---------------------------------------------------------------------------
# per-client LayerNorm (ops/csrc/layernorm.hip)
_ln_nchunk = _sig("blades_ln_nchunk", [c_int], c_int)
_ln_fwd = _sig("blades_ln_forward", [_P, _P, _P, c_long, c_int, c_float, _P, _P, _P, c_int, _P], c_int)
_ln_bwd = _sig("blades_ln_backward", [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_long, c_long, c_long, c_int, _P,
                                      _P, c_int, _P], c_int)


def ln_supported(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dtype in _ACT_DTYPES and x.dim() >= 2 and x.shape[-1] <= 1024 and x.numel() > 0


_ln_fwd2 = _opt_sig("blades_ln_forward2", [_P, _P, _P, c_long, c_int, c_float, _P, _P, _P, c_int, c_long, c_long, c_int,
                                           _P], c_int)
_ln_bwd2 = _opt_sig("blades_ln_backward2", [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_long, c_long, c_long, c_int,
                                            _P, _P, c_int, c_long, _P], c_int)


def _ln_affine(t, C: int, device, name: str):
    """[C] shared by all rows (stride 0), or [Kc, C] — one vector per client, any client stride (rows of the engine's
    per-client weight matrix are used in place)."""
    if t is None:
        return None, 0, 0
    t = t.detach()
    if t.dim() == 2:
        if t.shape[1] != C or t.dtype != torch.float32 or not t.is_cuda or t.stride(1) != 1:
            t = t.to(device=device, dtype=torch.float32).contiguous()
        return t, int(t.stride(0)) if t.shape[0] > 1 else C, int(t.shape[0])
    return _vec(t, C, device, name=name), 0, 0


def ln_forward(x, gamma, beta, eps: float):
    """LayerNorm over the last dimension.  ``gamma`` / ``beta``: ``[C]``, or ``[Kc, C]`` per client (rows of ``x`` are
    client-major).  Returns (y, mean [rows], rstd [rows])."""
    if not ln_supported(x):
        raise ValueError(f"ln_forward: unsupported input {x.dtype} {tuple(x.shape)}")
    x = x.contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    gamma, gs, gk = _ln_affine(gamma, C, x.device, "gamma")
    beta, bs, bk = _ln_affine(beta, C, x.device, "beta")
    Kc = max(gk, bk)
    if Kc and (rows % Kc or (gk and bk and gk != bk)):
        raise ValueError("ln_forward: per-client affine vectors do not match the rows")
    y = _empty_like(x)
    stats = _empty(2, rows, device=x.device, dtype=torch.float32)
    if Kc and _ln_fwd2 is None:
        raise RuntimeError("ln_forward: this library build has no per-client affine mode")
    if _ln_fwd2 is not None:
        rc = _ln_fwd2(_ptr(x), _ptr(gamma), _ptr(beta), rows, C, eps, _ptr(y), _ptr(stats[0]), _ptr(stats[1]),
                      x.element_size(), gs, bs, rows // Kc if Kc else 1, _stream())
    else:
        rc = _ln_fwd(_ptr(x), _ptr(gamma), _ptr(beta), rows, C, eps, _ptr(y), _ptr(stats[0]), _ptr(stats[1]),
                     x.element_size(), _stream())
    if rc != 0:
        raise ValueError("ln_forward: unsupported shape")
    return y, stats[0], stats[1]


def ln_backward(x, gy, gamma, mean, rstd, Kc: int, coef, U_block, off_gamma: int, off_beta: int, accumulate: bool):
    """dx of LayerNorm; coef[c] * (dgamma_c, dbeta_c) go to U_block[c, off_gamma / off_beta : + C] (an offset < 0
    skips that gradient).  Rows of ``x`` are client-major: rows / Kc per client.  ``gamma``: [C] or [Kc, C]."""
    if not (ln_supported(x) and gy.dtype == x.dtype and gy.shape == x.shape):
        raise ValueError("ln_backward: gy must match x")
    x, gy = x.contiguous(), gy.contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    if rows % Kc or mean.numel() != rows or rstd.numel() != rows:
        raise ValueError("ln_backward: row count does not match")
    R = rows // Kc
    gamma, gs, gk = _ln_affine(gamma, C, x.device, "gamma")
    if gk and gk != Kc:
        raise ValueError("ln_backward: per-client gamma does not match the number of clients")
    ld = 0
    if U_block is not None:
        K, Dl, ld = _chk_matrix(U_block)
        if K != Kc or max(off_gamma, off_beta) + C > Dl:
            raise ValueError("ln_backward: update-matrix block does not match")
        coef = _vec(coef, Kc, x.device, name="coef")
    partial = _empty(Kc * _ln_nchunk(R) * C * 2, device=x.device, dtype=torch.float32)
    dx = _empty_like(x)
    if gk and _ln_bwd2 is None:
        raise RuntimeError("ln_backward: this library build has no per-client affine mode")
    if _ln_bwd2 is not None:
        rc = _ln_bwd2(_ptr(x), _ptr(gy), _ptr(gamma), _ptr(mean), _ptr(rstd), Kc, R, C, _ptr(coef), _ptr(U_block), ld,
                      off_gamma, off_beta, int(accumulate), _ptr(partial), _ptr(dx), x.element_size(), gs, _stream())
    else:
        rc = _ln_bwd(_ptr(x), _ptr(gy), _ptr(gamma), _ptr(mean), _ptr(rstd), Kc, R, C, _ptr(coef), _ptr(U_block), ld,
                     off_gamma, off_beta, int(accumulate), _ptr(partial), _ptr(dx), x.element_size(), _stream())
    if rc != 0:
        raise ValueError("ln_backward: unsupported shape")
    return dx


# ---------------------------------------------------------------------------
# fused small-sequence attention (ops/csrc/mha.hip)
_mha_fwd = _sig("blades_mha_forward", [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_uint,
                                       c_ulonglong, c_int, _P], c_int)
_mha_bwd = _sig("blades_mha_backward", [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float,
                                        c_uint, c_ulonglong, c_int, _P], c_int)


def mha_supported(qkv: torch.Tensor) -> bool:
    """qkv [R, N, 3, H, 64] (the fused projection's output), N <= 64."""
    return (qkv.is_cuda and qkv.dtype in _ACT_DTYPES and qkv.dim() == 5 and qkv.shape[2] == 3 and qkv.shape[4] == 64
            and 1 <= qkv.shape[1] <= 64 and qkv.shape[0] > 0)


def _mha_drop(qkv, B, uids, tag, p):
    R = qkv.shape[0]
    if uids is None or p <= 0.0:
        return None, None, 0.0
    if not (uids.is_cuda and uids.dtype == torch.int64 and uids.is_contiguous() and R % B == 0
            and uids.numel() == R // B):
        raise ValueError("mha: uids must be a contiguous CUDA int64 vector with one entry per client")
    if not (tag.is_cuda and tag.dtype == torch.int64 and tag.numel() == 1):
        raise ValueError("mha: tag must be a 1-element CUDA int64 tensor")
    return uids, tag, float(p)


def mha_forward(qkv, B: int, scale: float, uids=None, tag=None, seed: int = 0, layer: int = 0, p: float = 0.0,
                npl: int = 0):
    """softmax(q k^T * scale) -> client-keyed dropout -> @ v for every (sample, head).  Returns
    (o [R, N, H*64], lse [R, H, N])."""
    if not mha_supported(qkv):
        raise ValueError(f"mha_forward: unsupported input {qkv.dtype} {tuple(qkv.shape)}")
    qkv = qkv.contiguous()
    R, N, _, H, _ = qkv.shape
    npl = 1 if qkv.dtype == torch.bfloat16 else (npl if npl in (2, 3) else 3)
    uids, tag, p = _mha_drop(qkv, B, uids, tag, p)
    o = _empty(R, N, H * 64, device=qkv.device, dtype=qkv.dtype)
    lse = _empty(R, H, N, device=qkv.device, dtype=torch.float32)
    rc = _mha_fwd(_ptr(qkv), _ptr(o), _ptr(lse), _ptr(uids), _ptr(tag), R, B if uids is not None else 1, N, H,
                  float(scale), p, (2 + 256 * layer) & 0xFFFFFFFF, seed & (2 ** 64 - 1), npl, _stream())
    if rc != 0:
        raise ValueError("mha_forward: rejected")
    return o, lse


def mha_backward(qkv, o, d_o, lse, B: int, scale: float, uids=None, tag=None, seed: int = 0, layer: int = 0,
                 p: float = 0.0, npl: int = 0):
    qkv, o, d_o = qkv.contiguous(), o.contiguous(), d_o.contiguous()
    R, N, _, H, _ = qkv.shape
    if d_o.dtype != qkv.dtype or o.dtype != qkv.dtype or tuple(d_o.shape) != (R, N, H * 64) or o.shape != d_o.shape:
        raise ValueError("mha_backward: d_o / o do not match qkv")
    npl = 1 if qkv.dtype == torch.bfloat16 else (npl if npl in (2, 3) else 3)
    uids, tag, p = _mha_drop(qkv, B, uids, tag, p)
    dqkv = _empty_like(qkv)
    rc = _mha_bwd(_ptr(qkv), _ptr(o), _ptr(d_o), _ptr(lse), _ptr(dqkv), _ptr(uids), _ptr(tag), R,
                  B if uids is not None else 1, N, H, float(scale), p, (2 + 256 * layer) & 0xFFFFFFFF,
                  seed & (2 ** 64 - 1), npl, _stream())
    if rc != 0:
        raise ValueError("mha_backward: rejected")
    return dqkv


# ---------------------------------------------------------------------------
# per-client Linear weight gradient (ops/csrc/linear_wgrad.hip)
_linear_wgrad = _sig("blades_linear_wgrad", [_P, _P, _P, c_long, c_long, _P, c_int, c_int, c_int, c_int, c_int, c_int,
                                             _P], c_int)


def linear_wgrad_(x, gy, Kc: int, U_block, off: int, coef, accumulate: bool, npl: int = 3):
    """U_block[c, off : off + Out*In] (+)= coef[c] * gy_c^T x_c for every client c; ``x`` is ``[Kc*R, In]``,
    ``gy`` ``[Kc*R, Out]`` (bf16, or fp32 split in-kernel into ``npl`` bf16 planes)."""
    if not (x.is_cuda and gy.is_cuda and x.dim() == 2 and gy.dim() == 2 and x.dtype == gy.dtype
            and x.dtype in _ACT_DTYPES):
        raise ValueError(f"linear_wgrad_: expected 2-D CUDA bf16/fp32 tensors of one dtype, got {x.dtype} "
                         f"{tuple(x.shape)} / {gy.dtype} {tuple(gy.shape)}")
    if x.shape[0] != gy.shape[0] or x.shape[0] % Kc:
        raise ValueError("linear_wgrad_: row counts do not match the number of clients")
    x = x if x.is_contiguous() and x.data_ptr() % 16 == 0 else x.contiguous().clone()
    gy = gy if gy.is_contiguous() and gy.data_ptr() % 16 == 0 else gy.contiguous().clone()
    R, In, Out = x.shape[0] // Kc, x.shape[1], gy.shape[1]
    K, Dl, ld = _chk_matrix(U_block)
    if K != Kc or off < 0 or off + Out * In > Dl:
        raise ValueError("linear_wgrad_: update-matrix block does not match")
    coef = _vec(coef, Kc, x.device, name="coef")
    npl = 1 if x.dtype == torch.bfloat16 else (npl if npl in (2, 3) else 3)
    rc = _linear_wgrad(_ptr(x), _ptr(gy), _ptr(U_block), ld, off, _ptr(coef), int(accumulate), Kc, R, In, Out, npl,
                       _stream())
    if rc != 0:
        raise ValueError(f"linear_wgrad_: unsupported configuration (rc={rc})")
    return True


# ---------------------------------------------------------------------------
# shared-weight convolution: forward and data gradient (ops/csrc/conv_igemm.hip)
_conv_igemm_ws = _sig("blades_conv_igemm_ws", [c_int] * 7, c_long)
_conv_igemm = _sig("blades_conv_igemm", [_P, _P, _P, _P, _P] + [c_int] * 14 + [c_long, c_long, _P, c_int, _P], c_int)
_conv_igemm_stats_nchunk = _sig("blades_conv_igemm_stats_nchunk", [c_int] * 14, c_int)

#: operand planes per compute dtype: fp32 tensors are split exactly into 3 bf16 planes (6 MFMA products, fp32
#: grade), "bf16x3" keeps 2 planes (3 products, ~2^-17), bf16 tensors run as they are
CONV_PLANES = {"fp32": 3, "bf16x3": 2, "bf16": 1}


def _chk_nhwc(t: torch.Tensor, name: str, dtype):
    if not (t.is_cuda and t.dtype == dtype and t.dim() == 4):
        raise ValueError(f"{name}: expected a CUDA {dtype} 4-D tensor, got {t.dtype} {tuple(t.shape)}")
    N, C, H, W = t.shape
    if t.numel() and t.stride() != (H * W * C, 1, W * C, C) and not (
            t.is_contiguous(memory_format=torch.channels_last) and (C == 1 or H * W == 1)):
        raise ValueError(f"{name}: expected channels-last strides, got {t.stride()} for shape {tuple(t.shape)}")


def conv_igemm_supported(x: torch.Tensor, weight: torch.Tensor, stride, pad, dilation, npl: int) -> bool:
    want = torch.bfloat16 if npl == 1 else torch.float32
    kh, kw = weight.shape[-2:]
    return (x.is_cuda and x.dtype == want and x.dim() == 4 and weight.dtype == torch.float32
            and stride[0] == stride[1] and pad[0] == pad[1] and tuple(dilation) == (1, 1) and kh * kw <= 25
            and x.shape[3] + 2 * pad[1] >= kw and x.shape[3] <= 256)


_conv_igemm2 = _opt_sig("blades_conv_igemm2", [_P, _P, _P, _P, _P] + [c_int] * 14 + [c_long, c_long, _P, c_int, c_int, _P],
                        c_int)
_conv_igemm3 = _opt_sig("blades_conv_igemm3", [_P, _P, _P, _P, _P] + [c_int] * 14 + [c_long, c_long, _P, c_int, c_int, _P, _P, _P],
                        c_int)
_conv_dgrad_stats_nchunk = _opt_sig("blades_conv_dgrad_stats_nchunk", [c_int] * 14, c_int)
_conv_wprep = _opt_sig("blades_conv_wprep", [_P, _P] + [c_int] * 7 + [c_long, _P], c_int)


def conv_planes(weight, mode: int, npl: int, out=None):
    """bf16 operand planes of a shared convolution weight ``[Co, Ci, kh, kw]`` for ``conv_forward`` (mode 0) /
    ``conv_dgrad`` (mode 1): the split / transposed / padded image the kernel stages its weight tiles from.  Built
    once per layer and round by the FedSGD engine (the weights do not change while the clients train) instead of
    once per chunk and call.  ``out``: a buffer of a previous call for the same weight shape, reused in place."""
    w = weight.detach()
    if not (w.is_cuda and w.dtype == torch.float32 and w.dim() == 4 and w.is_contiguous()):
        raise ValueError("conv_planes: weight must be a contiguous CUDA fp32 [Co, Ci, kh, kw] tensor")
    if _conv_wprep is None:
        return None
    Co, Ci, kh, kw = (int(v) for v in w.shape)
    n = _conv_igemm_ws(Co, Ci, kh, kw, mode, npl, 1)
    if out is None or out.numel() != n or out.device != w.device:
        out = _empty(n, dtype=torch.uint8, device=w.device)
    rc = _conv_wprep(_ptr(w), _ptr(out), Co, Ci, kh, kw, mode, npl, 1, 0, _stream())
    if rc != 0:
        raise ValueError(f"conv_planes: rejected (rc={rc})")
    return out


def _conv_call(inp, weight, bias, out, N, H, W, Ci, Ho, Wo, Co, kh, kw, stride, pad, mode, npl, stats_B=0, planes=None,
               accumulate=False, bn_bwd=None):
    """``weight``: [Co, Ci, kh, kw] shared by every image, or [Kc, Co, Ci, kh, kw] — one weight set per client
    (clients may be rows of a wider matrix: any stride along dim 0, contiguous inside a client).  ``planes``: the
    ``conv_planes`` image of a shared weight (the weight itself is then not read)."""
    w = weight.detach()
    nc, wstride, bstride = 1, 0, 0
    if w.dim() == 5:
        nc, wstride = int(w.shape[0]), int(w.stride(0))
        ok = tuple(w.shape[1:]) == (Co, Ci, kh, kw) and w[0].is_contiguous() and N % nc == 0
    else:
        ok = tuple(w.shape) == (Co, Ci, kh, kw) and w.is_contiguous()
    if not (ok and w.is_cuda and w.dtype == torch.float32):
        raise ValueError("conv_igemm: weight must be a CUDA fp32 [Co, Ci, kh, kw] or [Kc, Co, Ci, kh, kw] tensor, "
                         f"contiguous per client; got {tuple(w.shape)} strides {w.stride()}")
    if bias is not None:
        b = bias.detach()
        if nc > 1:
            if not (b.dim() == 2 and tuple(b.shape) == (nc, Co) and b.stride(1) == 1 and b.dtype == torch.float32):
                raise ValueError("conv_igemm: per-client bias must be fp32 [Kc, Co] with unit inner stride")
            bstride, bias = int(b.stride(0)), b
        else:
            bias = _vec(b, Co, inp.device, name="bias")
    need = _conv_igemm_ws(Co, Ci, kh, kw, mode, npl, nc)
    flags = 0
    if planes is not None and nc == 1 and _conv_igemm2 is not None:
        if not (planes.is_cuda and planes.dtype == torch.uint8 and planes.is_contiguous() and planes.numel() == need):
            raise ValueError("conv_igemm: `planes` is not the conv_planes image of this weight shape / mode")
        ws, flags = planes, 1
    else:
        ws = _empty(need, dtype=torch.uint8, device=inp.device)
    if accumulate:
        if _conv_igemm2 is None:
            raise RuntimeError("conv_igemm: this library build has no accumulate mode")
        flags |= 2
    partial, nch = None, 0
    if stats_B and mode == 0 and npl > 1:
        # BatchNorm partial sums from the epilogue, when no tile of the planned launch straddles two clients
        nch = _conv_igemm_stats_nchunk(N, H, W, Ci, Ho, Wo, Co, kh, kw, stride, pad, npl, nc, int(stats_B))
        if nch > 0:
            partial = _empty((N // int(stats_B)) * nch * Co * 2, device=inp.device)
    if bn_bwd is not None and mode == 1 and npl > 1 and _conv_igemm3 is not None:
        # first pass of the BatchNorm backward that consumes this data gradient, from the kernel's epilogue
        bx, bmask, bB = bn_bwd
        nch = _conv_dgrad_stats_nchunk(N, H, W, Ci, Ho, Wo, Co, kh, kw, stride, pad, npl, nc, int(bB))
        if nch > 0:
            partial = _empty((N // int(bB)) * nch * Ci * 2, device=inp.device)
            rc = _conv_igemm3(_ptr(inp), _ptr(w), _ptr(bias), _ptr(out), _ptr(ws), N, H, W, Ci, Ho, Wo, Co, kh, kw,
                              stride, pad, mode, npl, nc, wstride, bstride, _ptr(partial), int(bB), flags, _ptr(bx),
                              _ptr(bmask), _stream())
            if rc != 0:
                raise ValueError(f"conv_igemm: data gradient with BatchNorm sums rejected (rc={rc})")
            return partial, nch
    if _conv_igemm2 is not None:
        rc = _conv_igemm2(_ptr(inp), _ptr(w), _ptr(bias), _ptr(out), _ptr(ws), N, H, W, Ci, Ho, Wo, Co, kh, kw, stride,
                          pad, mode, npl, nc, wstride, bstride, _ptr(partial),
                          int(stats_B) if partial is not None else 0, flags, _stream())
    else:
        rc = _conv_igemm(_ptr(inp), _ptr(w), _ptr(bias), _ptr(out), _ptr(ws), N, H, W, Ci, Ho, Wo, Co, kh, kw, stride,
                         pad, mode, npl, nc, wstride, bstride, _ptr(partial),
                         int(stats_B) if partial is not None else 0, _stream())
    if rc != 0:
        raise ValueError(f"conv_igemm: unsupported configuration (rc={rc}): N={N} H={H} W={W} Ci={Ci} Co={Co} "
                         f"k={kh}x{kw} stride={stride} pad={pad} mode={mode} npl={npl}")
    return (partial, nch) if partial is not None else None


_stem_im2col_c = _sig("blades_stem_im2col", [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P], c_int)
_stem_im2col_w = _opt_sig("blades_stem_im2col_w", [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P], c_int)


def stem_width(Ci: int, k: int) -> int:
    """Channels per pixel of the stem's im2col tensor: 16 where ``Ci*k*k`` fits (single-channel images: half the bytes
    of the layer's three passes over that tensor), else 32."""
    return 16 if Ci * k * k <= 16 and _stem_im2col_w is not None else 32


def stem_im2col(x: torch.Tensor, k: int, pad: int) -> torch.Tensor:
    """fp32 channels-last ``[N, Ci, H, W]`` (``Ci*k*k <= 32``) -> ``[N, stem_width, H, W]`` channels-last with channel
    ``ci*k*k + r*k + s`` = the input under tap ``(r, s)`` (zero padding, zero beyond ``Ci*k*k``)."""
    _chk_nhwc(x, "x", torch.float32)
    N, Ci, H, W = x.shape
    wd = stem_width(Ci, k)
    out = _empty((N, wd, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if wd == 32:
        rc = _stem_im2col_c(_ptr(x), _ptr(out), N, H, W, Ci, int(k), int(pad), _stream())
    else:
        rc = _stem_im2col_w(_ptr(x), _ptr(out), N, H, W, Ci, int(k), int(pad), wd, _stream())
    if rc != 0:
        raise ValueError(f"stem_im2col: unsupported shape Ci={Ci} k={k}")
    return out


_stem_valu_ok = _opt_sig("blades_stem_valu_ok", [c_long, c_long, c_int, c_int], c_int)
_stem_fwd_valu = _opt_sig("blades_stem_fwd_valu", [_P, _P, c_long, _P, c_long, _P, c_long, c_long, c_int, _P], c_int)
_stem_wgrad_nchunk = _opt_sig("blades_stem_wgrad_nchunk", [c_long], c_int)
_stem_wgrad_valu = _opt_sig("blades_stem_wgrad_valu", [_P, _P, c_int, c_long, c_int, c_int, _P, _P, c_long, c_long, c_int,
                                                       _P, _P], c_int)
_STEM_VALU = _os.environ.get("BLADES_STEM_VALU", "1") != "0"


def stem_valu_ok(col: torch.Tensor, Co: int, groups: int) -> bool:
    """Can the vector-ALU first-layer kernels (ops/csrc/stem_valu.hip) take this im2col tensor?  ``groups`` = number of
    weight sets the rows are divided over (1: shared weights; Kc: per-client weights / per-client gradients)."""
    if _stem_valu_ok is None or not _STEM_VALU or not (col.is_cuda and col.dtype == torch.float32 and col.dim() == 4):
        return False
    N, wd, H, W = col.shape
    M = N * H * W
    if groups < 1 or M % groups or not col.is_contiguous(memory_format=torch.channels_last):
        return False
    return bool(_stem_valu_ok(M, M // groups, int(wd), int(Co)))


def stem_forward_valu(col: torch.Tensor, w2: torch.Tensor, bias) -> torch.Tensor:
    """``y[n, :, h, w] = w2 col[n, :, h, w] + bias`` in plain fp32 on the vector ALUs.  ``col`` ``[N, 16, H, W]``
    channels-last; ``w2`` ``[Co, 16, 1, 1]`` (shared) or ``[Kc, Co, 16, 1, 1]`` (image n uses set n // (N / Kc));
    ``bias`` ``[Co]`` / ``[Kc, Co]`` / None.  Returns ``[N, Co, H, W]`` channels-last."""
    N, wd, H, W = col.shape
    M = N * H * W
    if w2.dim() == 5:
        Kc, Co = int(w2.shape[0]), int(w2.shape[1])
        if w2.stride(2) != 1 or w2.stride(1) != wd or w2.stride(0) % 4 or w2.data_ptr() % 16:
            w2 = w2.contiguous()
        w_cs = int(w2.stride(0))
    else:
        Kc, Co, w_cs = 1, int(w2.shape[0]), 0
        w2 = w2.contiguous()
    b_cs = 0
    if bias is not None:
        if bias.dim() == 2:
            if bias.stride(1) != 1:
                bias = bias.contiguous()
            b_cs = int(bias.stride(0))
        elif not bias.is_contiguous():
            bias = bias.contiguous()
    out = _empty((N, Co, H, W), dtype=torch.float32, device=col.device, memory_format=torch.channels_last)
    rc = _stem_fwd_valu(_ptr(col), _ptr(w2), w_cs, _ptr(bias) if bias is not None else None, b_cs, _ptr(out), M, M // Kc, Co,
                        _stream())
    if rc != 0:
        raise ValueError(f"stem_forward_valu: rejected (rc={rc})")
    return out


def stem_wgrad_valu_(col, gy, Kc: int, U_block, off: int, coef, accumulate: bool, kreal: int) -> bool:
    """``U_block[c, off + o * kreal + k] (+)= coef[c] * sum over client c's pixels of gy[., o] * col[., k]`` (k < kreal),
    plain fp32.  ``col`` ``[N, 16, H, W]``, ``gy`` ``[N, Co, H, W]``, both channels-last fp32."""
    N, wd, H, W = col.shape
    Co = int(gy.shape[1])
    if not gy.is_contiguous(memory_format=torch.channels_last):
        gy = gy.contiguous(memory_format=torch.channels_last)
    K, Dl, ld = _chk_matrix(U_block)
    if K != Kc or off < 0 or off + Co * kreal > Dl:
        raise ValueError("stem_wgrad_valu_: update-matrix block does not match")
    R = N * H * W // Kc
    coef = _vec(coef, Kc, col.device, name="coef")
    partial = _empty(Kc * int(_stem_wgrad_nchunk(R)) * Co * wd, device=col.device)
    rc = _stem_wgrad_valu(_ptr(col), _ptr(gy), Kc, R, Co, int(kreal), _ptr(coef), _ptr(U_block), ld, off, int(accumulate),
                          _ptr(partial), _stream())
    if rc != 0:
        raise ValueError(f"stem_wgrad_valu_: rejected (rc={rc})")
    return True


_im2col_c = _opt_sig("blades_im2col", [_P, _P, c_long] + [c_int] * 13 + [_P], c_int)


def im2col(x: torch.Tensor, kh: int, kw: int, sh: int, sw: int, ph: int, pw: int) -> torch.Tensor:
    """Channels-last ``[N, Ci, H, W]`` (fp32 / bf16) -> ``[N*Ho*Wo, Ci*kh*kw]`` with column ``ci*kh*kw + r*kw + s`` =
    the input under tap (r, s) of channel ci (zero outside the image): the row order of ``weight.reshape(Co, -1)``."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in _ACT_DTYPES):
        raise ValueError(f"im2col: expected a CUDA fp32/bf16 4-D tensor, got {x.dtype} {tuple(x.shape)}")
    x = x.contiguous(memory_format=torch.channels_last)
    N, Ci, H, W = x.shape
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    if Ho < 1 or Wo < 1:
        raise ValueError("im2col: empty output")
    K = Ci * kh * kw
    col = _empty((N * Ho * Wo, K), dtype=x.dtype, device=x.device)
    rc = _im2col_c(_ptr(x), _ptr(col), N, H, W, Ci, Ho, Wo, kh, kw, sh, sw, ph, pw, K, x.element_size(), _stream())
    if rc != 0:
        raise ValueError(f"im2col: rejected (rc={rc})")
    return col


def conv_wgrad_im2col_(x, gy, Kc: int, U_block, off: int, coef, accumulate: bool, ks, stride, pad, npl: int = 3):
    """Per-client convolution weight gradient for geometries conv_wgrad.hip has no schedule for (kernel sizes other
    than 1 / 3, strides other than 1 / 2, rectangular kernels): ``dW_c = gy_c^T im2col(x_c)`` on the grouped-GEMM
    kernel, written (scaled by coef[c]) into ``U_block[c, off : off + Co*Ci*kh*kw]``.  Images are processed in
    slabs so that the im2col tensor stays under ~1 GiB."""
    kh, kw = int(ks[0]), int(ks[1])
    sh, sw = int(stride[0]), int(stride[1])
    ph, pw = int(pad[0]), int(pad[1])
    N, Ci, H, W = x.shape
    Co = gy.shape[1]
    if N % Kc:
        raise ValueError("conv_wgrad_im2col_: batch is not a multiple of the number of clients")
    B = N // Kc
    Ho, Wo = gy.shape[2], gy.shape[3]
    gy = gy.contiguous(memory_format=torch.channels_last)
    g2 = gy.permute(0, 2, 3, 1).reshape(N * Ho * Wo, Co)
    K = Ci * kh * kw
    per_client = B * Ho * Wo * K * x.element_size()
    cps = max(1, min(Kc, (1 << 30) // max(per_client, 1)))  # clients per slab
    for c0 in range(0, Kc, cps):
        c1 = min(Kc, c0 + cps)
        col = im2col(x[c0 * B: c1 * B], kh, kw, sh, sw, ph, pw)
        linear_wgrad_(col, g2[c0 * B * Ho * Wo: c1 * B * Ho * Wo], c1 - c0, U_block[c0:c1], off, coef[c0:c1], accumulate,
                      npl)
    return True


def conv_forward(x, weight, bias, stride: int, pad: int, npl: int, stats_B: int = 0, planes=None):
    """y = conv2d(x, weight, bias) for channels-last x; fp32 (npl 3 / 2) or bf16 (npl 1) in and out.
    ``stats_B`` > 0 (fp32 tensors, ``stats_B`` images per client): returns ``(y, pre_stats)`` where ``pre_stats`` is
    the ``(partial, nchunk)`` pair ``pcbn_forward`` accepts, or ``None`` if this geometry cannot provide it."""
    dt = torch.bfloat16 if npl == 1 else torch.float32
    _chk_nhwc(x, "x", dt)
    N, Ci, H, W = x.shape
    Co, Ci2, kh, kw = weight.shape[-4:]
    if Ci2 != Ci:
        raise ValueError("conv_forward: channel mismatch")
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if Ho < 1 or Wo < 1:
        raise ValueError("conv_forward: empty output")
    y = _empty((N, Co, Ho, Wo), dtype=dt, device=x.device, memory_format=torch.channels_last)
    if Co == 1 or Ho * Wo == 1:
        y = y.as_strided(y.shape, (Ho * Wo * Co, 1, Wo * Co, Co))
    pre = _conv_call(x, weight, bias, y, N, H, W, Ci, Ho, Wo, Co, kh, kw, stride, pad, 0, npl, stats_B, planes=planes)
    return (y, pre) if stats_B else y


def conv_dgrad(gy, weight, in_hw, stride: int, pad: int, npl: int, planes=None, accumulate_into=None, bn_bwd=None):
    """dx of conv2d(x, weight) given gy (channels-last); ``in_hw`` = (H, W) of x.  ``accumulate_into``: a tensor of
    dx's shape and layout that the result is added to in the kernel's epilogue (and that is returned) — the other
    gradient branch of a residual block, so no separate elementwise add runs over the tensor.

    ``bn_bwd = (x_bn, relu_mask, B)``: dx is the gradient of the output of a per-client BatchNorm (input ``x_bn``, same
    shape as dx; gate bitmask of ``pcbn_forward`` or None; B images per client).  The kernel's epilogue then also writes
    that BatchNorm's backward sums and the call returns ``(dx, pre)`` with ``pre`` = the argument for
    ``pcbn_backward(pre_stats=...)`` — or None where the geometry does not allow it (the caller's BatchNorm then
    runs its own reduction pass)."""
    dt = torch.bfloat16 if npl == 1 else torch.float32
    _chk_nhwc(gy, "gy", dt)
    N, Co, Ho, Wo = gy.shape
    Co2, Ci, kh, kw = weight.shape[-4:]
    H, W = in_hw
    if Co2 != Co or Ho != (H + 2 * pad - kh) // stride + 1 or Wo != (W + 2 * pad - kw) // stride + 1:
        raise ValueError("conv_dgrad: gy does not match the convolution geometry")
    if accumulate_into is not None:
        dx = accumulate_into
        _chk_nhwc(dx, "accumulate_into", dt)
        if tuple(dx.shape) != (N, Ci, H, W):
            raise ValueError(f"conv_dgrad: accumulate_into has shape {tuple(dx.shape)}, expected {(N, Ci, H, W)}")
    else:
        dx = _empty((N, Ci, H, W), dtype=dt, device=gy.device, memory_format=torch.channels_last)
        if Ci == 1 or H * W == 1:
            dx = dx.as_strided(dx.shape, (H * W * Ci, 1, W * Ci, Ci))
    if bn_bwd is not None:
        bx, bmask, bB = bn_bwd
        ok = (npl > 1 and bx.dtype == torch.float32 and tuple(bx.shape) == tuple(dx.shape) and bx.stride() == dx.stride()
              and Ci % 8 == 0 and N % int(bB) == 0
              and (bmask is None or (bmask.dtype == torch.uint8 and bmask.is_contiguous() and bmask.numel() * 8 == bx.numel())))
        pre = None
        if ok:
            pre = _conv_call(gy, weight, None, dx, N, H, W, Ci, Ho, Wo, Co, kh, kw, stride, pad, 1, npl, planes=planes,
                             accumulate=accumulate_into is not None, bn_bwd=(bx, bmask, int(bB)))
        else:
            _conv_call(gy, weight, None, dx, N, H, W, Ci, Ho, Wo, Co, kh, kw, stride, pad, 1, npl, planes=planes,
                       accumulate=accumulate_into is not None)
        return dx, pre
    _conv_call(gy, weight, None, dx, N, H, W, Ci, Ho, Wo, Co, kh, kw, stride, pad, 1, npl, planes=planes,
               accumulate=accumulate_into is not None)
    return dx


# ---------------------------------------------------------------------------
# Linear forward / data gradient on the same kernel: Y = X W^T (+ bias) is the 1x1 convolution of the [M, K, 1, 1]
# "image" X with the [N, K, 1, 1] filter W (reference: nn.Linear in fllib/models/mnist/mlp.py,
# fllib/models/fashionmnist/cnn.py, the CCT blocks).  Shared weights [N, K] or one set per client [Kc, N, K] (rows of
# client c = rows c*R .. (c+1)*R - 1 of X), fp32 tensors as exact bf16 planes or bf16 tensors.
def linear_supported(x: torch.Tensor, weight: torch.Tensor, npl: int) -> bool:
    want = torch.bfloat16 if npl == 1 else torch.float32
    return (x.is_cuda and x.dtype == want and x.dim() >= 2 and x.numel() > 0 and weight.dtype == torch.float32
            and weight.dim() in (2, 3) and weight.shape[-1] == x.shape[-1])


def _as_rows(t: torch.Tensor) -> torch.Tensor:
    t2 = t.reshape(-1, t.shape[-1])
    if not t2.is_contiguous() or t2.data_ptr() % 16:
        t2 = t2.contiguous()
        if t2.data_ptr() % 16:
            t2 = t2.clone()
    return t2


def _w4(weight: torch.Tensor):
    """[N, K] -> [N, K, 1, 1]; [Kc, N, K] (any client stride, contiguous inside a client) -> [Kc, N, K, 1, 1]."""
    w = weight.detach()
    if w.dim() == 2:
        return w.contiguous().view(w.shape[0], w.shape[1], 1, 1)
    Kc, N, K = w.shape
    if w.stride(2) != 1 or w.stride(1) != K:
        w = w.contiguous()
    return w.as_strided((Kc, N, K, 1, 1), (w.stride(0), K, 1, 1, 1))


_pc_linear = _opt_sig("blades_pc_linear", [_P, _P, c_long, _P, c_long, _P] + [c_int] * 6 + [_P], c_int)
# rows per client up to which per-client Linear layers read their fp32 weights directly (ops/csrc/pc_linear.hip): with
# few rows the layer is bound by the weight bytes, and the separate plane split costs 2.5x those; with many rows per
# client (token models) the planes are reused often enough for conv_igemm's large tiles to win
_PC_LINEAR_MAX_ROWS = int(_os.environ.get("BLADES_PC_LINEAR_MAX_ROWS", "256"))


def _pc_linear_call(a2, weight, bias, mode: int, npl: int):
    """Per-client Linear forward (mode 0) / data gradient (mode 1) straight from the fp32 weights; None = not taken."""
    if _pc_linear is None or weight.dim() != 3 or npl not in (2, 3) or a2.dtype != torch.float32:
        return None
    Kc, N_out, K_in = (int(v) for v in weight.shape)
    M = int(a2.shape[0])
    R = M // Kc
    if R > _PC_LINEAR_MAX_ROWS or weight.stride(2) != 1 or weight.stride(1) != K_in:
        return None
    if bias is not None and mode == 0:
        if bias.dim() != 2 or bias.stride(1) != 1 or bias.dtype != torch.float32:
            return None
    out = _empty((M, N_out if mode == 0 else K_in), dtype=torch.float32, device=a2.device)
    b = bias if mode == 0 else None
    rc = _pc_linear(_ptr(a2), _ptr(weight), int(weight.stride(0)), _ptr(b) if b is not None else None,
                    int(b.stride(0)) if b is not None else 0, _ptr(out), Kc, R, K_in, N_out, mode, npl, _stream())
    if rc != 0:
        raise ValueError(f"pc_linear: rejected (rc={rc})")
    return out


def linear_forward(x, weight, bias, npl: int, planes=None):
    """``x [..., K] @ weight^T (+ bias)`` -> ``[..., N]`` in x's dtype."""
    if not linear_supported(x, weight, npl):
        raise ValueError(f"linear_forward: unsupported operands {x.dtype} {tuple(x.shape)} / {tuple(weight.shape)}")
    x2 = _as_rows(x)
    M, K = x2.shape
    N = int(weight.shape[-2])
    if weight.dim() == 3 and M % int(weight.shape[0]):
        raise ValueError("linear_forward: rows are not a multiple of the number of clients")
    if weight.dim() == 3 and planes is None:
        y = _pc_linear_call(x2, weight.detach(), bias, 0, npl)
        if y is not None:
            return y.reshape(*x.shape[:-1], N)
    y = _empty((M, N), dtype=x.dtype, device=x.device)
    xv = x2.as_strided((M, K, 1, 1), (K, 1, K, K))
    yv = y.as_strided((M, N, 1, 1), (N, 1, N, N))
    _conv_call(xv, _w4(weight), bias, yv, M, 1, 1, K, 1, 1, N, 1, 1, 1, 0, 0, npl, planes=planes)
    return y.reshape(*x.shape[:-1], N)


def linear_dgrad(gy, weight, npl: int, planes=None):
    """``gy [..., N] @ weight`` -> ``[..., K]``."""
    if not (gy.is_cuda and gy.dim() >= 2 and weight.dim() in (2, 3) and weight.shape[-2] == gy.shape[-1]
            and gy.dtype == (torch.bfloat16 if npl == 1 else torch.float32) and weight.dtype == torch.float32):
        raise ValueError(f"linear_dgrad: unsupported operands {gy.dtype} {tuple(gy.shape)} / {tuple(weight.shape)}")
    g2 = _as_rows(gy)
    M, N = g2.shape
    K = int(weight.shape[-1])
    if weight.dim() == 3 and M % int(weight.shape[0]):
        raise ValueError("linear_dgrad: rows are not a multiple of the number of clients")
    if weight.dim() == 3 and planes is None:
        dx = _pc_linear_call(g2, weight.detach(), None, 1, npl)
        if dx is not None:
            return dx.reshape(*gy.shape[:-1], K)
    dx = _empty((M, K), dtype=gy.dtype, device=gy.device)
    gv = g2.as_strided((M, N, 1, 1), (N, 1, N, N))
    dv = dx.as_strided((M, K, 1, 1), (K, 1, K, K))
    _conv_call(gv, _w4(weight), None, dv, M, 1, 1, K, 1, 1, N, 1, 1, 1, 0, 1, npl, planes=planes)
    return dx.reshape(*gy.shape[:-1], K)


# ---------------------------------------------------------------------------
# streaming kernels of the fused executor (ops/csrc/elementwise.hip)
_bias_grad_nchunk = _opt_sig("blades_bias_grad_nchunk", [c_int, c_int], c_int)
_bias_grad = _opt_sig("blades_bias_grad", [_P, c_int, c_int, c_int, _P, _P, c_long, c_long, c_int, _P, c_int, _P], c_int)
_maxpool_fwd = _opt_sig("blades_maxpool_forward", [_P, _P, _P, c_long] + [c_int] * 13 + [_P], c_int)
_maxpool_bwd = _opt_sig("blades_maxpool_backward", [_P, _P, _P, c_long] + [c_int] * 12 + [_P], c_int)
_avgpool_fwd = _opt_sig("blades_avgpool_forward", [_P, _P, c_long, c_int, c_int, c_int, _P], c_int)
_avgpool_bwd = _opt_sig("blades_avgpool_backward", [_P, _P, c_long, c_int, c_int, c_int, _P], c_int)
_act_fwd = _opt_sig("blades_act_forward", [_P, _P, c_long, c_int, c_int, _P], c_int)
_act_bwd = _opt_sig("blades_act_backward", [_P, _P, _P, c_long, c_int, c_int, _P], c_int)
_ce_clamp = _opt_sig("blades_ce_clamp", [_P, c_long, _P, c_int, c_int, c_int, c_float, c_float, _P, _P, _P, _P], c_int)


def bias_grad_(g, Kc: int, U_block, off: int, coef, accumulate: bool):
    """U_block[c, off : off + C] (+)= coef[c] * sum over client c's rows of ``g`` ([Kc*R, C], fp32 or bf16)."""
    if not (g.is_cuda and g.dim() == 2 and g.dtype in _ACT_DTYPES and g.shape[0] % Kc == 0):
        raise ValueError(f"bias_grad_: expected a CUDA fp32/bf16 [Kc*R, C] tensor, got {g.dtype} {tuple(g.shape)}")
    if not g.is_contiguous():
        g = g.contiguous()
    R, C = g.shape[0] // Kc, g.shape[1]
    K, Dl, ld = _chk_matrix(U_block)
    if K != Kc or off < 0 or off + C > Dl:
        raise ValueError("bias_grad_: update-matrix block does not match")
    coef = _vec(coef, Kc, g.device, name="coef")
    partial = _empty(Kc * _bias_grad_nchunk(R, C) * C, device=g.device)
    rc = _bias_grad(_ptr(g), Kc, R, C, _ptr(coef), _ptr(U_block), ld, off, int(accumulate), _ptr(partial),
                    g.element_size(), _stream())
    if rc != 0:
        raise ValueError(f"bias_grad_: rejected (rc={rc})")
    return True


def _pool_geom(x, k, s, p):
    kh, kw = (k, k) if isinstance(k, int) else (int(k[0]), int(k[1]))
    s = k if s is None else s
    sh, sw = (s, s) if isinstance(s, int) else (int(s[0]), int(s[1]))
    ph, pw = (p, p) if isinstance(p, int) else (int(p[0]), int(p[1]))
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    return kh, kw, sh, sw, ph, pw, Ho, Wo


def maxpool_supported(x, k, s, p, dilation=1, ceil_mode=False) -> bool:
    if not (x.is_cuda and x.dim() == 4 and x.dtype in _ACT_DTYPES and x.numel() > 0):
        return False
    if dilation not in (1, (1, 1)) or ceil_mode:
        return False
    kh, kw, sh, sw, ph, pw, Ho, Wo = _pool_geom(x, k, s, p)
    return kh * kw <= 254 and Ho >= 1 and Wo >= 1 and 2 * ph <= kh and 2 * pw <= kw


def maxpool_forward(x, k, s, p, relu: bool):
    """Channels-last max-pool (ReLU of the input folded in when ``relu``).  Returns (y, idx) — idx feeds
    ``maxpool_backward``."""
    x = x.contiguous(memory_format=torch.channels_last)
    kh, kw, sh, sw, ph, pw, Ho, Wo = _pool_geom(x, k, s, p)
    N, C, H, W = x.shape
    y = _empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    idx = _empty(N * Ho * Wo * C, dtype=torch.uint8, device=x.device)
    rc = _maxpool_fwd(_ptr(x), _ptr(y), _ptr(idx), N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw, int(relu),
                      x.element_size(), _stream())
    if rc != 0:
        raise ValueError(f"maxpool_forward: rejected (rc={rc})")
    return y, idx


def maxpool_backward(gy, idx, in_shape, k, s, p):
    N, C, H, W = in_shape
    gy = gy.contiguous(memory_format=torch.channels_last)
    x_like = torch.empty(0, dtype=gy.dtype, device=gy.device).new_empty((N, C, H, W))
    kh, kw, sh, sw, ph, pw, Ho, Wo = _pool_geom(x_like, k, s, p)
    if tuple(gy.shape) != (N, C, Ho, Wo) or idx.numel() != gy.numel():
        raise ValueError("maxpool_backward: gy / idx do not match the pooling geometry")
    dx = _empty((N, C, H, W), dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
    rc = _maxpool_bwd(_ptr(gy), _ptr(idx), _ptr(dx), N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw, gy.element_size(),
                      _stream())
    if rc != 0:
        raise ValueError(f"maxpool_backward: rejected (rc={rc})")
    return dx


def avgpool_global_forward(x):
    """[N, C, H, W] channels-last -> [N, C] mean over the map."""
    x = x.contiguous(memory_format=torch.channels_last)
    N, C, H, W = x.shape
    y = _empty((N, C), dtype=x.dtype, device=x.device)
    if _avgpool_fwd(_ptr(x), _ptr(y), N, H * W, C, x.element_size(), _stream()) != 0:
        raise ValueError("avgpool_global_forward: rejected")
    return y


def avgpool_global_backward(gy, hw):
    gy = gy.contiguous()
    N, C = gy.shape
    H, W = hw
    dx = _empty((N, C, H, W), dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
    if _avgpool_bwd(_ptr(gy), _ptr(dx), N, H * W, C, gy.element_size(), _stream()) != 0:
        raise ValueError("avgpool_global_backward: rejected")
    return dx


def act_forward(x, mode: int):
    """mode 0: GELU (erf), 1: ReLU."""
    x = x if x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)) else x.contiguous()
    y = _empty_like(x)
    if _act_fwd(_ptr(x), _ptr(y), x.numel(), mode, x.element_size(), _stream()) != 0:
        raise ValueError("act_forward: rejected")
    return y


def act_backward(saved, gy, mode: int):
    if gy.stride() != saved.stride() or gy.dtype != saved.dtype:
        gy = gy.to(saved.dtype).contiguous()
        if gy.stride() != saved.stride():
            saved = saved.contiguous()
    dx = _empty_like(saved)
    if _act_bwd(_ptr(saved), _ptr(gy), _ptr(dx), saved.numel(), mode, saved.element_size(), _stream()) != 0:
        raise ValueError("act_backward: rejected")
    return dx


def ce_clamp(logits, labels, lo: float, hi: float, want_grad: bool = True):
    """Per-client clamped mean cross-entropy (reference fllib/tasks/mnist.py:8-18) of ``logits [Kc, B, C]`` (fp32) and
    ``labels [Kc, B]``: returns (loss [Kc], d(sum_c loss_c)/dlogits [Kc, B, C] or None).  Labels must lie in [0, C)
    (checked by the callers that create them; out-of-range entries are read as class 0)."""
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 3):
        raise ValueError("ce_clamp: logits must be a CUDA fp32 [Kc, B, C] tensor")
    lg = logits if logits.is_contiguous() else logits.contiguous()
    Kc, B, C = lg.shape
    y = labels.reshape(Kc, B)
    if y.dtype != torch.int64:
        y = y.long()
    y = y.contiguous()
    loss = _empty(Kc, device=lg.device)
    dz = _empty((Kc, B, C), device=lg.device) if want_grad else None
    rc = _ce_clamp(_ptr(lg), C, _ptr(y), Kc, B, C, float(lo), float(hi), c_void_p(0), _ptr(loss), _ptr(dz), _stream())
    if rc != 0:
        raise ValueError(f"ce_clamp: rejected (rc={rc})")
    return loss, dz


_seqpool_fwd = _opt_sig("blades_seqpool_forward", [_P, _P, _P, _P, c_long, c_int, c_int, _P], c_int)
_seqpool_bwd = _opt_sig("blades_seqpool_backward", [_P, _P, _P, _P, _P, c_long, c_int, c_int, _P], c_int)


def seqpool_supported(x, logit) -> bool:
    return (_seqpool_fwd is not None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3
            and logit.dtype == torch.float32 and logit.numel() == x.shape[0] * x.shape[1] and 1 <= x.shape[1] <= 1024)


def seqpool_forward(x, logit):
    """``y[b] = softmax_t(logit[b])^T x[b]`` for ``x [B, T, C]``, ``logit [B, T]`` (or ``[B, T, 1]``): (y [B, C], p [B, T])."""
    B, T, C = (int(v) for v in x.shape)
    x = x if x.is_contiguous() else x.contiguous()
    lg = logit.reshape(B, T)
    lg = lg if lg.is_contiguous() else lg.contiguous()
    y = _empty((B, C), device=x.device)
    p = _empty((B, T), device=x.device)
    if _seqpool_fwd(_ptr(x), _ptr(lg), _ptr(y), _ptr(p), B, T, C, _stream()) != 0:
        raise ValueError("seqpool_forward: rejected")
    return y, p


def seqpool_backward(x, p, gy):
    """Gradients of ``seqpool_forward``: (dx [B, T, C], dlogit [B, T])."""
    B, T, C = (int(v) for v in x.shape)
    x = x if x.is_contiguous() else x.contiguous()
    gy = gy if gy.is_contiguous() else gy.contiguous()
    dx = _empty((B, T, C), device=x.device)
    dl = _empty((B, T), device=x.device)
    if _seqpool_bwd(_ptr(x), _ptr(p), _ptr(gy), _ptr(dx), _ptr(dl), B, T, C, _stream()) != 0:
        raise ValueError("seqpool_backward: rejected")
    return dx, dl


# ---------------------------------------------------------------------------
# fused evaluation metrics
_ce_topk = _sig("blades_ce_topk", [_P, c_long, _P, c_long, c_int, _P, c_int, _P, c_int, _P, _P], c_int)
_ks_cache = {}


def ce_topk(logits, labels, ks):
    lg = logits.reshape(-1, logits.shape[-1])
    if lg.dtype != torch.float32:
        lg = lg.float()
    if lg.stride(1) != 1:
        lg = lg.contiguous()
    y = labels.reshape(-1)
    if y.dtype != torch.int64:
        y = y.long()
    y = y.contiguous()
    N, C = lg.shape
    if y.numel() != N:
        raise ValueError(f"ce_topk: {N} logit rows but {y.numel()} labels")
    if len(ks) > 8:
        raise ValueError("ce_topk: at most 8 values of k")
    import ctypes

    karr = _ks_cache.get(ks)
    if karr is None:
        karr = (ctypes.c_int * max(1, len(ks)))(*ks)
        _ks_cache[ks] = karr
    out = torch.zeros(1 + len(ks), dtype=torch.float64, device=lg.device)
    if N == 0:
        return out
    part = _empty(((N + 255) // 256) * 9, dtype=torch.float64, device=lg.device)
    rc = _ce_topk(_ptr(lg), lg.stride(0), _ptr(y), N, C, ctypes.cast(karr, c_void_p), len(ks), _ptr(part), 0,
                  _ptr(out), _stream())
    if rc != 0:
        raise ValueError(f"ce_topk rejected the call (code {rc})")
    return out
