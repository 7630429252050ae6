"""Update-matrix primitives: HIP kernels on the GPU, torch oracle on the CPU.

There is exactly one GPU implementation of every primitive — the hand-written
gfx950 kernel in ``csrc/`` loaded through ``_hip`` — and no silent fallback:
calling a primitive with a GPU tensor while ``libblades_hip.so`` is missing
raises.  CPU tensors run ``reference.py`` (the numerical contract the kernels
are tested against), which is what keeps the whole framework runnable and
testable on a machine without a GPU.
"""
import torch

from . import reference as ref

_FORCE_REFERENCE = False  # tests flip this to compare kernel vs oracle on GPU


def _use_hip(t: torch.Tensor) -> bool:
    return t.is_cuda and not _FORCE_REFERENCE


_hip_module = None


def _h():
    """The ctypes kernel module (lazy import: loading it needs the built .so)."""
    global _hip_module
    if _hip_module is None:
        import importlib

        _hip_module = importlib.import_module("blades_amd.ops._hip")
    return _hip_module


class force_reference:
    """Context manager: run the torch oracle even for GPU tensors."""

    def __enter__(self):
        global _FORCE_REFERENCE
        self._old = _FORCE_REFERENCE
        _FORCE_REFERENCE = True

    def __exit__(self, *a):
        global _FORCE_REFERENCE
        _FORCE_REFERENCE = self._old


def _f32c(U):
    assert U.dtype == torch.float32 and U.dim() == 2, "update matrix must be [K, D] fp32"
    assert U.stride(1) == 1, "update matrix rows must be contiguous"
    return U


def weighted_col_sum(U, w):
    if _use_hip(_f32c(U)):
        return _h().weighted_col_sum(U, w)
    return ref.weighted_col_sum(U, w)


def col_mean(U, mask=None):
    if _use_hip(_f32c(U)):
        K = U.shape[0]
        if mask is None:
            w = torch.full((K,), 1.0 / K, device=U.device, dtype=torch.float32)
        else:
            m = mask.to(device=U.device, dtype=torch.float32)
            w = m / m.sum()
        return _h().weighted_col_sum(U, w)
    return ref.col_mean(U, mask)


def col_mean_std(U, mask=None):
    if _use_hip(_f32c(U)):
        return _h().col_mean_std(U, mask)
    return ref.col_mean_std(U, mask)


def col_min_max(U, mask=None):
    if _use_hip(_f32c(U)):
        return _h().col_min_max(U, mask)
    return ref.col_min_max(U, mask)


def col_median(U, mask=None):
    if _use_hip(_f32c(U)):
        return _h().col_order_stat(U, mask, 0, 0)
    return ref.col_median(U, mask)


def col_trimmed_mean(U, b, mask=None):
    n = U.shape[0] if mask is None else int((mask != 0).sum())
    if n <= 2 * b:
        raise ValueError("Not enough inputs to compute trimmed mean")
    if _use_hip(_f32c(U)):
        return _h().col_order_stat(U, mask, 1, int(b))
    return ref.col_trimmed_mean(U, b, mask)


def row_stats(U, valid_cols=None):
    if _use_hip(_f32c(U)):
        return _h().row_stats(U, U.shape[1] if valid_cols is None else int(valid_cols))
    return ref.row_stats(U, valid_cols)


def row_sqnorm(U):
    if _use_hip(_f32c(U)):
        return _h().row_stats(U, U.shape[1], counts=False)[0]
    return ref.row_sqnorm(U)


def sqdist_to(U, m):
    if _use_hip(_f32c(U)):
        return _h().sqdist_to(U, m)
    return ref.sqdist_to(U, m)


def row_dot(U, v):
    if _use_hip(_f32c(U)):
        return _h().row_dot(U, v)
    return ref.row_dot(U, v)


def gram(U):
    if _use_hip(_f32c(U)):
        return _h().gram(U)
    return ref.gram(U)


def gram_batched(S):
    """Gram matrices of a stack ``[B, K, n]`` (DnC's iterations): one launch for K <= 128 on the GPU."""
    if S.is_cuda and not _FORCE_REFERENCE and S.dtype == torch.float32 and S.dim() == 3 and 1 <= S.shape[1] <= 128:
        return _h().gram_batched(S.contiguous())
    if S.is_cuda and not _FORCE_REFERENCE and S.dtype == torch.float32:
        return torch.stack([gram(s) for s in S])
    return ref.gram_batched(S)


def sample_distinct(D, n, iters, seed, device):
    """``iters`` sets of ``n`` distinct ids in ``[0, D)`` — a keyed pseudo-random permutation of ``[0, D)`` applied to
    ``0..n-1``; the same numbers on the CPU and on the GPU.  int64 ``[iters, n]`` on ``device``."""
    device = torch.device(device)
    if device.type == "cuda" and not _FORCE_REFERENCE:
        return _h().sample_distinct(D, n, iters, seed, device)
    return ref.sample_distinct(int(D), int(n), int(iters), int(seed)).to(device)


def gather_center_cols(U, idx):
    """Centred column sub-samples ``[iters, K, n]`` of ``U`` at the column ids ``idx[iters, n]``."""
    if _use_hip(_f32c(U)):
        return _h().gather_center_cols(U, idx.to(U.device))
    return ref.gather_center_cols(U, idx)


def row_scale_(U, s):
    if _use_hip(_f32c(U)):
        return _h().row_scale_(U, s)
    return ref.row_scale_(U, s)


def set_rows_(U, rows, v):
    if _use_hip(_f32c(U)) and v.dim() == 1:
        return _h().set_rows_(U, rows, v)
    return ref.set_rows_(U, rows, v)


def alie_forge_(U, benign_mask, mal_rows, z, sign=None):
    if _use_hip(_f32c(U)):
        return _h().alie_forge_(U, benign_mask, mal_rows, float(z), sign)
    return ref.alie_forge_(U, benign_mask, mal_rows, z, sign)


def ipm_forge_(U, benign_mask, mal_rows, scale):
    if _use_hip(_f32c(U)):
        return _h().ipm_forge_(U, benign_mask, mal_rows, float(scale))
    return ref.ipm_forge_(U, benign_mask, mal_rows, scale)


def server_step_(theta, buf, agg, lr, momentum, first):
    if theta.is_cuda and not _FORCE_REFERENCE:
        return _h().server_step_(theta, buf, agg, float(lr), float(momentum), bool(first))
    return ref.server_step_(theta, buf, agg, lr, momentum, first)


def guarded_step_available(theta, agg) -> bool:
    """The fused non-finite guard + server step (ops/csrc/optim.hip) applies: fp32 contiguous CUDA vectors."""
    return (theta.is_cuda and agg.is_cuda and not _FORCE_REFERENCE and theta.dtype == torch.float32
            and agg.dtype == torch.float32 and theta.is_contiguous() and agg.is_contiguous())


def nonfinite_flag(x):
    return _h().nonfinite_flag(x)


def server_step_guarded_(theta, buf, agg, lr, momentum, first, bad, skipped):
    return _h().server_step_guarded_(theta, buf, agg, float(lr), float(momentum), bool(first), bad, skipped)


def client_sgd_step_(W, G, M, lr, momentum, first, gscale=None):
    if W.is_cuda and not _FORCE_REFERENCE:
        return _h().client_sgd_step_(W, G, M, float(lr), float(momentum), bool(first), gscale)
    return ref.client_sgd_step_(W, G, M, lr, momentum, first, gscale)


def philox_uniform(n, seed, stream, offset=0, device="cpu"):
    if torch.device(device).type == "cuda" and not _FORCE_REFERENCE:
        return _h().philox_uniform(n, seed, stream, offset, device)
    return ref.philox_uniform(n, seed, stream, offset, device)


def philox_normal(n, seed, stream, offset=0, mean=0.0, std=1.0, device="cpu"):
    if torch.device(device).type == "cuda" and not _FORCE_REFERENCE:
        return _h().philox_normal(n, seed, stream, offset, mean, std, device)
    return ref.philox_normal(n, seed, stream, offset, mean, std, device)


def adaptive_forge(mean, mn, mx, seed, stream, offset=0, b=2.0):
    """Adaptive (Fang) attack vector in one pass (GPU: csrc/rng.hip adaptive_forge_kernel)."""
    if mean.is_cuda and not _FORCE_REFERENCE:
        return _h().adaptive_forge(mean.contiguous(), mn.contiguous(), mx.contiguous(), int(seed), int(stream), int(offset), b)
    return ref.adaptive_forge(mean, mn, mx, seed, stream, offset, b)


def perm_keys(n, seed, stream, offset, d_valid_end, device="cpu"):
    if torch.device(device).type == "cuda" and not _FORCE_REFERENCE:
        return _h().perm_keys(int(n), int(seed), int(stream), int(offset), int(d_valid_end), device)
    return ref.perm_keys(n, seed, stream, offset, d_valid_end, device)


def signguard_forge(keys, thresholds, seed, stream, offset=0):
    if keys.is_cuda and not _FORCE_REFERENCE:
        return _h().signguard_forge(keys.contiguous(), thresholds.contiguous(), int(seed), int(stream), int(offset))
    return ref.signguard_forge(keys, thresholds, seed, stream, offset)


def add_normal_rows_(U, rows, seed, stream, col_offset, d_total, mean, std, overwrite):
    if _use_hip(_f32c(U)):
        return _h().add_normal_rows_(U, rows, seed, stream, col_offset, d_total, mean, std, overwrite)
    return ref.add_normal_rows_(U, rows, seed, stream, col_offset, d_total, mean, std, overwrite)


def power_iteration_top(G, iters=8):
    """Top eigenvector of the K x K Gram matrix (DnC's outlier direction): the direction ``G^(64 + iters) v0``.
    Small client counts (K <= 128, the reference's 60-client experiments): one launch of ops/csrc/smallk.hip with G in
    LDS — the ~30 library calls of the reference form cost 0.7 ms of launch overhead per call at that size.  Larger K:
    six squarings + ``iters`` matrix-vector products on library GEMMs (K = 1000: 6 GEMMs of 1000^3), no host sync."""
    if G.is_cuda and not _FORCE_REFERENCE and G.dtype == torch.float32 and 1 <= G.shape[-1] <= _h().SMALLK_MAX:
        return _h().power_top(G, 64 + int(iters))  # (a [B, K, K] stack: one workgroup per matrix)
    if G.dim() == 3:
        return torch.stack([ref.power_iteration_top(g, iters) for g in G])
    return ref.power_iteration_top(G, iters)


def smallk_max(device) -> int:
    """Largest client count the single-launch Gram-space solvers take on this device (0: none)."""
    if getattr(device, "type", str(device)) != "cuda" or _FORCE_REFERENCE:
        return 0
    return int(_h().SMALLK_MAX)


def minmax_bisect(a, b, c, thr, lo, hi, tol):
    """The min-max attack's bisection in one launch (GPU): 0-d tensor gamma."""
    return _h().minmax_bisect(a, b, c, thr, lo, hi, tol)


def cclip(G, p, mm, n_iter, tau):
    """Centred-clipping coefficients (c0, c) from the Gram matrix in one launch (K <= smallk_max)."""
    return _h().cclip(G, p, mm, n_iter, tau)


def weiszfeld(G, w0, maxiter, eps, ftol):
    """Geometric-median coefficients from the Gram matrix in one launch (K <= smallk_max): (alpha, iterations)."""
    return _h().weiszfeld(G, w0, maxiter, eps, ftol)


def ce_topk(logits, labels, ks=(1,)):
    """Summed CE + top-k hit counts -> float64 ``[1 + len(ks)]`` (one fused kernel on the GPU)."""
    if logits.is_cuda and not _FORCE_REFERENCE:
        return _h().ce_topk(logits, labels, tuple(int(k) for k in ks))
    return ref.ce_topk(logits, labels, ks)


def shuffle_keys(uids, epochs, lens, nmax, seed):
    """Sort keys of the device batch sampler ``[rows, nmax]`` float64 (see reference.shuffle_keys)."""
    if uids.is_cuda and not _FORCE_REFERENCE:
        return _h().shuffle_keys(uids.contiguous(), epochs.contiguous(), lens.contiguous(), int(nmax), int(seed))
    return ref.shuffle_keys(uids, epochs, lens, int(nmax), int(seed))


def client_dropout_mult(uids, n, seed, tag, layer, p):
    """Client-keyed dropout multipliers ``[Kc, n]`` (see reference.client_dropout_mult)."""
    if uids.is_cuda and not _FORCE_REFERENCE and uids.dim() == 1 and torch.is_tensor(tag) and tag.is_cuda \
            and not torch._C._functorch.is_batchedtensor(uids):
        return _h().client_dropout_mult(uids, int(n), int(seed), tag, int(layer), float(p))
    return ref.client_dropout_mult(uids, n, seed, tag, layer, p)
