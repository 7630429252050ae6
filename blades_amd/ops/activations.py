"""Activation functions of the model zoo with a hand-written GPU path.

The models (``models/mlp.py``, ``models/cctnets``) call ``relu`` / ``gelu`` from here instead of ``torch.nn.functional``.
Everywhere except inside the fused executor they ARE the torch functions (CPU oracle, ``vmap`` / sequential engine
paths, user code).  While the fused executor runs a template (``fused_scope``) and the tensor lives on the GPU they run
ops/csrc/elementwise.hip: one kernel forward, one backward, no ATen elementwise / clamp kernels in the round
(reference: ``F.gelu`` in fllib/models/backbones/cctnets/utils/transformers.py, ``F.relu`` in fllib/models/mnist/mlp.py).
"""
import torch
import torch.nn.functional as F

_depth = 0


class fused_scope:
    """Marks the dynamic extent of a fused-executor forward/backward (engine/fused.py)."""

    def __enter__(self):
        global _depth
        _depth += 1
        return self

    def __exit__(self, *exc):
        global _depth
        _depth -= 1
        return False


def _hip_ok(x: torch.Tensor) -> bool:
    return _depth > 0 and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.numel() > 0


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        from blades_amd import ops

        h = ops._h()
        y = h.act_forward(x, mode)
        ctx.mode = mode
        ctx.save_for_backward(x if mode == 0 else y)  # GELU differentiates at its input, ReLU gates on its output
        return y

    @staticmethod
    def backward(ctx, gy):
        from blades_amd import ops

        (saved,) = ctx.saved_tensors
        return ops._h().act_backward(saved, gy, ctx.mode), None


def gelu(x: torch.Tensor) -> torch.Tensor:
    return _ActFn.apply(x, 0) if _hip_ok(x) else F.gelu(x)


def relu(x: torch.Tensor) -> torch.Tensor:
    return _ActFn.apply(x, 1) if _hip_ok(x) else F.relu(x)


class _SeqPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, logit):
        from blades_amd import ops

        y, p = ops._h().seqpool_forward(x, logit)
        ctx.save_for_backward(x, p)
        ctx.lshape = logit.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        from blades_amd import ops

        x, p = ctx.saved_tensors
        dx, dl = ops._h().seqpool_backward(x, p, gy)
        return dx, dl.reshape(ctx.lshape)


def seq_pool(x: torch.Tensor, logit: torch.Tensor) -> torch.Tensor:
    """Sequence pooling of the compact transformers: ``softmax(logit, dim=tokens)^T x`` for ``x [B, T, C]`` and
    ``logit [B, T, 1]`` -> ``[B, C]`` (reference: fllib/models/backbones/cctnets/utils/transformers.py, the
    ``attention_pool`` branch).  Inside the fused executor on the GPU: one kernel forward, one backward
    (ops/csrc/elementwise.hip) instead of softmax + transpose + a batched library GEMM and their backward kernels."""
    if _depth > 0 and x.is_cuda and x.dtype == torch.float32 and logit.dtype == torch.float32 and x.dim() == 3:
        from blades_amd import ops

        if ops._h().seqpool_supported(x, logit):
            return _SeqPoolFn.apply(x, logit)
    return torch.matmul(F.softmax(logit, dim=1).transpose(-1, -2), x).squeeze(-2)
