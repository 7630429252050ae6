"""Engine-side layer substitutions.

The batched engine runs a *template copy* of the task's model in which some
layers are replaced by implementations that behave identically per client but
suit vectorised execution.  Parameter names and shapes are unchanged, so the
flat parameter vector / ``functional_call`` dictionaries are shared with the
original model.
"""
import copy

import torch
from torch import nn


class ClientBatchNorm2d(nn.Module):
    """BatchNorm2d with batch statistics only (``track_running_stats=False``),
    statistics in fp32 regardless of the autocast dtype.

    Under ``vmap`` over clients the reduction dims (0, 2, 3) are one client's
    batch, so the statistics are per (client, channel) — SURVEY §7.3 item 1.
    (MIOpen's batch-norm cannot be used here: the vmap batching rule hands it
    bf16 affine parameters under autocast, which it rejects.)
    """

    def __init__(self, bn: nn.BatchNorm2d):
        super().__init__()
        self.num_features = bn.num_features
        self.eps = bn.eps
        self.affine = bn.affine
        if bn.affine:
            self.weight = bn.weight
            self.bias = bn.bias
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    def forward(self, x):
        xf = x.float()
        var, mean = torch.var_mean(xf, dim=(0, 2, 3), unbiased=False, keepdim=True)
        y = (xf - mean) * torch.rsqrt(var + self.eps)
        if self.affine:
            y = y * self.weight.float().view(1, -1, 1, 1) + self.bias.float().view(1, -1, 1, 1)
        return y.to(x.dtype) if torch.is_autocast_enabled() else y


def make_engine_model(model: nn.Module) -> nn.Module:
    """Shallow structural copy of ``model`` (parameters shared) with engine layers."""
    memo = {id(p): p for p in model.parameters()}  # share, don't copy, the parameters
    tmpl = copy.deepcopy(model, memo)

    def swap(mod: nn.Module):
        for name, child in list(mod.named_children()):
            if isinstance(child, nn.BatchNorm2d) and not child.track_running_stats:
                setattr(mod, name, ClientBatchNorm2d(child))
            else:
                swap(child)

    swap(tmpl)
    return tmpl


# ---------------------------------------------------------------------------------------------------------------
# Client-keyed dropout.  torch's dropout draws from the process RNG, so the masks a client sees would depend on the
# rank it is trained on, on chunking and on the execution path.  The engine templates use masks that are a
# counter-based function of (seed, client uid, round / local step, layer) instead: 1, 2 or 8 ranks — fused,
# grouped or vmap path — train every client with the same masks (reference: nn.Dropout in
# fllib/models/mnist/mlp.py, Dropout2d in fllib/models/fashionmnist/cnn.py, DropPath in cctnets).
class DropContext:
    """Which clients the batch in flight belongs to; shared by every ``ClientDropout`` of an engine."""

    def __init__(self, seed: int, device):
        self.seed = int(seed) * 1000003 + 17
        self.device = device
        self.layout = "major"       # "major": rows are [Kc*B, ...] client-major; "inner": [B, Kc*C, ...] (grouped
        self.uids = None            # executor); "single": one client's [B, ...] batch (vmap), uid = 0-dim tensor
        self.Kc = 0
        self.tag = torch.zeros(1, dtype=torch.int64, device=device)  # round * nb + step, on the device: a captured
        self.enabled = True                                          # HIP graph reads the current value
        self._tags = {}

    def set(self, uids: torch.Tensor, layout: str):
        self.uids, self.layout, self.Kc = uids, layout, (int(uids.numel()) if uids.dim() else 1)

    def bind(self, key) -> None:
        """Makes ``tag`` the device scalar that belongs to chunk ``key``.  The dropout kernels read the tag when they
        *execute*; chunks run on several HIP streams (and, with several local steps, at different step numbers at
        the same time), so a scalar shared by all chunks would be overwritten by one chunk's ``set_tag`` while
        another chunk's kernels are still queued.  Every chunk therefore fills and reads a scalar of
        its own, which also lives as long as any HIP graph captured for that chunk."""
        t = self._tags.get(key)
        if t is None:
            if len(self._tags) > 8192:
                self._tags.clear()  # (tensors referenced by live graphs stay alive through the graphs' keep lists)
            t = self._tags[key] = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.tag = t

    def set_tag(self, value: int):
        self.tag.fill_(int(value))


class ClientDropout(nn.Module):
    """kind "elem": nn.Dropout; "channel": nn.Dropout2d (whole channels; on a 2-D input every element, as torch
    does); "sample": DropPath (whole samples)."""

    def __init__(self, p: float, kind: str, layer: int, ctx: DropContext):
        super().__init__()
        self.p, self.kind, self.layer, self.ctx = float(p), kind, int(layer), ctx

    def extra_repr(self):
        return f"p={self.p}, kind={self.kind}, layer={self.layer}"

    def forward(self, x):
        ctx = self.ctx
        if not self.training or self.p <= 0.0 or not ctx.enabled or ctx.uids is None:
            return x
        from blades_amd import ops

        lay, Kc = ctx.layout, ctx.Kc
        if lay == "single":
            B = x.shape[0]
            keep = 1 if self.kind == "sample" else (2 if self.kind == "channel" and x.dim() > 2 else x.dim())
            n = 1
            for d in x.shape[:keep]:
                n *= int(d)
            m = ops.client_dropout_mult(ctx.uids, n, ctx.seed, ctx.tag, self.layer, self.p)
            return x * m.reshape(*x.shape[:keep], *([1] * (x.dim() - keep))).to(x.dtype)
        if lay == "major":
            B = x.shape[0] // Kc
            keep = 1 if self.kind == "sample" else (2 if self.kind == "channel" and x.dim() > 2 else x.dim())
            per = 1
            for d in x.shape[1:keep]:
                per *= int(d)
            m = ops.client_dropout_mult(ctx.uids, B * per, ctx.seed, ctx.tag, self.layer, self.p)
            return x * m.reshape(*x.shape[:keep], *([1] * (x.dim() - keep))).to(x.dtype)
        # "inner": x is [B, Kc*C, ...] with the client index outermost inside dim 1
        B, C = x.shape[0], x.shape[1] // Kc
        if self.kind == "sample":
            m = ops.client_dropout_mult(ctx.uids, B, ctx.seed, ctx.tag, self.layer, self.p)  # [Kc, B]
            m = m.t().reshape(B, Kc, 1).expand(B, Kc, C).reshape(B, Kc * C, *([1] * (x.dim() - 2)))
            return x * m.to(x.dtype)
        keep = 2 if self.kind == "channel" and x.dim() > 2 else x.dim()
        per = C
        for d in x.shape[2:keep]:
            per *= int(d)
        m = ops.client_dropout_mult(ctx.uids, B * per, ctx.seed, ctx.tag, self.layer, self.p)  # [Kc, B*per]
        m = m.reshape(Kc, B, C, *x.shape[2:keep]).transpose(0, 1).reshape(B, Kc * C, *x.shape[2:keep])
        return x * m.reshape(*m.shape, *([1] * (x.dim() - keep))).to(x.dtype)


def swap_dropout(root: nn.Module, ctx: DropContext) -> nn.Module:
    """Replaces every dropout-like leaf of ``root`` (in place) by a ``ClientDropout``; layer numbers follow the
    module order, which is the same in every structural copy of a model."""
    from blades_amd.models.cctnets.layers import DropPath

    counter = [0]

    def visit(mod):
        for name, child in list(mod.named_children()):
            kind = None
            if isinstance(child, (nn.Dropout2d, nn.Dropout3d, nn.Dropout1d)):
                kind = "channel"
            elif isinstance(child, (nn.Dropout, nn.AlphaDropout)):
                kind = "elem"
            elif isinstance(child, DropPath):
                kind = "sample"
            if kind is not None:
                setattr(mod, name, ClientDropout(child.p, kind, counter[0], ctx))
                counter[0] += 1
            elif not isinstance(child, ClientDropout):
                visit(child)

    holder = nn.Module()
    holder.root = root
    visit(holder)
    return holder.root
