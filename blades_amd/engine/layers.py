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
