"""Per-client-weight executor for local training with more than one step (the "general path").

After the first local step every client has its own weights, so the shared-weight trick of
``engine/fused.py`` no longer applies.  Instead of vectorising a per-client function with ``torch.vmap``
(which re-lays every activation out around every convolution and decomposes BatchNorm into a dozen
elementwise kernels), the whole chunk is run as **one wide network in a client-innermost layout**:

* activations are ``[B, Kc*C, H, W]`` (channels-last), client ``k`` owning channels ``[k*C, (k+1)*C)``;
* ``Conv2d``  -> one grouped convolution with ``groups = Kc`` and weight ``[Kc*Cout, Cin, kh, kw]``;
* ``Linear``  -> one batched GEMM over the client axis;
* ``BatchNorm2d`` with batch statistics -> a plain BatchNorm over ``Kc*C`` channels: the reduction over
  ``(B, H, W)`` of channel ``k*C + c`` *is* client ``k``'s statistic, so per-client semantics need no
  special kernel;
* everything else (ReLU, pooling, residual adds, flatten) is layout-agnostic.

The layout is kept for the whole forward/backward pass — no transposes between layers — and the
per-client weight gradients fall out of autograd as the gradients of the grouped weights.

Reference behaviour being reproduced: ``fllib/clients/client.py:46-74`` (``num_batch_per_round`` local
steps per client on its own weights).  A model qualifies if all its parameters live in Conv2d (groups=1),
Linear or BatchNorm2d (``track_running_stats=False``) leaves *and* a numerical self-check against the
per-client functional forward passes (models whose ``forward`` reshapes in a layout-dependent way, e.g.
``unsqueeze(1)``, fail it and stay on the vmap path).
"""
import copy
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import nn


class _Slot:
    """Per-call tensors of one parameterised layer: ``[Kc, *param.shape]`` leaves."""

    __slots__ = ("weight", "bias")

    def __init__(self):
        self.weight: Optional[torch.Tensor] = None
        self.bias: Optional[torch.Tensor] = None


class GConv2d(nn.Module):
    def __init__(self, conv: nn.Conv2d, slot: _Slot):
        super().__init__()
        self.stride, self.padding, self.dilation = conv.stride, conv.padding, conv.dilation
        self.slot = slot

    def forward(self, x):
        w = self.slot.weight  # [Kc, Cout, Cin, kh, kw]
        Kc = w.shape[0]
        b = self.slot.bias.reshape(-1) if self.slot.bias is not None else None
        return F.conv2d(x, w.reshape(Kc * w.shape[1], *w.shape[2:]), b, self.stride, self.padding, self.dilation, Kc)


class GLinear(nn.Module):
    def __init__(self, lin: nn.Linear, slot: _Slot):
        super().__init__()
        self.slot = slot

    def forward(self, x):
        w = self.slot.weight  # [Kc, out, in]
        Kc, out_f, in_f = w.shape
        if x.dim() != 2 or x.shape[1] != Kc * in_f:
            raise ValueError("GLinear expects [B, Kc*in_features]")
        xb = x.reshape(x.shape[0], Kc, in_f).transpose(0, 1)  # [Kc, B, in]
        y = torch.bmm(xb, w.transpose(1, 2).to(xb.dtype))  # [Kc, B, out]
        if self.slot.bias is not None:
            y = y + self.slot.bias.to(y.dtype).unsqueeze(1)
        return y.transpose(0, 1).reshape(x.shape[0], Kc * out_f)


class GBatchNorm2d(nn.Module):
    def __init__(self, bn: nn.BatchNorm2d, slot: _Slot):
        super().__init__()
        self.eps = bn.eps
        self.slot = slot

    def forward(self, x):
        w = self.slot.weight.reshape(-1) if self.slot.weight is not None else None
        b = self.slot.bias.reshape(-1) if self.slot.bias is not None else None
        return F.batch_norm(x, None, None, w, b, True, 0.0, self.eps)


def supports(model: nn.Module) -> bool:
    for m in model.modules():
        if not list(m.parameters(recurse=False)):
            continue
        if any(not p.requires_grad for p in m.parameters(recurse=False)):
            return False
        if isinstance(m, nn.Conv2d) and m.groups == 1 and m.padding_mode == "zeros" and not isinstance(m.padding, str):
            continue
        if isinstance(m, nn.Linear):
            continue
        if isinstance(m, nn.BatchNorm2d) and not m.track_running_stats:
            continue
        return False
    return True


class GroupedLocalSGD:
    """``run(W, X, Y, G)``: per-client gradients of one local step into ``G[Kc, D]``; returns losses ``[Kc]``."""

    def __init__(self, task, names: List[str], shapes, offsets, numels, amp_dtype, device):
        self.task = task
        self.names, self.shapes = list(names), [tuple(s) for s in shapes]
        self.offsets, self.numels = [int(o) for o in offsets], [int(n) for n in numels]
        self.amp_dtype, self.device = amp_dtype, device
        self.slots: Dict[str, _Slot] = {}
        self.template = self._build(task.model)
        self._param_slot = []  # (slot, "weight"|"bias") per parameter, in names order
        for n in self.names:
            mod, _, leaf = n.rpartition(".")
            self._param_slot.append((self.slots[mod], leaf))

    def _build(self, model: nn.Module) -> nn.Module:
        tmpl = copy.deepcopy(model)
        for p in tmpl.parameters():
            p.requires_grad_(False)

        def swap(mod, prefix):
            for name, child in list(mod.named_children()):
                path = f"{prefix}{name}"
                if isinstance(child, (nn.Conv2d, nn.Linear, nn.BatchNorm2d)) and list(child.parameters(recurse=False)):
                    slot = self.slots.setdefault(path, _Slot())
                    cls = GConv2d if isinstance(child, nn.Conv2d) else GLinear if isinstance(child, nn.Linear) \
                        else GBatchNorm2d
                    setattr(mod, name, cls(child, slot))
                elif isinstance(child, nn.BatchNorm2d):  # affine=False: no parameters, still per-client statistics
                    setattr(mod, name, GBatchNorm2d(child, _Slot()))
                else:
                    swap(child, path + ".")

        holder = nn.Module()
        holder.root = tmpl
        swap(holder, "")
        # parameter names of the original model are relative to `root`
        self.slots = {("" if k == "root" else k[len("root."):] if k.startswith("root.") else k): v
                      for k, v in self.slots.items()}
        return holder.root

    # ------------------------------------------------------------------
    def run(self, W: torch.Tensor, X: torch.Tensor, Y: torch.Tensor, G: torch.Tensor, train: bool = True):
        with torch.enable_grad():  # callers sit under no_grad (engine round loop)
            return self._run(W, X, Y, G, train)

    def _run(self, W, X, Y, G, train):
        Kc, B = Y.shape
        leaves = []
        for (slot, leaf), s, o, n in zip(self._param_slot, self.shapes, self.offsets, self.numels):
            t = W[:, o: o + n].detach().reshape(Kc, *s).clone().requires_grad_(True)
            setattr(slot, leaf, t)
            leaves.append(t)
        x = X.transpose(0, 1)  # [B, Kc, *sample]
        if x.dim() == 5 or x.dim() == 3:  # images with a channel axis / flat features: clients fold into it
            x = x.reshape(B, Kc * x.shape[2], *x.shape[3:])
        elif x.dim() == 2:
            x = x.reshape(B, Kc)
        # channel-less images [B, Kc, H, W] already read as "Kc single-channel images side by side"
        if self.amp_dtype is not None:
            x = x.to(self.amp_dtype)
        if x.dim() == 4 and x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        self.template.train(train)
        enabled = self.amp_dtype is not None
        with torch.enable_grad(), torch.autocast(self.device.type, dtype=self.amp_dtype or torch.bfloat16,
                                                 enabled=enabled):
            logits = self.template(x)
        lg = logits.float().reshape(B, Kc, -1).transpose(0, 1)  # [Kc, B, classes]
        with torch.enable_grad():
            losses = torch.vmap(lambda z, y: self.task.loss(_identity, z, y))(lg, Y)
            losses.sum().backward()
        missing = [nm for nm, t in zip(self.names, leaves) if t.grad is None]
        if missing:
            raise RuntimeError(f"no gradient reached {missing[:4]} ({len(missing)} of {len(leaves)} parameters); "
                               f"grad mode {torch.is_grad_enabled()}, loss requires_grad {losses.requires_grad}")
        for t, o, n in zip(leaves, self.offsets, self.numels):
            G[:, o: o + n].copy_(t.grad.reshape(Kc, n))
        for slot, leaf in self._param_slot:
            setattr(slot, leaf, None)
        return losses.detach()

    # ------------------------------------------------------------------
    @torch.no_grad()
    def self_check(self, reference_grads, D: int, sample_shape, num_classes: int, rtol: float = 2e-2) -> bool:
        """Compare against ``reference_grads(W, X, Y) -> G`` (the per-client functional path) on a tiny random
        problem in eval mode (dropout off; BN still uses batch statistics).  Any exception or mismatch means the
        model's ``forward`` is not layout-agnostic and the caller must stay on the vmap path."""
        amp, self.amp_dtype = self.amp_dtype, None
        try:
            g = torch.Generator().manual_seed(0)
            Kc, B = 2, 4
            flat = torch.cat([p.detach().reshape(-1).float().cpu() for p in self.task.model.parameters()
                              if p.requires_grad])
            W = (flat.unsqueeze(0) * (1.0 + 0.05 * torch.randn(Kc, D, generator=g))).to(self.device)
            X = torch.randn(Kc, B, *sample_shape, generator=g).to(self.device)
            Y = torch.randint(0, num_classes, (Kc, B), generator=g).to(self.device)
            G = torch.zeros(Kc, D, device=self.device)
            self.run(W, X, Y, G, train=False)
            R = reference_grads(W, X, Y)
            err = (G - R).norm() / R.norm().clamp_min(1e-12)
            self.check_error = float(err)
            return bool(torch.isfinite(err)) and float(err) < rtol
        except Exception as e:  # noqa: BLE001 — any failure = unsupported model
            import traceback

            self.check_error = traceback.format_exc()
            return False
        finally:
            self.amp_dtype = amp


def _identity(z):
    return z
