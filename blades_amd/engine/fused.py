"""Shared-weight / per-client-gradient executor (the FedSGD fast path).

When every client of a chunk starts the round from the same weights (one local
step), forward and the data-gradient pass are ordinary big-batch ops over
``[Kc*B, ...]`` with *shared* weights; only two things are per client:

* normalisation statistics — ``PCBatchNorm2d`` reduces over one client's
  ``(B, H, W)`` only;
* parameter gradients — ``dW_c = X_c^T dY_c`` for each client, a grouped
  GEMM / grouped conv-wgrad whose result is written (scaled by ``-lr`` and the
  client's sign/clip coefficient) straight into row ``c`` of the update matrix.

No per-client weight copy is ever made.  The model is
executed through a *template* whose ``Conv2d`` / ``Linear`` / ``BatchNorm2d``
leaves are replaced by the ``PC*`` layers below; any other parameterised layer
makes ``supports()`` return False and the engine falls back to the generic
vectorised path.

Samples are ordered client-major (index = ``c*B + b``); 4-D activations are
channels-last, fp32 by default (the kernels split them exactly into bf16 planes
for the matrix cores) or bf16 in the ``bf16`` compute mode; reductions and the
per-client gradients are fp32.
"""
import copy
import os as _os
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import nn


class _WrittenSet(set):
    """ids of the parameters whose columns of the sink's block already hold a gradient (a kernel writing such a
    parameter accumulates instead of overwriting).  ``always``: every parameter counts as written — every kernel then
    ADDS its scaled result onto what the columns hold, which turns a sink pointed at the clients' weight rows with
    coefficients ``-lr * gscale`` into the SGD step itself (``FusedFedSGD.run_per_client(step_coef=...)``)."""

    always = False

    def __contains__(self, k):
        return self.always or set.__contains__(self, k)

    def has(self, k) -> bool:
        return set.__contains__(self, k)


class GradSink:
    """Routes per-client parameter gradients into rows of the update matrix."""

    def __init__(self):
        self.block: Optional[torch.Tensor] = None  # [Kc, >=D] view of U_row rows (row stride = ld)
        self.scale: Optional[torch.Tensor] = None  # [Kc] fp32 coefficient per client (e.g. -lr * sign)
        self.Kc = 0
        self.B = 0
        self.offsets: Dict[int, tuple] = {}  # id(param) -> (offset, numel)
        self._written = _WrittenSet()
        # operand planes of the hand-written convolution kernels (ops/csrc/conv_igemm.hip, conv_wgrad.hip):
        # 3 = fp32 activations, exact 3 x bf16 split (fp32 grade); 2 = fp32 activations, "bf16x3"; 1 = bf16
        self.npl = 3
        # per-client weights (several local steps per round): [Kc, >= D] matrix whose row c holds client c's flat
        # parameters; None = every client shares the model's own parameters (FedSGD)
        self.W: Optional[torch.Tensor] = None
        # True between FusedFedSGD.begin_round() / end_round(): the shared weights are constant, so every PCConv2d
        # holds ready-made operand planes (ops.conv_planes) and no chunk splits the weights again
        self.planes_valid = False

    def begin(self, block, scale, Kc, B):
        self.block, self.scale, self.Kc, self.B = block, scale, Kc, B
        self._written = _WrittenSet()

    def param(self, p):
        """The tensor a layer computes with: the shared parameter, or its per-client view ``[Kc, *p.shape]`` into
        ``W`` (no copy: client stride = row stride of ``W``, contiguous inside a client)."""
        if self.W is None or p is None:
            return p
        off, n = self.offsets[id(p)]
        shape = (self.Kc,) + tuple(p.shape)
        strides = [1] * len(p.shape)
        for i in range(len(p.shape) - 2, -1, -1):
            strides[i] = strides[i + 1] * p.shape[i + 1]
        return self.W.as_strided(shape, (self.W.stride(0),) + tuple(strides), self.W.storage_offset() + off)

    def view(self, param) -> torch.Tensor:
        off, n = self.offsets[id(param)]
        return self.block[:, off: off + n]

    def write_bias(self, param, g3: torch.Tensor):
        """``g3``: [Kc, R, C] gradient of a layer's output -> ``scale[c] * sum_r g3[c, r]`` into the bias columns of
        row c.  On the GPU one kernel reduces and writes the update rows (ops/csrc/elementwise.hip) instead of a
        library reduction plus a scaled copy."""
        if g3.is_cuda and g3.dtype in (torch.float32, torch.bfloat16) and self.block is not None:
            from blades_amd import ops

            h = ops._h()
            if hasattr(h, "bias_grad_"):
                off, n = self.offsets[id(param)]
                acc = id(param) in self._written
                h.bias_grad_(g3.reshape(-1, g3.shape[-1]), self.Kc, self.block, off, self.scale, acc)
                self._written.add(id(param))
                return
        self.write(param, g3.sum(dim=1, dtype=torch.float32))

    def write(self, param, g: torch.Tensor):
        """``g``: [Kc, *param.shape] (any float dtype) -> scaled fp32 row slices."""
        v = self.view(param)
        g = g.reshape(self.Kc, -1)
        if id(param) in self._written:  # a parameter used twice in the forward pass
            v.add_(g.float() * self.scale.unsqueeze(1))
        else:
            torch.mul(g.float(), self.scale.unsqueeze(1), out=v)
            self._written.add(id(param))


# ---------------------------------------------------------------------------
_DGRAD_AS_FWD = _os.environ.get("BLADES_DGRAD_AS_FWD", "1") != "0"
# "hip" (default): forward and data-gradient convolutions run ops/csrc/conv_igemm.hip; "miopen": the library's
# kernels (A/B comparisons only — the fp32 library path has no matrix-core kernel on gfx950)
_CONV_IMPL = _os.environ.get("BLADES_CONV_IMPL", "hip").lower()
# BatchNorm statistics from the convolution kernel's epilogue (0: separate statistics pass, for A/B runs)
_BN_STATS_IN_CONV = _os.environ.get("BLADES_BN_STATS_IN_CONV", "1") != "0"
# first-layer 3x3 convolution as a 1x1 convolution over an im2col tensor (0: padded 3-channel implicit GEMM)
_STEM_IM2COL = _os.environ.get("BLADES_STEM_IM2COL", "1") != "0"


def _stem_width(Ci: int, k: int, cuda: bool) -> int:
    """Channels per pixel of the stem's im2col tensor (16 for single-channel images on the GPU, else 32)."""
    return _hipmod().stem_width(Ci, k) if cuda else 32


def _stem_im2col(x: torch.Tensor, k: int, pad: int) -> torch.Tensor:
    """[N, Ci, H, W] -> [N, 16 | 32, H, W] (channels-last storage), channel ci*k*k + r*k + s = x[ci, h + r - pad,
    w + s - pad], zero beyond Ci*k*k: the row order of ``weight.reshape(Co, Ci*k*k)``."""
    N, Ci, H, W = x.shape
    if x.is_cuda and x.dtype == torch.float32:
        return _hipmod().stem_im2col(x, k, pad)  # ops/csrc/im2col.hip
    cols = F.unfold(x, k, padding=pad)  # [N, Ci*k*k, H*W]
    out = x.new_zeros((N, H * W, 32))
    out[:, :, : Ci * k * k] = cols.transpose(1, 2)
    return out.view(N, H, W, 32).permute(0, 3, 1, 2)


def _own_conv(x, mod) -> bool:
    """Does ops/csrc/conv_igemm.hip run this layer?  (GPU, channels-last 4-D input of the kernels' dtype.)"""
    if not (x.is_cuda and x.dim() == 4 and _CONV_IMPL == "hip"):
        return False
    npl = mod.sink.npl
    if x.dtype != (torch.bfloat16 if npl == 1 else torch.float32):
        return False
    return _hipmod().conv_igemm_supported(x, mod.weight, mod.stride, mod.padding, mod.dilation, npl)


def dgrad_same_conv(gy: torch.Tensor, w: torch.Tensor, padding) -> torch.Tensor:
    """Data gradient of a stride-1, dilation-1 convolution with odd kernel and ``padding = k // 2``:
    ``dx = conv2d(gy, flip_hw(w)^T)`` with the same padding."""
    wt = w.transpose(0, 1).flip(2, 3)
    if gy.is_cuda:
        wt = wt.contiguous(memory_format=torch.channels_last)
    return F.conv2d(gy, wt, None, 1, padding, 1, 1)


class GradJoin:
    """Meeting point of the two gradient branches of a residual block's input (main-branch first convolution and
    shortcut).  Whichever backward node runs first leaves its tensor here and contributes no gradient of its own;
    the second adds its result onto that tensor — a convolution does so in its data-gradient kernel's epilogue
    (``conv_dgrad(accumulate_into=...)``) — and returns the sum.  No elementwise add pass over the activation-sized
    tensor remains (2.6 % of the ResNet-18 round's kernel time, profiles/r2_v12_fp32_serial_kernel_stats_*)."""

    __slots__ = ("pending",)

    def __init__(self):
        self.pending = None


class BnBackHint:
    """Link from a per-client BatchNorm's output to the convolution that consumes it.  When that convolution's
    data-gradient kernel produces the *finished* gradient of the BatchNorm output (sole consumer, or the branch of a
    GradJoin that adds onto the other), its epilogue also accumulates the BatchNorm's backward sums (sum g, sum g x
    under the ReLU gate) and leaves them here; the BatchNorm backward then skips its reduction pass over the two
    activation-sized tensors (profiles/r3_*serial_kernel_stats*: no pcbn_reduce<1> row)."""

    __slots__ = ("x", "gate", "B", "pre", "dx")

    def __init__(self, x, gate, B):
        self.x, self.gate, self.B = x, gate, B
        self.pre = None
        self.dx = None


def _bn_bwd_in_dgrad() -> bool:
    return _os.environ.get("BLADES_BN_BWD_IN_DGRAD", "1") != "0"


class _JoinTap(torch.autograd.Function):
    """Identity on the shortcut branch; its backward hands the branch's gradient to the block's GradJoin."""

    @staticmethod
    def forward(ctx, x, join):
        ctx.join = join
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        join = ctx.join
        if join.pending is None:
            join.pending = g
            return None, None
        total, join.pending = join.pending, None
        return total.add_(g), None


class _PCConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, mod, want_stats=False):
        ctx.join, mod.join = mod.join, None
        ctx.sole, mod.sole = mod.sole, False
        ctx.bn_hint = getattr(x, "_bn_hint", None) if _bn_bwd_in_dgrad() else None
        w = weight.to(x.dtype)
        b = mod.bias.to(x.dtype) if mod.bias is not None else None
        mod.bn_stats = None
        if x.is_cuda and x.dim() == 4:
            # keep the whole network in NHWC: the engine's convolution, per-client BN and weight-gradient kernels are
            # channels-last (and so are the library's bf16 kernels of the fallback path).  A one-channel image is
            # "contiguous" in both layouts and PyTorch then assumes NCHW — say NHWC through the strides.
            if x.shape[1] == 1 and x.is_contiguous():
                x = x.as_strided(x.shape, (x.stride(0), 1, x.stride(2), x.stride(3)))
            else:
                x = x.contiguous(memory_format=torch.channels_last)
            w = w.contiguous(memory_format=torch.channels_last)
        ctx.mod = mod
        ctx.own = _own_conv(x, mod)
        ctx.stem = False
        Ci, kh, kw = int(weight.shape[-3]), int(weight.shape[-2]), int(weight.shape[-1])
        if (ctx.own and _STEM_IM2COL and mod.sink.npl > 1 and not x.requires_grad and Ci <= 3 and kh == kw == 3
                and list(mod.stride) == [1, 1] and list(mod.padding) == [1, 1] and mod.sink.Kc
                and x.shape[0] % mod.sink.Kc == 0):
            # Stem: K = 27 of the 9 x 32 padded channels of the implicit GEMM are real.  As a 1x1 convolution over
            # the im2col tensor the forward pass is one K-step instead of nine and the weight gradient a plain
            # [pixels x 64]^T [pixels x 32] GEMM (a quarter of the MFMAs), for one extra 128-byte-per-pixel tensor.
            col = _stem_im2col(x, 3, 1)
            planes = mod.planes("stem")
            if planes is not None:
                w2 = mod._planes["stem_w"]
            else:
                wk = mod.sink.param(weight).detach()
                wd = int(col.shape[1])
                w2 = F.pad(wk.reshape(*wk.shape[:-3], Ci * 9), (0, wd - Ci * 9)).reshape(*wk.shape[:-3], wd, 1, 1)
            ctx.stem = True
            ctx.save_for_backward(col, w2)
            if not want_stats and _hipmod().stem_valu_ok(col, int(weight.shape[-4]), 1 if w2.dim() == 4 else mod.sink.Kc):
                # single-channel first layer without a BatchNorm behind it (the Fashion-MNIST / MNIST CNN): 512 FMAs
                # against 192 bytes per pixel — a streaming fp32 kernel on the vector ALUs (ops/csrc/stem_valu.hip)
                return _hipmod().stem_forward_valu(col, w2, mod.sink.param(mod.bias))
            if want_stats and _BN_STATS_IN_CONV:
                y, mod.bn_stats = _hipmod().conv_forward(col, w2, mod.sink.param(mod.bias), 1, 0, mod.sink.npl,
                                                         stats_B=x.shape[0] // mod.sink.Kc, planes=planes)
                return y
            return _hipmod().conv_forward(col, w2, mod.sink.param(mod.bias), 1, 0, mod.sink.npl, planes=planes)
        if ctx.own:
            # hand-written implicit-GEMM kernel; reads the fp32 master weights itself (split into bf16 planes) —
            # the shared ones, or each client's own set when the sink carries per-client weights
            wk = mod.sink.param(weight).detach()
            ctx.save_for_backward(x, wk)
            Kc = mod.sink.Kc
            planes = mod.planes("fwd")
            if want_stats and mod.sink.npl > 1 and Kc and x.shape[0] % Kc == 0 and _BN_STATS_IN_CONV:
                # the BatchNorm that follows gets its per-client sums from this kernel's epilogue
                y, mod.bn_stats = _hipmod().conv_forward(x, wk, mod.sink.param(mod.bias), int(mod.stride[0]),
                                                         int(mod.padding[0]), mod.sink.npl, stats_B=x.shape[0] // Kc,
                                                         planes=planes)
                return y
            return _hipmod().conv_forward(x, wk, mod.sink.param(mod.bias), int(mod.stride[0]), int(mod.padding[0]),
                                          mod.sink.npl, planes=planes)
        if mod.sink.W is not None:
            raise NotImplementedError("per-client weights need the hand-written convolution kernels")
        ctx.save_for_backward(x, w)
        return F.conv2d(x, w, b, mod.stride, mod.padding, mod.dilation, 1)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        mod = ctx.mod
        sink: GradSink = mod.sink
        Kc, B = sink.Kc, sink.B
        gy = gy.contiguous(memory_format=torch.channels_last) if gy.dim() == 4 else gy
        dx = None
        if ctx.stem:
            off, _ = sink.offsets[id(mod.weight)]
            acc = id(mod.weight) in sink._written
            if _hipmod().stem_valu_ok(x, int(gy.shape[1]), Kc):
                ok = _hipmod().stem_wgrad_valu_(x, gy, Kc, sink.block, off, sink.scale, acc, int(mod.weight.shape[1]) * 9)
            else:
                ok = _hipmod().conv_wgrad_(x, gy, Kc, sink.block, off, sink.scale, acc, 1, 1, 0, sink.npl,
                                           cin_store=int(mod.weight.shape[1]) * 9)
            if not ok:
                raise RuntimeError("stem weight gradient: the 1x1 kernel rejected the im2col problem")
            sink._written.add(id(mod.weight))
            if mod.bias is not None:
                sink.write_bias(mod.bias, gy.permute(0, 2, 3, 1).reshape(Kc, -1, gy.shape[1]))
            return None, None, None, None
        join = ctx.join
        if ctx.needs_input_grad[0] and ctx.own:
            other = None
            if join is not None and join.pending is not None:
                other, join.pending = join.pending, None
                if not (other.dtype == x.dtype and other.shape == x.shape
                        and other.is_contiguous(memory_format=torch.channels_last)):
                    other = other.to(x.dtype).contiguous(memory_format=torch.channels_last)
            hint = ctx.bn_hint
            # this kernel writes the finished gradient of x when x has no other consumer, or when it adds onto the
            # other branch's tensor: then it can also do the first pass of the BatchNorm backward behind x
            final = hint is not None and (other is not None or (join is None and ctx.sole))
            if final and hint.x.shape == x.shape and Kc and x.shape[0] % Kc == 0:
                dx, hint.pre = _hipmod().conv_dgrad(gy, w, (int(x.shape[2]), int(x.shape[3])), int(mod.stride[0]),
                                                    int(mod.padding[0]), sink.npl, planes=mod.planes("dgrad"),
                                                    accumulate_into=other, bn_bwd=(hint.x, hint.gate, hint.B))
                hint.dx = dx
            else:
                dx = _hipmod().conv_dgrad(gy, w, (int(x.shape[2]), int(x.shape[3])), int(mod.stride[0]),
                                          int(mod.padding[0]), sink.npl, planes=mod.planes("dgrad"),
                                          accumulate_into=other)
            if join is not None and other is None:
                join.pending, dx = dx, None  # first of the two branches: the other one adds onto this tensor
        elif ctx.needs_input_grad[0]:
            ks = w.shape[2:]
            same = (gy.is_cuda and _DGRAD_AS_FWD and tuple(mod.stride) == (1, 1) and tuple(mod.dilation) == (1, 1)
                    and ks[0] % 2 == 1 and ks[1] % 2 == 1
                    and tuple(mod.padding) == (ks[0] // 2, ks[1] // 2))
            if same:
                # stride-1 "same" convolution: the data gradient is itself a forward convolution of gy with the
                # spatially flipped, in/out-transposed filter.  MIOpen's forward kernels (CK xdl) run this
                # shape ~1.4x faster than its backward-data igemm and need no zero-fill of dx beforehand.
                dx = dgrad_same_conv(gy, w, mod.padding)
            else:
                dx = torch.ops.aten.convolution_backward(
                    gy, x, w, None, mod.stride, mod.padding, mod.dilation, False, [0, 0], 1,
                    [True, False, False])[0]
            if join is not None:
                if join.pending is None:
                    join.pending, dx = dx, None
                else:
                    other, join.pending = join.pending, None
                    dx = other.add_(dx)
        if not mod.wgrad_hip(x, gy, Kc):
            sink.write(mod.weight, mod.wgrad(x, gy, Kc, B))  # [Kc, Cout, Cin, kh, kw]
        if mod.bias is not None:
            # per-client bias gradient: fp32 accumulation straight from the (NHWC) bf16 tensor, no fp32 copy
            g = gy.permute(0, 2, 3, 1) if gy.dim() == 4 else gy.unsqueeze(1)
            sink.write_bias(mod.bias, g.reshape(Kc, -1, gy.shape[1]))
        return dx, None, None, None


class PCConv2d(nn.Module):
    def __init__(self, conv: nn.Conv2d, sink: GradSink):
        super().__init__()
        assert conv.groups == 1 and conv.padding_mode == "zeros"
        self.weight, self.bias = conv.weight, conv.bias
        self.stride, self.padding, self.dilation = list(conv.stride), list(conv.padding), list(conv.dilation)
        self.sink = sink
        self.bn_stats = None  # (partial, nchunk) of the last forward(..., want_stats=True), consumed by the BN
        self.join = None      # GradJoin of the residual block whose input this convolution reads (set per forward)
        self.sole = False     # this convolution is the only consumer of its input (set per forward)
        self._planes = {}     # "fwd" / "dgrad" / "stem": ops.conv_planes images of the shared weight, per round

    def forward(self, x, want_stats: bool = False, join=None, sole: bool = False):
        self.join, self.sole = join, sole
        return _PCConvFn.apply(x, self.weight, self, want_stats)

    def prepare_planes(self):
        """Splits the shared weight into the convolution kernels' bf16 operand planes — once per round, for every
        chunk's forward and data-gradient launches (reference analogue: the weights a Ray worker receives once per
        round, fllib/core/execution/worker_group.py:74-83)."""
        w = self.weight.detach()
        if not (w.is_cuda and w.dtype == torch.float32 and w.dim() == 4 and _CONV_IMPL == "hip"):
            return
        h, npl, pl = _hipmod(), self.sink.npl, self._planes
        w = w.contiguous()
        Co, Ci, kh, kw = (int(v) for v in w.shape)
        if kh * kw > 25:
            return
        pl["fwd"] = h.conv_planes(w, 0, npl, out=pl.get("fwd"))
        pl["dgrad"] = h.conv_planes(w, 1, npl, out=pl.get("dgrad"))
        if (_STEM_IM2COL and npl > 1 and Ci <= 3 and kh == kw == 3 and list(self.stride) == [1, 1]
                and list(self.padding) == [1, 1]):
            w2 = pl.get("stem_w")
            wd = _stem_width(Ci, 3, True)
            if w2 is None:
                w2 = pl["stem_w"] = torch.zeros((Co, wd, 1, 1), device=w.device, dtype=torch.float32)
            w2.view(Co, wd)[:, : Ci * 9].copy_(w.reshape(Co, Ci * 9))
            pl["stem"] = h.conv_planes(w2, 0, npl, out=pl.get("stem"))

    def planes(self, kind: str):
        if not self.sink.planes_valid or self.sink.W is not None:
            return None
        return self._planes.get(kind)

    def wgrad_hip(self, x, gy, Kc) -> bool:
        """MFMA kernel path: dW_c lands, scaled, in the update-matrix rows.  False = not applicable."""
        if not x.is_cuda or _os.environ.get("BLADES_DISABLE_HIP_WGRAD") == "1":
            return False
        h = _hipmod()
        ks = self.weight.shape[2:]
        if not h.conv_wgrad_supported(x, gy, ks, self.stride, self.padding, self.dilation):
            if (tuple(self.dilation) == (1, 1) and x.dim() == 4 and x.dtype == gy.dtype
                    and x.dtype in (torch.float32, torch.bfloat16) and hasattr(h, "conv_wgrad_im2col_")
                    and _os.environ.get("BLADES_WGRAD_IM2COL", "1") != "0"):
                # no tile schedule for this geometry (4x4 / stride 4 patch embedding, 5x5 / 7x7 tokenizers, ...):
                # dW_c = gy_c^T im2col(x_c) on the grouped-GEMM kernel instead of MIOpen's grouped weight gradient
                sink = self.sink
                off, _ = sink.offsets[id(self.weight)]
                acc = id(self.weight) in sink._written
                h.conv_wgrad_im2col_(x, gy, Kc, sink.block, off, sink.scale, acc, ks, self.stride, self.padding,
                                     sink.npl)
                sink._written.add(id(self.weight))
                return True
            return False
        x = x.contiguous(memory_format=torch.channels_last)
        sink = self.sink
        off, _ = sink.offsets[id(self.weight)]
        acc = id(self.weight) in sink._written
        ok = h.conv_wgrad_(x, gy, Kc, sink.block, off, sink.scale, acc, int(ks[0]), int(self.stride[0]),
                           int(self.padding[0]), sink.npl)
        if not ok and int(self.stride[0]) == 2 and int(ks[0]) == 3 and int(self.padding[0]) == 1 \
                and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            # The split planes of a strided 3x3 layer's input window do not fit in LDS.  Its weight gradient is
            # the stride-1 weight gradient against gy spread onto the input grid (zeros in between): same
            # kernel, fast path, at 4x the MFMA work of these three (of twenty) layers.
            N, Co, Ho, Wo = gy.shape
            up = torch.empty((N, Co, 2 * Ho, 2 * Wo), dtype=gy.dtype, device=gy.device,
                             memory_format=torch.channels_last).zero_()
            up[:, :, ::2, ::2] = gy
            ok = h.conv_wgrad_(x, up, Kc, sink.block, off, sink.scale, acc, 3, 1, 1, sink.npl)
        if ok:
            sink._written.add(id(self.weight))
        return ok

    def wgrad(self, x, gy, Kc, B):
        """Per-client weight gradient.  Reference implementation: regroup to
        ``[B, Kc*C, H, W]`` and use the grouped-conv weight-gradient (one group per
        client).  The HIP kernel (ops/csrc/conv_wgrad.hip) replaces this on the GPU."""
        N, Ci, H, W = x.shape
        Co = gy.shape[1]
        xg = x.reshape(Kc, B, Ci, H, W).transpose(0, 1).reshape(B, Kc * Ci, H, W)
        gg = gy.reshape(Kc, B, Co, gy.shape[2], gy.shape[3]).transpose(0, 1).reshape(B, Kc * Co, gy.shape[2],
                                                                                     gy.shape[3])
        if x.is_cuda:
            xg = xg.contiguous(memory_format=torch.channels_last)
            gg = gg.contiguous(memory_format=torch.channels_last)
        wshape = [Kc * Co, Ci, self.weight.shape[2], self.weight.shape[3]]
        wfake = x.new_empty(wshape)
        dw = torch.ops.aten.convolution_backward(
            gg, xg, wfake, None, self.stride, self.padding, self.dilation, False, [0, 0], Kc,
            [False, True, False])[1]
        return dw.reshape(Kc, Co, Ci, self.weight.shape[2], self.weight.shape[3])


_LINEAR_IMPL = _os.environ.get("BLADES_LINEAR_IMPL", "hip").lower()  # "lib": hipBLASLt / bmm, for A/B runs


def _own_linear(x, mod) -> bool:
    """Does the framework's own MFMA kernel run this Linear layer?  (ops/_hip.py linear_forward: conv_igemm on the
    [M, K, 1, 1] view — fp32 tensors as exact bf16 planes, shared or per-client weights.)"""
    if not (x.is_cuda and _LINEAR_IMPL == "hip"):
        return False
    npl = mod.sink.npl
    if x.dtype != (torch.bfloat16 if npl == 1 else torch.float32):
        return False
    return _hipmod().linear_supported(x, mod.weight, npl)


class _PCLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, mod):
        ctx.mod = mod
        ctx.own = _own_linear(x, mod)
        if ctx.own:
            # y = x W^T + b on the matrix cores, fp32 grade; with per-client weights row m uses client m // R's set
            w = mod.sink.param(weight).detach()
            ctx.save_for_backward(x, w)
            return _hipmod().linear_forward(x, w, mod.sink.param(mod.bias), mod.sink.npl, planes=mod.planes("fwd"))
        if mod.sink.W is not None:
            # per-client weights: one strided-batched library GEMM over the clients
            Kc = mod.sink.Kc
            w = mod.sink.param(weight).detach().to(x.dtype)  # [Kc, out, in]
            ctx.save_for_backward(x, w)
            y = torch.bmm(x.reshape(Kc, -1, x.shape[-1]), w.transpose(1, 2))
            if mod.bias is not None:
                y = y + mod.sink.param(mod.bias).detach().to(x.dtype).unsqueeze(1)
            return y.reshape(*x.shape[:-1], w.shape[1])
        w = weight.to(x.dtype)
        ctx.save_for_backward(x, w)
        return F.linear(x, w, mod.bias.to(x.dtype) if mod.bias is not None else None)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        mod = ctx.mod
        sink: GradSink = mod.sink
        Kc = sink.Kc
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.own:
                dx = _hipmod().linear_dgrad(gy.to(x.dtype), w, sink.npl, planes=mod.planes("dgrad")).reshape(x.shape)
            elif w.dim() == 3:
                dx = torch.bmm(gy.reshape(Kc, -1, gy.shape[-1]), w).reshape(x.shape)
            else:
                dx = gy @ w
        g3 = gy.reshape(Kc, -1, gy.shape[-1])  # [Kc, B*, out]
        x3 = x.reshape(Kc, -1, x.shape[-1])  # [Kc, B*, in]
        if x.is_cuda and x.dtype == gy.dtype and x.dtype in (torch.float32, torch.bfloat16) \
                and _os.environ.get("BLADES_DISABLE_HIP_WGRAD") != "1":
            # grouped GEMM kernel: coef[c] * gy_c^T x_c lands in the update-matrix rows (no [Kc, out, in] temp)
            off, _ = sink.offsets[id(mod.weight)]
            acc = id(mod.weight) in sink._written
            _hipmod().linear_wgrad_(x3.reshape(-1, x.shape[-1]), g3.reshape(-1, gy.shape[-1]), Kc, sink.block, off,
                                    sink.scale, acc, sink.npl)
            sink._written.add(id(mod.weight))
        else:
            sink.write(mod.weight, torch.bmm(g3.transpose(1, 2), x3))
        if mod.bias is not None:
            sink.write_bias(mod.bias, g3)
        return dx, None, None


class PCLinear(nn.Module):
    def __init__(self, lin: nn.Linear, sink: GradSink):
        super().__init__()
        self.weight, self.bias, self.sink = lin.weight, lin.bias, sink
        self._planes = {}

    def forward(self, x):
        return _PCLinearFn.apply(x, self.weight, self)

    def prepare_planes(self):
        """Operand planes of the shared weight for this round's forward / data-gradient launches (see PCConv2d)."""
        w = self.weight.detach()
        if not (w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and _LINEAR_IMPL == "hip"):
            return
        h, npl, pl = _hipmod(), self.sink.npl, self._planes
        w4 = w.contiguous().view(w.shape[0], w.shape[1], 1, 1)
        pl["fwd"] = h.conv_planes(w4, 0, npl, out=pl.get("fwd"))
        pl["dgrad"] = h.conv_planes(w4, 1, npl, out=pl.get("dgrad"))

    def planes(self, kind: str):
        if not self.sink.planes_valid or self.sink.W is not None:
            return None
        return self._planes.get(kind)


def _bn_forward_ref(x, res, gamma, beta, Kc, eps, relu):
    N, C, H, W = x.shape
    B = N // Kc
    xf = x.float().reshape(Kc, B, C, H, W)
    var, mean = torch.var_mean(xf, dim=(1, 3, 4), unbiased=False, keepdim=True)
    invstd = torch.rsqrt(var + eps)
    y = (xf - mean) * invstd
    if gamma is not None:
        y = y * gamma.float().view(1, 1, C, 1, 1) + beta.float().view(1, 1, C, 1, 1)
    y = y.reshape(N, C, H, W)
    if res is not None:
        y = y + res.float()
    if relu:
        y = torch.relu(y)
    y = y.to(x.dtype)
    if x.is_cuda:
        y = y.contiguous(memory_format=torch.channels_last)
    return y, mean.reshape(Kc, C), invstd.reshape(Kc, C)


def _bn_backward_ref(x, gy, y, gamma, mean, invstd, Kc, relu, want_gres):
    N, C, H, W = x.shape
    B = N // Kc
    g = gy.float()
    if relu:
        g = g * (y > 0)
    g5 = g.reshape(Kc, B, C, H, W)
    xhat = (x.float().reshape(Kc, B, C, H, W) - mean.view(Kc, 1, C, 1, 1)) * invstd.view(Kc, 1, C, 1, 1)
    dbeta = g5.sum(dim=(1, 3, 4))
    dgamma = (g5 * xhat).sum(dim=(1, 3, 4))
    gam = gamma.float().view(1, C) if gamma is not None else torch.ones(1, C, device=x.device)
    m = float(B * H * W)
    dx = (gam * invstd).view(Kc, 1, C, 1, 1) * (g5 - (dbeta / m).view(Kc, 1, C, 1, 1)
                                               - xhat * (dgamma / m).view(Kc, 1, C, 1, 1))
    dx = dx.reshape(N, C, H, W).to(x.dtype)
    gres = g.to(x.dtype) if want_gres else None
    if x.is_cuda:
        dx = dx.contiguous(memory_format=torch.channels_last)
        if gres is not None:
            gres = gres.contiguous(memory_format=torch.channels_last)
    return dx, gres, dgamma, dbeta


class _PCBatchNormFn(torch.autograd.Function):
    """Per-client batch-statistics BN with fused residual add and ReLU.

    x (and residual): [Kc*B, C, H, W]; statistics over one client's (B, H, W).
    GPU bf16 channels-last inputs run the HIP kernels of ops/csrc/pcbn.hip (which also write
    the per-client dgamma/dbeta straight into the update-matrix rows); everything else runs the
    torch reference above."""

    @staticmethod
    def forward(ctx, x, res, anchor, mod, relu, pre_stats=None):
        sink: GradSink = mod.sink
        Kc = sink.Kc
        use_hip = x.is_cuda and _hipmod().pcbn_supported(x) and (res is None or _hipmod().pcbn_supported(res))
        if use_hip:
            # the kernel leaves the ReLU gate as a bitmask (1/16 of y's bytes) for the backward pass
            y, mean, invstd, gate = _hipmod().pcbn_forward(x, res, sink.param(mod.weight), sink.param(mod.bias),
                                                           Kc, mod.eps, relu, want_mask=True, pre_stats=pre_stats)
        elif sink.W is not None:
            raise NotImplementedError("per-client weights need the hand-written batch-norm kernels")
        else:
            y, mean, invstd = _bn_forward_ref(x, res, mod.weight, mod.bias, Kc, mod.eps, relu)
            gate = y if relu else None
        ctx.mod, ctx.relu, ctx.has_res, ctx.use_hip = mod, relu, res is not None, use_hip
        ctx.save_for_backward(x, gate, mean, invstd)
        ctx.hint = None
        if use_hip and mod.want_hint and _bn_bwd_in_dgrad() and x.dtype == torch.float32 and Kc:
            ctx.hint = mod.last_hint = BnBackHint(x, gate, x.shape[0] // Kc)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, mean, invstd = ctx.saved_tensors
        mod = ctx.mod
        sink: GradSink = mod.sink
        Kc = sink.Kc
        want_gres = ctx.has_res and ctx.relu
        if ctx.use_hip:
            gy = gy.contiguous(memory_format=torch.channels_last)
            U, offg, offb, acc = None, 0, 0, False
            if mod.weight is not None:
                U = sink.block
                offg, offb = sink.offsets[id(mod.weight)][0], sink.offsets[id(mod.bias)][0]
                acc = id(mod.weight) in sink._written
                sink._written.add(id(mod.weight))
                sink._written.add(id(mod.bias))
            pre, hint = None, ctx.hint
            if hint is not None:
                if hint.pre is not None and hint.dx is not None and hint.dx.data_ptr() == gy.data_ptr():
                    pre = hint.pre
                hint.pre = hint.dx = hint.x = hint.gate = None
            dx, gres = _hipmod().pcbn_backward(x, gy, y, sink.param(mod.weight), mean, invstd, Kc, sink.scale, U,
                                               offg, offb, acc, want_gres, pre_stats=pre)
        else:
            dx, gres, dgamma, dbeta = _bn_backward_ref(x, gy, y, mod.weight, mean, invstd, Kc, ctx.relu, want_gres)
            if mod.weight is not None:
                sink.write(mod.weight, dgamma)
                sink.write(mod.bias, dbeta)
        if ctx.has_res and not ctx.relu:
            gres = gy
        return dx, (gres if ctx.has_res else None), None, None, None, None


def _hipmod():
    from blades_amd import ops

    return ops._h()


# ---------------------------------------------------------------------------
# Transformer pieces (CCT / CVT / ViT-lite, reference fllib/models/backbones/cctnets): with shared weights the
# attention itself has no parameters, so only LayerNorm's affine pair and the broadcast embeddings need a
# per-client gradient; the token mixing runs as one big batch over all clients of the chunk.
class _PCLayerNormFn(torch.autograd.Function):
    """LayerNorm over the last dimension with per-client affine gradients.  Written with plain reductions:
    ATen's row-moments kernel needs 0.56 ms for the 196k x 128 rows of a CCT chunk on gfx950, five fused
    elementwise/reduce passes need less than half of that."""

    @staticmethod
    def forward(ctx, x, mod):
        if len(mod.normalized_shape) != 1:
            raise NotImplementedError("PCLayerNorm normalises the last dimension only")
        ctx.hip = x.is_cuda and _hipmod().ln_supported(x)
        if ctx.hip:  # one-wave-per-row kernel (ops/csrc/layernorm.hip); backward recomputes xhat from x
            # (per-client weights, local steps > 1: gamma / beta are [Kc, C] views into the clients' weight rows)
            gamma = mod.sink.param(mod.weight)
            y, mean, rstd = _hipmod().ln_forward(x, gamma, mod.sink.param(mod.bias), mod.eps)
            ctx.mod = mod
            ctx.gamma = gamma.detach() if gamma is not None else None
            ctx.save_for_backward(x, mean, rstd)
            return y
        if mod.sink.W is not None:
            raise NotImplementedError("per-client LayerNorm weights need the hand-written kernel (GPU)")
        xf = x.float()
        var, mean = torch.var_mean(xf, dim=-1, unbiased=False, keepdim=True)
        rstd = torch.rsqrt(var + mod.eps)
        xhat = (xf - mean) * rstd
        y = xhat
        if mod.weight is not None:
            y = y * mod.weight.float()
        if mod.bias is not None:
            y = y + mod.bias.float()
        ctx.mod = mod
        ctx.save_for_backward(xhat.to(x.dtype), rstd)
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, gy):
        mod = ctx.mod
        sink: GradSink = mod.sink
        Kc = sink.Kc
        if ctx.hip:
            x, mean, rstd = ctx.saved_tensors
            offg = offb = -1
            acc = False
            if mod.weight is not None:
                offg = sink.offsets[id(mod.weight)][0]
                acc = id(mod.weight) in sink._written
                sink._written.add(id(mod.weight))
            if mod.bias is not None:
                offb = sink.offsets[id(mod.bias)][0]
                sink._written.add(id(mod.bias))
            dx = _hipmod().ln_backward(x, gy.to(x.dtype), ctx.gamma, mean, rstd, Kc, sink.scale, sink.block, offg,
                                       offb, acc)
            return dx, None
        xhat, rstd = ctx.saved_tensors
        xh, g = xhat.float(), gy.float()
        C = xh.shape[-1]
        if mod.weight is not None:
            sink.write(mod.weight, (g * xh).reshape(Kc, -1, C).sum(dim=1))
            if mod.bias is not None:
                sink.write(mod.bias, g.reshape(Kc, -1, C).sum(dim=1))
            g = g * mod.weight.float()
        dx = rstd * (g - g.mean(dim=-1, keepdim=True) - xh * (g * xh).mean(dim=-1, keepdim=True))
        return dx.to(gy.dtype), None


class PCLayerNorm(nn.Module):
    def __init__(self, ln: nn.LayerNorm, sink: GradSink):
        super().__init__()
        self.weight, self.bias = ln.weight, ln.bias
        self.normalized_shape, self.eps, self.sink = tuple(ln.normalized_shape), ln.eps, sink

    def forward(self, x):
        return _PCLayerNormFn.apply(x, self)


class _MHAFn(torch.autograd.Function):
    """Fused attention core (ops/csrc/mha.hip): scores, softmax, client-keyed dropout and the value product of every
    (sample, head) in one kernel; backward recomputes the probabilities from the saved row log-sum-exp."""

    @staticmethod
    def forward(ctx, qkv, B, scale, drop, npl):
        h = _hipmod()
        uids, tag, seed, layer, p = drop if drop is not None else (None, None, 0, 0, 0.0)
        o, lse = h.mha_forward(qkv, B, scale, uids, tag, seed, layer, p, npl)
        ctx.save_for_backward(qkv, o, lse)
        ctx.cfg = (B, scale, uids, tag, seed, layer, p, npl)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, lse = ctx.saved_tensors
        B, scale, uids, tag, seed, layer, p, npl = ctx.cfg
        return _hipmod().mha_backward(qkv, o, d_o, lse, B, scale, uids, tag, seed, layer, p, npl), None, None, None, None


class PCAttention(nn.Module):
    """``cctnets.layers.Attention`` with the softmax(q k^T) v core on the fused kernel.  Child names and order are
    the original's (qkv, attn_drop, proj, proj_drop), so ``swap_dropout`` numbers the dropout layers identically in
    every structural copy of the model and the mask keys match the other execution paths."""

    def __init__(self, attn: nn.Module, sink: GradSink):
        super().__init__()
        self.num_heads, self.scale, self.sink = attn.num_heads, attn.scale, sink
        self.qkv = attn.qkv
        self.attn_drop = attn.attn_drop
        self.proj = attn.proj
        self.proj_drop = attn.proj_drop

    def _drop_cfg(self):
        d = self.attn_drop
        p = float(getattr(d, "p", 0.0))
        if not self.training or p <= 0.0:
            return None, True
        ctx = getattr(d, "ctx", None)
        if ctx is None:  # a plain nn.Dropout (no engine context): torch path keeps its semantics
            return None, False
        if not ctx.enabled or ctx.uids is None:
            return None, True
        if ctx.layout != "major" or getattr(d, "kind", "elem") != "elem":
            return None, False
        return (ctx.uids, ctx.tag, ctx.seed, d.layer, p), True

    def forward(self, x):
        Bn, N, C = x.shape
        H = self.num_heads
        qkv = self.qkv(x)
        drop, ok = self._drop_cfg()
        q5 = qkv.reshape(Bn, N, 3, H, C // H)
        if ok and x.is_cuda and _os.environ.get("BLADES_DISABLE_HIP_MHA") != "1" and _hipmod().mha_supported(q5) \
                and Bn % max(self.sink.Kc, 1) == 0:
            o = _MHAFn.apply(q5, Bn // max(self.sink.Kc, 1), float(self.scale), drop, self.sink.npl)
        else:
            q, k, v = q5.permute(2, 0, 3, 1, 4)
            attn = self.attn_drop(((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1))
            o = (attn @ v).transpose(1, 2).reshape(Bn, N, C)
        return self.proj_drop(self.proj(o))


class _PCBroadcastFn(torch.autograd.Function):
    """``p`` ([1, ...], shared by every sample) expanded over the batch; backward leaves each client's sum of the
    incoming gradient in its update row."""

    @staticmethod
    def forward(ctx, p, proxy, n, dtype):
        ctx.proxy = proxy
        sink: GradSink = proxy.sink
        if sink.W is not None:
            # per-client weights (local steps > 1): sample row c*B + b gets client c's copy of the embedding
            pc = sink.param(p).detach().to(dtype)  # [Kc, 1, ...]
            Kc = pc.shape[0]
            if n % Kc:
                raise NotImplementedError("broadcast parameter: batch is not a multiple of the number of clients")
            return pc.expand(Kc, n // Kc, *p.shape[1:]).reshape(n, *p.shape[1:])
        return p.to(dtype).expand(n, *p.shape[1:])

    @staticmethod
    def backward(ctx, g):
        proxy = ctx.proxy
        sink: GradSink = proxy.sink
        sink.write(proxy.param, g.float().reshape(sink.Kc, -1, *g.shape[1:]).sum(dim=1))
        return None, None, None, None


class PCParam:
    """Stands in for a bare ``nn.Parameter`` of shape ``[1, ...]`` that a model broadcasts over its batch
    (positional / class embeddings): supports ``x + p``, ``p + x`` and ``p.expand(n, -1, ...)``."""

    def __init__(self, param: nn.Parameter, sink: GradSink):
        self.param, self.sink = param, sink
        self.shape = param.shape

    def _bcast(self, n, dtype):
        return _PCBroadcastFn.apply(self.param, self, int(n), dtype)

    def __radd__(self, x):
        return x + self._bcast(x.shape[0], x.dtype)

    __add__ = __radd__

    def expand(self, *sizes):
        sizes = sizes[0] if len(sizes) == 1 and isinstance(sizes[0], (tuple, list)) else sizes
        return self._bcast(sizes[0], self.param.dtype)


class _MaxPoolFn(torch.autograd.Function):
    """Channels-last max-pool on ops/csrc/elementwise.hip, the preceding ReLU folded in (relu o maxpool = maxpool o
    relu); the forward pass leaves one byte per output element (window position of the maximum), the backward pass
    gathers — no ATen ``max_pool2d_with_indices`` / clamp kernels in the CNN and CCT rounds."""

    @staticmethod
    def forward(ctx, x, k, s, p, relu):
        y, idx = _hipmod().maxpool_forward(x, k, s, p, relu)
        ctx.save_for_backward(idx)
        ctx.geom = (tuple(x.shape), k, s, p)
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        shape, k, s, p = ctx.geom
        return _hipmod().maxpool_backward(gy, idx, shape, k, s, p), None, None, None, None


class PCMaxPool2d(nn.Module):
    def __init__(self, pool: nn.MaxPool2d, relu: bool):
        super().__init__()
        self.k, self.s, self.p = pool.kernel_size, pool.stride, pool.padding
        self.dilation, self.ceil_mode, self.relu = pool.dilation, pool.ceil_mode, relu

    def extra_repr(self):
        return f"kernel_size={self.k}, stride={self.s}, padding={self.p}, relu_folded={self.relu}"

    def forward(self, x):
        if x.is_cuda and x.dim() == 4 and _hipmod().maxpool_supported(x, self.k, self.s, self.p, self.dilation,
                                                                      self.ceil_mode):
            return _MaxPoolFn.apply(x, self.k, self.s, self.p, self.relu)
        if self.relu:
            x = F.relu(x)
        return F.max_pool2d(x, self.k, self.s, self.p, self.dilation, self.ceil_mode)


class PCActivation(nn.Module):
    """nn.ReLU / nn.GELU leaves of a template: the dispatching functions of ops/activations.py."""

    def __init__(self, kind: str):
        super().__init__()
        self.kind = kind

    def extra_repr(self):
        return self.kind

    def forward(self, x):
        from blades_amd.ops import activations as A

        return A.relu(x) if self.kind == "relu" else A.gelu(x)


class _GlobalAvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.hw = (int(x.shape[2]), int(x.shape[3]))
        return _hipmod().avgpool_global_forward(x)

    @staticmethod
    def backward(ctx, gy):
        return _hipmod().avgpool_global_backward(gy, ctx.hw)


def global_avg_pool(x: torch.Tensor, k: int) -> torch.Tensor:
    """``F.avg_pool2d(x, k)`` flattened to [N, C] when the window is the whole map (the ResNet head)."""
    if x.is_cuda and x.dim() == 4 and x.shape[2] == x.shape[3] == k and x.dtype in (torch.float32, torch.bfloat16):
        return _GlobalAvgPoolFn.apply(x)
    out = F.avg_pool2d(x, k)
    return out.reshape(out.size(0), -1)


class PCBatchNorm2d(nn.Module):
    def __init__(self, bn: nn.BatchNorm2d, sink: GradSink):
        super().__init__()
        self.eps = bn.eps
        self.weight = bn.weight if bn.affine else None
        self.bias = bn.bias if bn.affine else None
        self.sink = sink
        self.last_hint, self.want_hint = None, False

    def forward(self, x, residual=None, relu: bool = False, pre_stats=None):
        anchor = self.weight if self.weight is not None else x
        self.last_hint = None
        # (inside the Function grad mode is off: decide here whether a backward pass can follow at all — evaluation
        # must not keep the BatchNorm input alive through the hint)
        self.want_hint = torch.is_grad_enabled() and x.requires_grad
        y = _PCBatchNormFn.apply(x, residual, anchor, self, relu, pre_stats)
        if self.last_hint is not None:
            y._bn_hint, self.last_hint = self.last_hint, None  # read by the convolution that consumes y
        return y


def _conv_bn(conv, bn, x, residual=None, relu: bool = False, join=None, sole: bool = False):
    """``bn(conv(x))`` with the BatchNorm statistics taken from the convolution kernel's epilogue when both are
    the engine's own modules (no separate statistics pass over the convolution output)."""
    if isinstance(conv, PCConv2d) and isinstance(bn, PCBatchNorm2d):
        y = conv(x, want_stats=True, join=join, sole=sole)
        pre, conv.bn_stats = conv.bn_stats, None
        return bn(y, residual=residual, relu=relu, pre_stats=pre)
    return bn(conv(x), residual=residual, relu=relu)


class PCResBlock(nn.Module):
    """BasicBlock / Bottleneck of models/resnet.py with BN+ReLU and BN+add+ReLU fused."""

    def __init__(self, blk):
        super().__init__()
        self.blk = blk
        self.bottleneck = hasattr(blk, "conv3")

    def forward(self, x):
        b = self.blk
        # The block input's gradient has two branches (first convolution, shortcut): the second data-gradient
        # kernel to run adds onto the first one's tensor (GradJoin) instead of autograd adding the two afterwards.
        join = None
        if (torch.is_grad_enabled() and x.requires_grad and x.is_cuda and isinstance(b.conv1, PCConv2d)
                and _own_conv(x, b.conv1) and _os.environ.get("BLADES_GRAD_JOIN", "1") != "0"):
            join = GradJoin()
        if len(b.shortcut) > 0:
            sc = _conv_bn(b.shortcut[0], b.shortcut[1], x, join=join if isinstance(b.shortcut[0], PCConv2d) else None)
            if join is not None and not isinstance(b.shortcut[0], PCConv2d):
                join = None
        out = _conv_bn(b.conv1, b.bn1, x, relu=True, join=join)
        if len(b.shortcut) == 0:
            # (after conv1: the tap's backward node is then the younger one and runs first, so the convolution's
            # data-gradient kernel is the branch that does the adding)
            sc = _JoinTap.apply(x, join) if join is not None else x
        # (`out` has no consumer but the next convolution: its data-gradient kernel does the BatchNorm backward sums)
        if self.bottleneck:
            out = _conv_bn(b.conv2, b.bn2, out, relu=True, sole=True)
            return _conv_bn(b.conv3, b.bn3, out, residual=sc, relu=True, sole=True)
        return _conv_bn(b.conv2, b.bn2, out, residual=sc, relu=True, sole=True)


class PCResNet(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, x):
        n = self.net
        out = _conv_bn(n.conv1, n.bn1, x, relu=True)
        out = n.layer4(n.layer3(n.layer2(n.layer1(out))))
        return n.linear(global_avg_pool(out, 4))


# ---------------------------------------------------------------------------
def supports(model: nn.Module) -> bool:
    """True iff every parameterised leaf is a layer the fused executor handles."""
    for m in model.modules():
        own = list(m.parameters(recurse=False))
        if not own:
            continue
        if _broadcast_params(m) is not None:  # a frozen (sinusoidal) positional embedding stays a shared constant
            continue
        if any(not p.requires_grad for p in own):
            return False  # a frozen layer: the per-client layers write every parameter's gradient
        if isinstance(m, nn.Conv2d) and m.groups == 1 and m.padding_mode == "zeros" and not isinstance(
                m.padding, str):
            continue
        if isinstance(m, nn.Linear):
            continue
        if isinstance(m, nn.BatchNorm2d) and not m.track_running_stats:
            continue
        if isinstance(m, nn.LayerNorm):
            continue
        if _broadcast_params(m) is not None:
            continue
        return False
    return True


def _broadcast_params(m: nn.Module):
    """Names of the bare parameters of ``m`` when all of them are ``[1, ...]`` tensors the module broadcasts over
    its batch (the transformer classifier's positional / class embedding); ``None`` if ``m`` is anything else."""
    from blades_amd.models.cctnets.layers import TransformerClassifier

    if not isinstance(m, TransformerClassifier):
        return None
    names = [n for n, p in m.named_parameters(recurse=False)]
    if all(n in ("positional_emb", "class_emb") for n in names):
        return names
    return None


def _cct_attention():
    from blades_amd.models.cctnets.layers import Attention

    return Attention


def build_template(model: nn.Module, sink: GradSink) -> nn.Module:
    memo = {id(p): p for p in model.parameters()}
    tmpl = copy.deepcopy(model, memo)

    def swap(mod):
        if isinstance(mod, nn.Sequential):
            # ReLU directly in front of a max-pool: folded into the pooling kernel (FashionCNN, the CCT tokenizers)
            kids = list(mod.named_children())
            for (n0, c0), (n1, c1) in zip(kids, kids[1:]):
                if type(c0) is nn.ReLU and isinstance(c1, nn.MaxPool2d):
                    setattr(mod, n0, nn.Identity())
                    setattr(mod, n1, PCMaxPool2d(c1, relu=True))
        for name, child in list(mod.named_children()):
            if isinstance(child, nn.Conv2d):
                setattr(mod, name, PCConv2d(child, sink))
            elif isinstance(child, nn.Linear):
                setattr(mod, name, PCLinear(child, sink))
            elif isinstance(child, nn.BatchNorm2d):
                setattr(mod, name, PCBatchNorm2d(child, sink))
            elif isinstance(child, nn.LayerNorm):
                setattr(mod, name, PCLayerNorm(child, sink))
            elif isinstance(child, nn.MaxPool2d):
                setattr(mod, name, PCMaxPool2d(child, relu=False))
            elif type(child) is nn.ReLU:
                setattr(mod, name, PCActivation("relu"))
            elif type(child) is nn.GELU and getattr(child, "approximate", "none") == "none":
                setattr(mod, name, PCActivation("gelu"))
            elif isinstance(child, _cct_attention()):
                swap(child)  # its Linear layers first
                setattr(mod, name, PCAttention(child, sink))
            else:
                names = _broadcast_params(child)
                for pn in names or []:
                    prm = child._parameters[pn]
                    if prm is not None and prm.requires_grad:
                        del child._parameters[pn]
                        setattr(child, pn, PCParam(prm, sink))
                swap(child)

    holder = nn.Module()
    holder.root = tmpl
    swap(holder)  # also handles a model that is itself a single leaf layer
    tmpl = holder.root
    from blades_amd.models import resnet as _rn

    if isinstance(tmpl, _rn.ResNet):
        for stage in (tmpl.layer1, tmpl.layer2, tmpl.layer3, tmpl.layer4):
            for i, blk in enumerate(list(stage)):
                stage[i] = PCResBlock(blk)
        tmpl = PCResNet(tmpl)
    return tmpl


class FusedFedSGD:
    """Runs one shared-weight forward/backward for a chunk of clients and leaves
    ``scale_c * dL_c/dtheta`` in the chunk's rows of the update matrix."""

    def __init__(self, task, names, offsets, numels, amp_dtype, device, conv_planes: int = 3):
        self.task = task
        self.sink = GradSink()
        self.template = build_template(task.model, self.sink)
        from blades_amd.utils.torch_utils import named_trainable_parameters

        for (n, p), o, m in zip(named_trainable_parameters(task.model), offsets[:-1], numels):
            self.sink.offsets[id(p)] = (int(o), int(m))
        self.amp_dtype = amp_dtype
        self.device = device
        self.sink.npl = 1 if amp_dtype == torch.bfloat16 else (2 if conv_planes == 2 else 3)
        # In-place client SGD step (run_per_client(step_coef=...)): only for templates without normalisation layers —
        # the BatchNorm / LayerNorm backward kernels write the affine gradients before their apply pass has read the
        # affine weights — i.e. the MLP and the CNN, where the separate step is 7 % of a local step.
        self.inplace_step_ok = (_os.environ.get("BLADES_INPLACE_CLIENT_STEP", "1") != "0" and not any(
            isinstance(m, (PCBatchNorm2d, PCLayerNorm)) for m in self.template.modules()))
        self.pc_steps_ok = 0  # completed run_per_client calls (the in-place step is taken after a first regular one)
        from blades_amd.tasks.mnist import MNIST

        self._stock_loss = (getattr(type(task), "loss", None) is MNIST.loss
                            and _os.environ.get("BLADES_FUSED_CE", "1") != "0")

    def begin_round(self):
        """The shared weights stay as they are until ``end_round()``: split every convolution's weight into its
        operand planes now, once, on the current stream (before the chunk streams fork from it)."""
        if self.device.type != "cuda" or _os.environ.get("BLADES_ROUND_PLANES", "1") == "0":
            return
        for m in self.template.modules():
            if isinstance(m, (PCConv2d, PCLinear)):
                m.prepare_planes()
        self.sink.planes_valid = True

    def end_round(self):
        self.sink.planes_valid = False

    def run(self, X: torch.Tensor, Y: torch.Tensor, block: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
        """X [Kc, B, ...], Y [Kc, B]; ``block`` = the chunk's rows of U (row stride
        preserved); returns per-client losses [Kc] (fp32, detached)."""
        Kc, B = Y.shape
        self.sink.begin(block, scale, Kc, B)
        x = X.reshape(Kc * B, *X.shape[2:])
        if self.amp_dtype is not None:
            x = x.to(self.amp_dtype)
        if x.dim() == 4 and x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        self.template.train()
        from blades_amd.ops.activations import fused_scope

        with fused_scope():
            # the input needs no gradient: first-layer backward skips its dgrad
            logits = self.template(x)
            return self._loss_backward(logits, Y, Kc, B)

    def _loss_backward(self, logits, Y, Kc, B):
        """Per-client losses [Kc] and the backward pass of their sum.  The stock task loss (clamped mean
        cross-entropy, reference fllib/tasks/mnist.py:8-18) and its gradient come from one kernel
        (ops/csrc/elementwise.hip ce_clamp); a task with its own ``loss`` goes through autograd under ``vmap``."""
        if logits.is_cuda and self._stock_loss and logits.dim() == 2:
            lg = logits.detach().float().reshape(Kc, B, -1)
            losses, dz = _hipmod().ce_clamp(lg, Y, 0.0, 1e6)
            logits.backward(dz.reshape(logits.shape).to(logits.dtype))
            return losses
        lg = logits.float().reshape(Kc, B, -1)
        losses = torch.vmap(lambda z, y: self.task.loss(_identity, z, y))(lg, Y)
        losses.sum().backward()
        return losses.detach()


    def run_per_client(self, W: torch.Tensor, X: torch.Tensor, Y: torch.Tensor, G: Optional[torch.Tensor],
                       step_coef: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One local step with *per-client* weights (several batches per round, reference
        fllib/clients/client.py:55-61): row c of ``W`` holds client c's flat parameters; the per-client gradients
        go, unscaled, into ``G[Kc, D]``.  Same kernels as the shared-weight path — the convolution and batch-norm
        kernels pick a client's weight set by image index, Linear layers run as one strided-batched GEMM."""
        Kc, B = Y.shape
        if step_coef is not None:
            # plain SGD without momentum / clipping on a model whose backward pass reads every weight before it
            # writes that weight's gradient: the gradient kernels' epilogues add ``step_coef[c] * dW_c`` (= -lr *
            # gscale * dW) straight onto the weight rows — no gradient matrix, no separate pass over W and G
            assert self.inplace_step_ok
            self.sink.begin(W, step_coef, Kc, B)
            self.sink._written.always = True
        else:
            ones = torch.ones(Kc, device=W.device, dtype=torch.float32)
            self.sink.begin(G, ones, Kc, B)
        self.sink.W = W
        try:
            x = X.reshape(Kc * B, *X.shape[2:])
            if self.amp_dtype is not None:
                x = x.to(self.amp_dtype)
            if x.dim() == 4 and x.is_cuda:
                x = x.contiguous(memory_format=torch.channels_last)
            self.template.train()
            from blades_amd.ops.activations import fused_scope

            with torch.enable_grad(), fused_scope():
                logits = self.template(x)
                losses = self._loss_backward(logits, Y, Kc, B)
            missing = [k for k in self.sink.offsets if not self.sink._written.has(k)]
            if missing:
                raise NotImplementedError(f"{len(missing)} parameters received no per-client gradient")
            self.pc_steps_ok += 1
            return losses
        finally:
            self.sink.W = None

    @torch.no_grad()
    def forward_only(self, X: torch.Tensor) -> torch.Tensor:
        """Evaluation forward of a chunk: X [Kc, m, ...] -> logits [Kc, m, classes] (fp32).
        ``eval()`` mode (dropout off); BN still uses per-client batch statistics, like the reference's
        ``track_running_stats=False`` models do in eval."""
        Kc, m = X.shape[0], X.shape[1]
        self.sink.begin(None, None, Kc, m)
        x = X.reshape(Kc * m, *X.shape[2:])
        if self.amp_dtype is not None:
            x = x.to(self.amp_dtype)
        if x.dim() == 4 and x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        self.template.eval()
        from blades_amd.ops.activations import fused_scope

        with fused_scope():
            out = self.template(x).float().reshape(Kc, m, -1)
        self.template.train()
        return out


def _identity(z):
    return z
