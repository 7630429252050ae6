"""Batch-level image transforms (replacing torchvision in the data path).

The reference applies its ``train_transform`` to the **whole batch tensor**
(``fllib/datasets/fldataset.py:247-249``): one crop window, one flip decision
and one erase rectangle per batch, not per sample.  ``BatchAugment`` keeps that
semantic and splits the work in two steps so both execution paths share it:

* ``draw(n_batches, generator)`` — the random decisions, a small int/float table
  ``[n_batches, 10]`` (crop i,j,size, flip, erase flag + rect);
* ``apply(x, params)`` — the deterministic image op on ``[n_batches, B, C, H, W]``
  (bilinear crop-resize, flip, normalise, erase).  On the GPU this is the HIP
  kernel ``augment_batch`` (ops/csrc/augment.hip); the torch code below is its
  reference.

CIFAR-10 parameters (``fllib/datasets/cifar10.py:56-68``): RandomResizedCrop(32,
scale=(0.75,1), ratio=(1,1)), RandomHorizontalFlip(0.5), Normalize(mean,std),
RandomErasing(p=0.25, scale=(0.02,0.33), ratio=(0.3,3.3), value=0).
"""
import math
from typing import Optional, Sequence

import torch

# column layout of the parameter table
P_I, P_J, P_S, P_FLIP, P_ERASE, P_EI, P_EJ, P_EH, P_EW = range(9)
N_PARAMS = 9


class Normalize:
    def __init__(self, mean: Sequence[float], std: Sequence[float]):
        self.mean = tuple(float(m) for m in mean)
        self.std = tuple(float(s) for s in std)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        c = len(self.mean)
        shape = [1] * x.dim()
        shape[-3] = c
        mean = torch.tensor(self.mean, device=x.device, dtype=x.dtype).view(shape)
        std = torch.tensor(self.std, device=x.device, dtype=x.dtype).view(shape)
        return (x - mean) / std


class BatchAugment:
    def __init__(self, size: int = 32, scale=(0.75, 1.0), flip_p: float = 0.5, mean=None, std=None,
                 erase_p: float = 0.25, erase_scale=(0.02, 0.33), erase_ratio=(0.3, 3.3)):
        self.size = size
        self.scale = scale
        self.flip_p = flip_p
        self.normalize = Normalize(mean, std) if mean is not None else None
        self.erase_p = erase_p
        self.erase_scale = erase_scale
        self.erase_ratio = erase_ratio

    # ------------------------------------------------------------------
    N_UNIFORMS = 9

    def draw(self, n: int, generator: Optional[torch.Generator] = None, height: int = 32, width: int = 32,
             uniforms: Optional[torch.Tensor] = None):
        """Random decisions for ``n`` batches -> float32 CPU tensor [n, N_PARAMS].

        ``uniforms`` ([n, N_UNIFORMS] in [0,1)) replaces the generator: the engine passes
        counter-based draws keyed by (client, round) so augmentation does not depend on how
        clients are spread over ranks."""
        g = generator
        if uniforms is not None:
            cols = iter(uniforms.float().cpu().unbind(dim=1))
            u = lambda *s: next(cols)  # noqa: E731
        else:
            u = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
        area = height * width
        frac = self.scale[0] + (self.scale[1] - self.scale[0]) * u(n)
        side = torch.round(torch.sqrt(frac * area)).clamp(1, min(height, width))  # ratio = 1
        i = torch.floor(u(n) * (height - side + 1))
        j = torch.floor(u(n) * (width - side + 1))
        flip = (u(n) < self.flip_p).float()
        erase = (u(n) < self.erase_p).float()
        # erase rectangle (torchvision tries up to 10 aspect draws; one draw +
        # clamping keeps the marginal distribution close and is branch-free)
        e_area = area * (self.erase_scale[0] + (self.erase_scale[1] - self.erase_scale[0]) * u(n))
        lr = (math.log(self.erase_ratio[0]), math.log(self.erase_ratio[1]))
        ratio = torch.exp(lr[0] + (lr[1] - lr[0]) * u(n))
        eh = torch.round(torch.sqrt(e_area * ratio)).clamp(1, height - 1)
        ew = torch.round(torch.sqrt(e_area / ratio)).clamp(1, width - 1)
        ei = torch.floor(u(n) * (height - eh + 1))
        ej = torch.floor(u(n) * (width - ew + 1))
        return torch.stack([i, j, side, flip, erase, ei, ej, eh, ew], dim=1).float()

    def apply(self, x: torch.Tensor, params: torch.Tensor) -> torch.Tensor:
        """``x``: [n, B, C, H, W] float; ``params``: [n, N_PARAMS]."""
        if x.is_cuda:
            from blades_amd import ops

            _hip = ops._h()

            mean = self.normalize.mean if self.normalize else (0.0,) * x.shape[2]
            std = self.normalize.std if self.normalize else (1.0,) * x.shape[2]
            # a host table is validated right here without a sync; a device table must have been check()ed by
            # whoever filled it (the engine's HIP-graph path uploads a checked host table into a static buffer)
            return _hip.augment_batch(x, params, self.size, mean, std, validate=not params.is_cuda)
        return self.apply_reference(x, params)

    @staticmethod
    def check(params: torch.Tensor, height: int, width: int) -> None:
        """Host-side validity check of a parameter table (crop windows inside the image)."""
        pc = params.detach().float().cpu()
        if pc.shape[0] and (float((pc[:, P_I] + pc[:, P_S]).max()) > height
                            or float((pc[:, P_J] + pc[:, P_S]).max()) > width
                            or float(pc[:, :3].min()) < 0 or float(pc[:, P_S].min()) < 1):
            raise ValueError("augmentation table: crop window out of bounds")

    def apply_reference(self, x: torch.Tensor, params: torch.Tensor) -> torch.Tensor:
        n, B, C, H, W = x.shape
        S = self.size
        out = torch.empty(n, B, C, S, S, dtype=x.dtype, device=x.device)
        p = params.cpu()
        for b in range(n):
            i, j, s = int(p[b, P_I]), int(p[b, P_J]), int(p[b, P_S])
            crop = x[b, :, :, i: i + s, j: j + s]
            img = torch.nn.functional.interpolate(crop, size=(S, S), mode="bilinear", align_corners=False)
            if p[b, P_FLIP] > 0:
                img = img.flip(-1)
            if self.normalize is not None:
                img = self.normalize(img)
            if p[b, P_ERASE] > 0:
                ei, ej, eh, ew = (int(p[b, k]) for k in (P_EI, P_EJ, P_EH, P_EW))
                img[..., ei: ei + eh, ej: ej + ew] = 0
            out[b] = img
        return out

    def __call__(self, x: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """One batch ``[B, C, H, W]`` (sequential loader path)."""
        params = self.draw(1, generator, x.shape[-2], x.shape[-1])
        return self.apply_reference(x.unsqueeze(0), params)[0]


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x
