"""Tensor helpers (subset of ``fllib/utils/torch_utils.py`` that is in use)."""
import os
import random
from typing import Iterable, Optional, Union

import numpy as np
import torch


def set_torch_seed(seed: Optional[int]) -> None:
    """Seeds python / numpy / torch (reference ``torch_utils.py:160-177``)."""
    if seed is None:
        return
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def named_trainable_parameters(model: torch.nn.Module):
    """``(name, parameter)`` of the parameters that make up the flat model / update vector: the trainable ones, in
    ``named_parameters`` order.  Frozen parameters (a sinusoidal positional embedding, a frozen word embedding) are
    constants of the model and take no columns — the rule of the reference's flatten helpers and server
    (fllib/utils/torch_utils.py:145,194, fllib/algorithms/server.py:122-124).  The reference's *client* side still
    emits columns for them (fllib/tasks/task.py:162-168), which shifts every later parameter's slice on the server;
    here the engine, the sequential client path and the server all use this one enumeration."""
    return [(n, p) for n, p in model.named_parameters() if p.requires_grad]


def trainable_parameters(model: torch.nn.Module):
    return [p for _, p in named_trainable_parameters(model)]


def parameters_to_vector(parameters: Iterable[torch.Tensor]) -> torch.Tensor:
    return torch.cat([p.reshape(-1) for p in parameters if getattr(p, "requires_grad", True)])


def vector_to_parameters(vec: torch.Tensor, parameters: Iterable[torch.Tensor]) -> None:
    pointer = 0
    for p in parameters:
        if not getattr(p, "requires_grad", True):
            continue
        n = p.numel()
        p.data.copy_(vec[pointer: pointer + n].view_as(p))
        pointer += n


def clip_tensor_norm_(tensor: torch.Tensor, max_norm: float, norm_type: float = 2.0,
                      error_if_nonfinite: bool = False) -> torch.Tensor:
    """In-place L2 clip of one tensor; returns the tensor.

    (The reference version loops over a list and returns after the first
    element — ``torch_utils.py:281-322`` — which is only right for a single
    tensor, the only way it is called.)
    """
    total_norm = torch.linalg.vector_norm(tensor, norm_type)
    if error_if_nonfinite and not torch.isfinite(total_norm):
        raise RuntimeError("non-finite norm in clip_tensor_norm_")
    coef = torch.clamp(max_norm / (total_norm + 1e-6), max=1.0)
    tensor.mul_(coef.to(tensor.device))
    return tensor


def clip_para_norm_(parameters: Union[torch.Tensor, Iterable[torch.Tensor]], max_norm: float,
                    norm_type: float = 2.0) -> torch.Tensor:
    """In-place clip of the joint norm of parameter *values* (``:235-278``)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    params = [p for p in parameters]
    if not params:
        return torch.tensor(0.0)
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.detach(), norm_type) for p in params]),
                                     norm_type)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for p in params:
        p.detach().mul_(coef.to(p.device))
    return total


def get_device(num_gpus: float = 0, index: int = 0) -> torch.device:
    if num_gpus > 0 and torch.cuda.is_available():
        return torch.device("cuda", index)
    return torch.device("cpu")


def apply_grad_clipping(optimizer: torch.optim.Optimizer, grad_clip: Optional[float] = None) -> dict:
    """Clip the global gradient norm of every param group to ``grad_clip`` (reference
    ``fllib/utils/torch_utils.py`` keeps RLlib's helper of this name).  Returns ``{"grad_gnorm": …}``."""
    if grad_clip is None:
        return {}
    total = 0.0
    clipped = False
    for group in optimizer.param_groups:
        params = [p for p in group["params"] if p.grad is not None]
        if params:
            n = torch.nn.utils.clip_grad_norm_(params, grad_clip)
            total += float(n)
            clipped = True
    return {"grad_gnorm": total} if clipped else {}
