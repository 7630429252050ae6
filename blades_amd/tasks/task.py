"""Task = model + loss + optimizer + metrics (reference ``fllib/tasks/task.py``).

The sequential API (``train_one_batch`` / ``zero_psudo_grad`` /
``compute_psudo_grad`` / ``evaluate`` / ``get|set_weights``) keeps reference
semantics.  The batched engine only borrows ``model`` (as a stateless template)
and ``loss`` from a Task.  Metrics are computed with a fused CE-sum + top-k
routine instead of torchmetrics.
"""
import copy
import threading
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from blades_amd.models.catalog import ModelCatalog
from blades_amd.utils.annotations import DeveloperAPI, OverrideToImplementCustomLogic, is_overridden
from blades_amd.utils.from_config import force_list, from_config


def ce_sum_and_topk(logits: torch.Tensor, target: torch.Tensor, ks=(1,)):
    """(sum of CE, {k: #correct@k}) — what CustomCrossEntropy + Accuracy(top_k)
    accumulate in the reference (``task.py:32-49,104-121``)."""
    logits = logits.float()
    ce = torch.nn.functional.cross_entropy(logits, target, reduction="sum")
    maxk = min(max(ks), logits.shape[-1])
    top = logits.topk(maxk, dim=-1).indices
    hit = top.eq(target.unsqueeze(-1))
    return ce, {k: hit[..., : min(k, maxk)].any(dim=-1).sum() for k in ks}


class TaskSpec:
    def __init__(self, task_class=None, alg_config=None):
        self.task_class = task_class
        self.config = alg_config

    def __eq__(self, other):
        return isinstance(other, TaskSpec) and self.task_class == other.task_class and self.config is other.config

    def __hash__(self):
        return hash(str(self.task_class))

    def build(self, device: str) -> "Task":
        return from_config(self.task_class, None, device=device, alg_config=self.config)


@DeveloperAPI
class Task:
    def __init__(self, device: str, alg_config):
        self.config = alg_config
        self.device = device
        self._model = self._init_model().to(self.device)
        self._lock = threading.RLock()
        self._optimizers = None
        self._saved_state_dict = None
        dc = alg_config.dataset_config if hasattr(alg_config, "dataset_config") else alg_config.get(
            "dataset_config", {})
        self._num_classes = (dc or {}).get("num_classes", None)

    @property
    def model(self) -> torch.nn.Module:
        return self._model

    @property
    def top_ks(self):
        if self._num_classes:
            return tuple(range(1, self._num_classes, 2))
        return (1,)

    @staticmethod
    @OverrideToImplementCustomLogic
    def loss(model, data: torch.Tensor, target: torch.Tensor):
        raise NotImplementedError

    @OverrideToImplementCustomLogic
    def make_model(self) -> Optional[torch.nn.Module]:
        return None

    def _init_model(self):
        if is_overridden(self.make_model):
            return self.make_model()
        gm = self.config["global_model"] if not isinstance(self.config, dict) else self.config.get("global_model")
        return ModelCatalog.get_model(gm)

    # -- pseudo-gradient bookkeeping (sequential path) ----------------------
    def zero_psudo_grad(self):
        self._saved_state_dict = copy.deepcopy(self._model.state_dict())

    def compute_psudo_grad(self) -> Dict[str, torch.Tensor]:
        pseudo_grad = {}
        for name, param in self._model.named_parameters():
            if not param.requires_grad:  # frozen parameters take no columns of the update (utils.torch_utils)
                continue
            pseudo_grad[name] = param.data - self._saved_state_dict[name]
        self._model.load_state_dict(self._saved_state_dict)
        return pseudo_grad

    def train_one_batch(self, data, target, on_backward_end: Callable = None) -> float:
        self._model.train()
        for opt in self._optimizers:
            opt.zero_grad()
        loss = self.loss(self._model, data, target)
        loss.backward()
        if on_backward_end:
            on_backward_end(self)
        for opt in self._optimizers:
            opt.step()
        return loss.item()

    def evaluate(self, test_loader) -> dict:
        device = next(self._model.parameters()).device
        self._model.eval()
        ks = self.top_ks
        ce_sum = torch.zeros((), device=device)
        hits = {k: torch.zeros((), device=device, dtype=torch.long) for k in ks}
        n = 0
        with torch.no_grad():
            for data, target in test_loader:
                data, target = data.to(device), target.to(device)
                ce, h = ce_sum_and_topk(self._model(data), target, ks)
                ce_sum += ce
                for k in ks:
                    hits[k] += h[k]
                n += len(target)
        result = {"length": n, "ce_loss": (ce_sum / max(n, 1)).item()}
        if self._num_classes:
            for k in ks:
                result[f"acc_top_{k}"] = (hits[k].float() / max(n, 1)).cpu()
        return result

    @OverrideToImplementCustomLogic
    def init_optimizer(self, lr=1.0, momentum=0.0) -> List[torch.optim.Optimizer]:
        self._optimizers = force_list([torch.optim.SGD(self._model.parameters(), lr=lr, momentum=momentum)])
        return self._optimizers

    def get_optimizer_states(self):
        return [opt.state_dict() for opt in self._optimizers]

    def set_optimizer_states(self, states):
        for opt, state in zip(self._optimizers, states):
            opt.load_state_dict(state)

    @DeveloperAPI
    def get_weights(self) -> Dict[str, np.ndarray]:
        return {k: v.cpu().detach().numpy() for k, v in self._model.state_dict().items()}

    @DeveloperAPI
    def set_weights(self, weights) -> None:
        device = next(self._model.parameters()).device
        sd = {k: torch.as_tensor(v, device=device) for k, v in weights.items()}
        self._model.load_state_dict(sd)
