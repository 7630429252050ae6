"""Client callbacks.

Hook names and call points follow ``fllib/clients/callbacks.py:5-100``.  On top
of the four Python hooks, a callback may declare a **batched spec** — a small
dict describing its effect in a form the batched engine can fuse into its
kernels for all clients at once (label flip, gradient sign flip, gradient-norm
clip, update clip + noise).  A callback without a spec still works: its client
is routed through the sequential per-client path, which calls the Python hooks
exactly like the reference.
"""
import warnings
from typing import List, Optional

import torch


class ClientCallback:
    def __init__(self) -> None:
        self._client = None

    def setup(self, client, **info):
        self._client = client

    def on_train_round_begin(self) -> None:
        """Start of a local round (declared by the reference, never invoked there)."""

    def on_train_batch_begin(self, data, target):
        return data, target

    def on_backward_end(self, task):
        """After backward, before the optimizer step (gradient manipulation)."""

    def on_train_round_end(self):
        """After local training (pseudo-gradient manipulation)."""

    # -- fused equivalent -------------------------------------------------
    def batched_spec(self) -> Optional[dict]:
        """Returns the fused description of this callback, ``{}`` for a no-op,
        or ``None`` when the callback has Python-only behaviour.

        Keys understood by the engine: ``label_flip`` (num_classes),
        ``grad_scale`` (float), ``grad_clip`` (max total grad norm),
        ``update_clip`` (max update norm), ``update_noise_std`` (float).
        """
        cls = type(self)
        overridden = any(
            getattr(cls, h) is not getattr(ClientCallback, h)
            for h in ("on_train_round_begin", "on_train_batch_begin", "on_backward_end", "on_train_round_end")
        )
        return None if overridden else {}


class ClientCallbackList:
    def __init__(self, callbacks: List[ClientCallback]):
        self._callbacks = list(callbacks)
        self._client = None

    def __iter__(self):
        return iter(self._callbacks)

    def __len__(self):
        return len(self._callbacks)

    def setup(self, client, **info):
        self._client = client
        for cb in self._callbacks:
            try:
                cb.setup(client, **info)
            except TypeError as e:
                if "argument" in str(e):
                    warnings.warn(f"Please update `setup` of {cb.__class__} to accept (client, **info).",
                                  FutureWarning)
                    cb.setup()
                else:
                    raise

    def append(self, callback: ClientCallback):
        callback.setup(self._client)
        self._callbacks.append(callback)

    def on_train_round_begin(self) -> None:
        for cb in self._callbacks:
            cb.on_train_round_begin()

    def on_train_batch_begin(self, data, target):
        for cb in self._callbacks:
            data, target = cb.on_train_batch_begin(data, target)
        return data, target

    def on_backward_end(self, task):
        for cb in self._callbacks:
            cb.on_backward_end(task)

    def on_train_round_end(self):
        for cb in self._callbacks:
            cb.on_train_round_end()

    def batched_spec(self) -> Optional[dict]:
        """Merged spec of all callbacks, or None if any is Python-only."""
        merged: dict = {}
        for cb in self._callbacks:
            spec = cb.batched_spec()
            if spec is None:
                return None
            for k, v in spec.items():
                if k == "grad_scale":
                    merged[k] = merged.get(k, 1.0) * v
                elif k in ("grad_clip", "update_clip"):
                    merged[k] = min(merged.get(k, float("inf")), v)
                elif k == "update_noise_std":
                    merged[k] = (merged.get(k, 0.0) ** 2 + v ** 2) ** 0.5
                else:
                    merged[k] = v
        return merged


class ClippingCallback(ClientCallback):
    """Clips the total gradient norm before every local step
    (reference ``blades/clients/callbacks.py:6-15``)."""

    def __init__(self, clip_threshold) -> None:
        super().__init__()
        self._clip_threshold = clip_threshold

    def on_backward_end(self, task):
        torch.nn.utils.clip_grad_norm_(task.model.parameters(), self._clip_threshold)

    def batched_spec(self):
        return {"grad_clip": float(self._clip_threshold)}


class SignFlipCallback(ClientCallback):
    """``p.grad = -p.grad`` (reference ``signflip_adversary.py:8-12``)."""

    def on_backward_end(self, task):
        for p in task.model.parameters():
            if p.grad is not None:
                p.grad.data = -p.grad.data

    def batched_spec(self):
        return {"grad_scale": -1.0}


class LabelFlipCallback(ClientCallback):
    """``target -> num_classes - 1 - target`` (reference ``labelflip_adversary.py:11-13``)."""

    def __init__(self, num_classes: int = 10):
        super().__init__()
        self.num_classes = num_classes

    def on_train_batch_begin(self, data, target):
        return data, self.num_classes - 1 - target

    def batched_spec(self):
        return {"label_flip": int(self.num_classes)}
