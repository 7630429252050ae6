"""Client-level differential privacy (reference ``blades/clients/dp_client.py``)."""
import torch

from blades_amd.types import NotProvided
from blades_amd.utils.torch_utils import clip_tensor_norm_
from .callbacks import ClientCallback
from .client import Client
from .client_config import ClientConfig


class DPCliengConfig(ClientConfig):  # (sic) name kept for API parity
    def __init__(self, client_class=None) -> None:
        super().__init__(class_specifier=client_class or DPClient)
        self.clip_threshold = 100.0
        self.noise_factor = 0

    def training(self, num_batch_per_round=NotProvided, lr=NotProvided, clip_threshold=NotProvided,
                 noise_factor=NotProvided, momentum=NotProvided) -> ClientConfig:
        super().training(num_batch_per_round, lr, momentum=momentum)
        if clip_threshold is not NotProvided:
            self.clip_threshold = clip_threshold
        if noise_factor is not NotProvided:
            self.noise_factor = noise_factor
        return self


DPClientConfig = DPCliengConfig


class _DPCallback(ClientCallback):
    """On round end: L2-clip the update to ``clip_threshold`` and add
    ``N(0, noise_factor)`` (reference ``dp_client.py:37-43``)."""

    def __init__(self, clip_threshold, noise_factor) -> None:
        super().__init__()
        self._clip_threshold = clip_threshold
        self._noise_factor = noise_factor

    def on_train_round_end(self):
        update = self._client.pseudo_grad_vec
        clip_tensor_norm_(update, max_norm=self._clip_threshold)
        if self._noise_factor > 0:
            noise = torch.normal(mean=0.0, std=float(self._noise_factor), size=update.shape).to(update.device)
            update += noise

    def batched_spec(self):
        return {"update_clip": float(self._clip_threshold), "update_noise_std": float(self._noise_factor)}


class DPClient(Client):
    def __init__(self, client_config: ClientConfig, is_malicious=False):
        super().__init__(client_config, is_malicious)
        self.add_callback(_DPCallback(client_config.clip_threshold, client_config.noise_factor))
