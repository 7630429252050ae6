"""Clients.

``BaseClient`` is the generic FL client (reference ``fllib/clients/client.py``);
``Client`` adds the malicious-proxy machinery (reference
``blades/clients/client.py``).  In this framework the *fast* path never calls
``train_one_round`` per client — the batched engine trains all clients of a
rank at once and reads only each client's *metadata* (callbacks' fused spec,
``local_training``, ``is_malicious``).  ``train_one_round`` remains the
sequential path with the reference's exact semantics: it is the test oracle,
the path for clients with Python-only callbacks, and what the
reference-restatement baseline in ``benchmarks/`` measures.
"""
import copy
from typing import Iterable, Optional, Type

import torch

from blades_amd.constants import CLIENT_UPDATE
from blades_amd.utils.from_config import force_list, from_config
from .callbacks import ClientCallback, ClientCallbackList
from .client_config import ClientConfig


class BaseClient:
    def __init__(self, client_config: ClientConfig) -> None:
        self.config = client_config
        self._train_round = 0
        callback = from_config(self.config.callbacks_config)
        self.callbacks = ClientCallbackList(force_list(callback))
        self.callbacks.setup(self)
        self.pseudo_grad_vec = None

    # -- worker session ---------------------------------------------------
    def _session(self):
        from blades_amd.parallel.session import get_session

        return get_session(self.client_id)

    def setup(self):
        self._session().task.init_optimizer(self.config.lr, self.config.momentum)

    @property
    def client_id(self):
        return self.config.uid

    @property
    def is_malicious(self):
        return False

    @property
    def local_training(self) -> bool:
        return True

    def set_callbacks(self, callbacks):
        self.callbacks = ClientCallbackList(force_list(callbacks))
        self.callbacks.setup(self)

    def add_callback(self, callback: ClientCallback):
        self.callbacks.append(callback)

    def active_callbacks(self) -> ClientCallbackList:
        return self.callbacks

    def evaluate(self, test_loader):
        return self._session().task.evaluate(test_loader)

    def train_one_round(self, data_reader: Iterable):
        """Sequential local round (reference ``fllib/clients/client.py:46-74``)."""
        self._train_round += 1
        task = self._session().task
        task.zero_psudo_grad()
        num_batches = self.config.num_batch_per_round
        running_loss = 0.0
        for _ in range(num_batches):
            data, target = next(data_reader)
            data, target = self.callbacks.on_train_batch_begin(data, target)
            running_loss += task.train_one_batch(data, target, self.callbacks.on_backward_end)
        pseudo_grad = task.compute_psudo_grad()
        self.pseudo_grad_vec = torch.cat([t.reshape(-1) for t in pseudo_grad.values()])
        self.callbacks.on_train_round_end()
        return {
            "id": self.client_id,
            CLIENT_UPDATE: self.pseudo_grad_vec,
            "avg_loss": running_loss / num_batches,
        }


class Client(BaseClient):
    """Client that can be turned malicious in place (``to_malicious``)."""

    def __init__(self, client_config: ClientConfig, is_malicious: bool = False):
        super().__init__(client_config=client_config)
        self._is_malicious = is_malicious
        self._is_under_attack = False

    @property
    def is_malicious(self):
        return self._is_malicious

    def attack(self):
        if not self.is_malicious:
            raise ValueError(
                "The client is not inherently malicious, but it needs to be transformed "
                "using the `to_malicious` method prior to launching an attack."
            )
        self._is_malicious = True

    def restore(self):
        self._is_under_attack = False

    def to_malicious(self, target_cls: Optional[Type] = None, callbacks_cls: Optional[Type] = None,
                     local_training: Optional[bool] = True):
        """Re-classes ``self`` in place into a :class:`ClientProxy` around a
        copy of itself (reference ``blades/clients/client.py:43-58``), so every
        holder of this object (client manager, adversary) sees the change."""
        benign = copy.copy(self)
        benign.callbacks = ClientCallbackList(list(self.callbacks))
        benign.callbacks.setup(benign)
        self.__class__ = ClientProxy
        ClientProxy.__init__(self, benign, target_cls, local_training=local_training)
        self.attack()
        if callbacks_cls:
            self.set_callbacks(callbacks_cls() if isinstance(callbacks_cls, type) else callbacks_cls)


class ClientProxy:
    """A client under adversarial control.  Routes calls to the malicious or
    the benign personality (reference ``blades/clients/client.py:61-118``)."""

    is_malicious = True

    def __init__(self, client: BaseClient, malicious_cls: Type = None, local_training: Optional[bool] = False):
        self._is_under_attack = False
        self._local_training = local_training
        self.benign_client = client
        if malicious_cls:
            self.malicious_client = malicious_cls(client_config=client.config)
        else:
            self.malicious_client = copy.copy(client)
            self.malicious_client.callbacks = ClientCallbackList(list(client.callbacks))
            self.malicious_client.callbacks.setup(self.malicious_client)

    @property
    def client_id(self):
        return self.benign_client.client_id

    @property
    def config(self):
        return self.benign_client.config

    @property
    def local_training(self) -> bool:
        return bool(self._local_training)

    @property
    def pseudo_grad_vec(self):
        return self._active().pseudo_grad_vec

    def _active(self):
        return self.malicious_client if self._is_under_attack else self.benign_client

    def attack(self):
        self._is_under_attack = True

    def restore(self):
        self._is_under_attack = False

    def setup(self):
        self.benign_client.setup()

    def set_callbacks(self, callbacks):
        self._active().add_callback(callbacks)

    def add_callback(self, callback):
        self._active().add_callback(callback)

    def active_callbacks(self) -> ClientCallbackList:
        return self._active().callbacks

    def train_one_round(self, data_reader: Iterable):
        if not self._local_training:
            return {"id": self.client_id}
        return self._active().train_one_round(data_reader=data_reader)

    def evaluate(self, test_loader):
        return self._active().evaluate(test_loader)
