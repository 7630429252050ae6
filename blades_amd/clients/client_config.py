"""Client configuration (fields/defaults of ``fllib/clients/client_config.py:22-233``)."""
import copy
import logging
from typing import Callable, Dict

from blades_amd.types import NotProvided, PartialAlgorithmConfigDict
from blades_amd.utils.from_config import from_config, log_once
from .callbacks import ClientCallback

logger = logging.getLogger(__name__)

_TRAINING_KEYS = ("num_batch_per_round", "lr", "momentum")


class ClientConfig:
    def __init__(self, class_specifier) -> None:
        self.uid = None
        self.cls = class_specifier
        self.num_batch_per_round = 1
        self.lr = 0.1
        self.momentum = 0.0
        self.callbacks_config = ClientCallback
        # Declared by the reference but not consumed anywhere; kept so config
        # dumps (params.json) have the same keys.
        self.dataset_identifier = "mnist"
        self.train_batch_size = 32
        self.eva_batch_size = 64
        self.model_identifier = "mlp"
        self.model_config = None
        self.loss_identifier = "cross_entropy"
        self.loss_kargs = {}
        self.opt_identifier = "sgd"
        self.opt_kargs = {"lr": 0.1}
        self.lr_scheduler = None
        self.lr_scheduler_config = None

    def training(self, num_batch_per_round=NotProvided, lr=NotProvided, momentum=NotProvided) -> "ClientConfig":
        if num_batch_per_round is not NotProvided:
            self.num_batch_per_round = num_batch_per_round
        if lr is not NotProvided:
            self.lr = lr
        if momentum is not NotProvided:
            self.momentum = momentum
        return self

    def client_id(self, uid: str) -> "ClientConfig":
        self.uid = uid
        return self

    def callbacks(self, callbacks_config) -> "ClientConfig":
        if callbacks_config is None:
            callbacks_config = ClientCallback
        if not callable(callbacks_config) and not isinstance(callbacks_config, dict):
            raise ValueError(f"`callbacks_config` must be a class or a type-dict, got {callbacks_config}!")
        self.callbacks_config = callbacks_config
        return self

    def dataset(self, dataset_identifier=NotProvided, *, train_batch_size=NotProvided, eva_batch_size=NotProvided):
        if dataset_identifier is not NotProvided:
            self.dataset_identifier = dataset_identifier
        if train_batch_size is not NotProvided:
            self.train_batch_size = train_batch_size
        if eva_batch_size is not NotProvided:
            self.eva_batch_size = eva_batch_size
        return self

    def model(self, model_identifier=NotProvided, *, model_config=NotProvided, loss=NotProvided):
        if model_identifier is not NotProvided:
            self.model_identifier = model_identifier
        if model_config is not NotProvided:
            self.model_config = model_config
        if loss is not NotProvided:
            self.loss = loss
        return self

    def optimizer(self, opt_identifier=NotProvided, *, opt_kargs=NotProvided, lr_scheduler=NotProvided,
                  lr_scheduler_config=NotProvided):
        if opt_identifier is not NotProvided:
            self.opt_identifier = opt_identifier
        if opt_kargs is not NotProvided:
            self.opt_kargs = opt_kargs
        if lr_scheduler is not NotProvided:
            self.lr_scheduler = lr_scheduler
        if lr_scheduler_config is not NotProvided:
            self.lr_scheduler_config = lr_scheduler_config
        return self

    def build(self, use_copy: bool = True) -> Callable:
        if self.cls is None:
            raise ValueError("client_cls is not set")
        return from_config(self.cls, None, client_config=copy.deepcopy(self) if use_copy else self)

    def to_dict(self) -> dict:
        return copy.deepcopy(vars(self))

    def __getitem__(self, item):
        return getattr(self, item)

    def update_from_dict(self, config_dict: PartialAlgorithmConfigDict) -> "ClientConfig":
        for key, value in config_dict.items():
            if key in _TRAINING_KEYS:
                self.training(**{key: value})
            else:
                if not hasattr(self, key) and log_once("unknown_property_in_client_config_" + key):
                    logger.warning(f"{type(self).__name__}: unknown property `{key}` set from dict.")
                setattr(self, key, value)
        return self
