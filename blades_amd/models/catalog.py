"""Model catalog (reference ``fllib/models/catalog.py:14-47``).

String dispatch is by substring like the reference — ``"cct"``, ``"resnet"``
(= **ResNet-10**, trap noted in SURVEY), ``"mlp"``, ``"cnn"`` — with explicit
names (``resnet18``, ``resnet34`` …, every ``cct_*``/``cvt_*`` factory) checked
first so the north-star ResNet-18 config is expressible.
"""
from torch.nn import Module

from blades_amd.constants import FLLIB_MODEL
from blades_amd.utils.registry import _global_registry
from .cnn import FashionCNN
from .mlp import MLP
from . import resnet as _resnet

_EXPLICIT = {
    "resnet10": _resnet.ResNet10,
    "resnet18": _resnet.ResNet18,
    "resnet34": _resnet.ResNet34,
    "resnet50": _resnet.ResNet50,
    "resnet101": _resnet.ResNet101,
    "resnet152": _resnet.ResNet152,
}


class ModelCatalog:
    @staticmethod
    def get_model(model_config=None) -> Module:
        if isinstance(model_config, str):
            key = model_config.lower()
            if key in _EXPLICIT:
                return _EXPLICIT[key]()
            if "cct" in key or "cvt" in key:
                from . import cctnets

                if hasattr(cctnets, key):
                    return getattr(cctnets, key)()
                return cctnets.cct_2_3x2_32()
            if "resnet" in key:
                return _resnet.ResNet10()
            if "mlp" in key:
                return MLP()
            if "cnn" in key:
                return FashionCNN()
            if _global_registry.contains(FLLIB_MODEL, model_config):
                return _global_registry.get(FLLIB_MODEL, model_config)()
            raise NotImplementedError(f"Unknown model `{model_config}`")
        if isinstance(model_config, Module):
            return model_config
        if isinstance(model_config, type) and issubclass(model_config, Module):
            return model_config()
        if isinstance(model_config, dict) and model_config.get("custom_model"):
            cm = model_config["custom_model"]
            cls = cm if isinstance(cm, type) else _global_registry.get(FLLIB_MODEL, cm)
            return cls(**model_config.get("custom_model_config", {}))
        raise NotImplementedError(f"Cannot build a model from {model_config!r}")

    @staticmethod
    def register_custom_model(model_name: str, model_class: type) -> None:
        _global_registry.register(FLLIB_MODEL, model_name, model_class)
