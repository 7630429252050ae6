"""Process-global name -> object registry.

Replaces ``ray.tune.registry._global_registry`` which the reference uses for
custom datasets/models (``fllib/datasets/catalog.py:86-100``,
``fllib/models/catalog.py:34-47``) and trainables (``blades/train.py:21-31``).
"""
from typing import Any, Dict, Tuple

TRAINABLE = "trainable"


class _Registry:
    def __init__(self):
        self._store: Dict[Tuple[str, str], Any] = {}

    def register(self, category: str, key: str, value: Any) -> None:
        self._store[(category, key)] = value

    def contains(self, category: str, key: str) -> bool:
        return (category, key) in self._store

    def get(self, category: str, key: str) -> Any:
        if (category, key) not in self._store:
            raise KeyError(f"`{key}` is not registered under `{category}`")
        return self._store[(category, key)]

    def keys(self, category: str):
        return [k for (c, k) in self._store if c == category]


_global_registry = _Registry()


def register_trainable(name: str, trainable) -> None:
    _global_registry.register(TRAINABLE, name, trainable)


def get_trainable_cls(name: str):
    return _global_registry.get(TRAINABLE, name)
