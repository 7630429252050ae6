"""Type aliases and the ``NotProvided`` sentinel (reference ``fllib/types.py:37-76``)."""
from typing import TYPE_CHECKING, Any, Dict, List, Union  # noqa: F401

import torch

TensorType = torch.Tensor
ModelWeights = Dict[str, Any]
ResultDict = Dict[str, Any]
AlgorithmConfigDict = Dict[str, Any]
PartialAlgorithmConfigDict = Dict[str, Any]
ClientID = str


class _NotProvided:
    """Singleton marking "argument not given" in the fluent config setters.

    ``None`` is a legal value for many settings, so the setters compare against
    this sentinel instead.  It survives pickling / deep-copy as a singleton.
    """

    _instance = None

    def __new__(cls):
        if cls._instance is None:
            cls._instance = super().__new__(cls)
        return cls._instance

    def __repr__(self):
        return "NotProvided"

    def __bool__(self):
        return False

    def __reduce__(self):
        return (_NotProvided, ())

    def __copy__(self):
        return self

    def __deepcopy__(self, memo):
        return self


NotProvided = _NotProvided()
