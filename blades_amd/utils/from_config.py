"""Object construction from ``{"type": ..., **kwargs}`` specs.

The reference delegates this to ``ray.rllib.utils.from_config`` (used at
``fllib/algorithms/server.py:64-66``, ``fllib/clients/client.py:19``,
``blades/adversaries/adversary.py:56-60``).  Ray is not a dependency here, so
this module implements the part of that contract the YAML files rely on:

* ``type`` may be a class, a dotted path, or a bare class name that is looked up
  first in the module of the default class and then in its parent package;
* dotted paths that start with the reference's package names (``blades.`` /
  ``fllib.``) resolve to this package, so unmodified reference YAMLs work;
* remaining dict keys become constructor kwargs, merged with call-site kwargs.
"""
import copy
import importlib
import logging
from typing import Any, Optional

logger = logging.getLogger(__name__)

# Reference module path prefix -> where that code lives in this package.
_ALIASES = (
    ("blades.adversaries", "blades_amd.adversaries"),
    ("blades.clients", "blades_amd.clients"),
    ("blades.algorithms", "blades_amd.algorithms"),
    ("fllib.aggregators", "blades_amd.aggregators"),
    ("fllib.algorithms", "blades_amd.algorithms"),
    ("fllib.clients", "blades_amd.clients"),
    ("fllib.tasks", "blades_amd.tasks"),
    ("fllib.datasets", "blades_amd.datasets"),
    ("fllib.models", "blades_amd.models"),
    ("fllib.core.execution", "blades_amd.parallel"),
    ("fllib.communication", "blades_amd.parallel"),
)

_warned = set()


def log_once(key: str) -> bool:
    """True the first time ``key`` is seen (stand-in for ``ray.util.log_once``)."""
    if key in _warned:
        return False
    _warned.add(key)
    return True


def force_list(x) -> list:
    if x is None:
        return []
    if isinstance(x, (list, tuple)):
        return list(x)
    return [x]


def deep_update(original: dict, new: dict, new_keys_allowed: bool = True) -> dict:
    """Recursively merges ``new`` into ``original`` (in place) and returns it."""
    for k, v in new.items():
        if k not in original and not new_keys_allowed:
            raise KeyError(f"Unknown config key `{k}`")
        if isinstance(original.get(k), dict) and isinstance(v, dict):
            deep_update(original[k], v, new_keys_allowed)
        else:
            original[k] = v
    return original


def _translate(path: str) -> str:
    for old, new in _ALIASES:
        if path == old or path.startswith(old + "."):
            return new + path[len(old):]
    return path


def resolve_class(spec: str, default_cls: Optional[type] = None):
    """Resolves a class given a dotted path or a bare class name."""
    if "." in spec:
        module_name, _, attr = _translate(spec).rpartition(".")
        last_err = None
        for cand in (module_name, spec.rpartition(".")[0]):
            try:
                module = importlib.import_module(cand)
                if hasattr(module, attr):
                    return getattr(module, attr)
            except ImportError as e:  # try the untranslated path next
                last_err = e
        raise ValueError(f"Cannot resolve `{spec}`: {last_err}")
    # Bare name: default class' module, then its parent package(s).
    search = []
    if default_cls is not None:
        mod = default_cls.__module__
        while mod:
            search.append(mod)
            mod = mod.rpartition(".")[0]
    for mod in search:
        try:
            module = importlib.import_module(mod)
        except ImportError:
            continue
        if hasattr(module, spec):
            return getattr(module, spec)
    raise ValueError(
        f"Cannot resolve bare class name `{spec}` relative to {default_cls}."
    )


def from_config(cls: Any = None, config: Any = None, **kwargs):
    """Instantiates an object from a class / dotted path / ``{"type":…}`` dict.

    Call patterns supported (all appear in the reference's call sites)::

        from_config(Mean, {"type": "Median"})
        from_config(Mean, "Trimmedmean")            # bare name
        from_config("blades.clients.Client", None, client_config=cfg)
        from_config(ClientCallback)                 # a class -> instance
        from_config({"type": "pkg.Cls", "a": 1})    # dict as first argument
        from_config(cls, {"adversary_class": ..}, clients=[..])
    """
    if isinstance(cls, dict) and config is None:
        cls, config = None, cls
    default_cls = cls if isinstance(cls, type) else None
    if isinstance(cls, str):
        cls = resolve_class(cls)

    ctor_kwargs = dict(kwargs)
    type_ = cls
    if isinstance(config, dict):
        config = copy.deepcopy(config)
        t = config.pop("type", None)
        if t is not None:
            type_ = t
        ctor_kwargs = {**config, **kwargs}
    elif isinstance(config, (str, type)):
        type_ = config
    elif config is not None and not isinstance(config, dict):
        # An already-built object: pass through (RLlib semantics).
        return config

    if isinstance(type_, str):
        type_ = resolve_class(type_, default_cls)
    if type_ is None:
        raise ValueError("from_config: no class given")
    if not callable(type_):
        return type_
    try:
        return type_(**ctor_kwargs)
    except TypeError as e:
        raise TypeError(f"Failed to construct {type_} with {list(ctor_kwargs)}: {e}") from e
