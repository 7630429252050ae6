"""API-stability decorators (reference ``fllib/utils/annotations.py``).

They only tag the object (``_annotated``) and optionally warn; no behaviour.
"""
import functools
import warnings


def _tag(kind):
    def deco(obj=None, **kwargs):
        def wrap(o):
            try:
                o._annotated = kind
            except (AttributeError, TypeError):
                pass
            return o

        if obj is None or not callable(obj):
            return wrap
        return wrap(obj)

    return deco


PublicAPI = _tag("PublicAPI")
DeveloperAPI = _tag("DeveloperAPI")
ExperimentalAPI = _tag("ExperimentalAPI")
OverrideToImplementCustomLogic = _tag("OverrideToImplementCustomLogic")


def override(cls):
    """Asserts at decoration time that ``cls`` defines the overridden method."""

    def check(method):
        if method.__name__ not in dir(cls):
            raise NameError(f"{method} does not override any method of {cls}")
        return method

    return check


def is_overridden(obj) -> bool:
    """True iff a method tagged ``OverrideToImplementCustomLogic`` was replaced."""
    return getattr(obj, "_annotated", None) != "OverrideToImplementCustomLogic"


def Deprecated(old=None, *, new=None, help=None, error=False):
    def deco(obj):
        @functools.wraps(obj)
        def wrapped(*a, **k):
            msg = f"`{old or obj.__name__}` is deprecated."
            if new:
                msg += f" Use `{new}` instead."
            if help:
                msg += " " + help
            if error:
                raise DeprecationWarning(msg)
            warnings.warn(msg, DeprecationWarning, stacklevel=2)
            return obj(*a, **k)

        return wrapped

    return deco
