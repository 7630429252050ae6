"""Grid-search expansion (replaces Ray Tune's variant generator).

A config value ``{"grid_search": [v1, v2, …]}`` at any depth (values may be
scalars, lists, or dicts that themselves contain nested grids — the reference
YAMLs use lists of dicts, e.g. ``fedavg_60_mnist.yaml:51-58``) expands to the
cross product, in the order the keys appear.
"""
import copy
import itertools
from typing import Any, Dict, Iterator, List, Tuple


def _is_grid(v) -> bool:
    return isinstance(v, dict) and set(v.keys()) == {"grid_search"}


def _find_grids(node, path=()) -> List[Tuple[tuple, list]]:
    out = []
    if _is_grid(node):
        out.append((path, node["grid_search"]))
    elif isinstance(node, dict):
        for k, v in node.items():
            out.extend(_find_grids(v, path + (k,)))
    elif isinstance(node, list):
        for i, v in enumerate(node):
            out.extend(_find_grids(v, path + (i,)))
    return out


def _set(node, path, value):
    for k in path[:-1]:
        node = node[k]
    node[path[-1]] = value


def expand_grid(config: Dict[str, Any]) -> Iterator[Tuple[Dict[str, Any], Dict[str, Any]]]:
    """Yields ``(resolved_config, {dotted.path: chosen value})`` for every variant."""
    grids = _find_grids(config)
    if not grids:
        yield copy.deepcopy(config), {}
        return
    # resolve the outermost grids first; a chosen value may contain further grids
    top = [g for g in grids if not any(g[0][: len(o[0])] == o[0] and g is not o for o in grids)]
    paths = [p for p, _ in top]
    for combo in itertools.product(*[vals for _, vals in top]):
        cfg = copy.deepcopy(config)
        chosen = {}
        for p, v in zip(paths, combo):
            if not p:
                cfg = copy.deepcopy(v)
            else:
                _set(cfg, p, copy.deepcopy(v))
            chosen["/".join(str(x) for x in p)] = v
        for sub, sub_chosen in expand_grid(cfg):
            yield sub, {**chosen, **sub_chosen}


def count_variants(config) -> int:
    return sum(1 for _ in expand_grid(config))


def _short(v) -> str:
    if isinstance(v, dict):
        t = v.get("type")
        rest = {k: x for k, x in v.items() if k != "type"}
        s = str(t).split(".")[-1] if t is not None else ""
        if rest:
            s += "_" + "_".join(f"{k}={_short(x)}" for k, x in rest.items())
        return s or "dict"
    if isinstance(v, float):
        return f"{v:g}"
    return str(v)


def variant_tag(chosen: Dict[str, Any], maxlen: int = 120) -> str:
    """Short, filesystem-safe description of a variant (used in trial dir names)."""
    parts = [f"{k.split('/')[-1]}={_short(v)}" for k, v in chosen.items()]
    tag = ",".join(parts)
    for ch in " /\\:*?\"<>|'{}[]":
        tag = tag.replace(ch, "_")
    return tag[:maxlen]
