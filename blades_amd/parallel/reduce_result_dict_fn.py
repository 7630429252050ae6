"""Reduction of per-client / per-rank result dicts (reference ``fllib/core/execution/reduce_result_dict_fn.py:9-18``,
which maps ``np.mean`` over the leaves with dm-tree).  Nested dicts / lists / tuples are walked directly."""
from typing import Any, Callable, List

import numpy as np


def map_structure(fn: Callable, *structs: Any) -> Any:
    head = structs[0]
    if isinstance(head, dict):
        for s in structs[1:]:
            if not isinstance(s, dict) or s.keys() != head.keys():
                raise ValueError("result dicts do not share one structure")
        return type(head)((k, map_structure(fn, *(s[k] for s in structs))) for k in head)
    if isinstance(head, (list, tuple)):
        if any(not isinstance(s, (list, tuple)) or len(s) != len(head) for s in structs[1:]):
            raise ValueError("result sequences do not share one structure")
        return type(head)(map_structure(fn, *parts) for parts in zip(*structs))
    return fn(*structs)


def _reduce_mean_results(results: List[dict]) -> dict:
    """Average of every leaf over the list of result dicts."""
    if not results:
        return {}
    return map_structure(lambda *x: np.mean(x), *results)


def _reduce_weighted_mean_results(results: List[dict], weights) -> dict:
    """Same, weighted (e.g. by each client's number of test samples, as ``Fedavg.evaluate`` does)."""
    w = np.asarray(list(weights), dtype=np.float64)
    return map_structure(lambda *x: float(np.average(np.asarray(x, dtype=np.float64), weights=w)), *results)
