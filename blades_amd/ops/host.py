"""ctypes bindings of ``libblades_host.so`` (built on demand with g++)."""
import ctypes

import numpy as np

from . import build as _build

_lib = None

LINKAGES = {"single": 0, "average": 1, "complete": 2}


def lib():
    global _lib
    if _lib is None:
        path = _build.build_host()
        _lib = ctypes.CDLL(str(path))
        _lib.blades_agglomerative2.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib.blades_kmeans.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.blades_dirichlet_partition.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_double, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
        _lib.blades_lpt_assign.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return _lib


def agglomerative2(dist: np.ndarray, linkage: str = "average") -> np.ndarray:
    """Labels (0/1) of the two-cluster cut for a precomputed distance matrix."""
    d = np.ascontiguousarray(dist, dtype=np.float64)
    n = d.shape[0]
    assert d.shape == (n, n)
    labels = np.zeros(n, dtype=np.int32)
    rc = lib().blades_agglomerative2(d.ctypes.data, n, LINKAGES[linkage], labels.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"agglomerative2 failed rc={rc}")
    return labels


def kmeans(x: np.ndarray, k: int = 2, seed: int = 0, n_init: int = 10, max_iter: int = 300):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n, d = x.shape
    labels = np.zeros(n, dtype=np.int32)
    centers = np.zeros((k, d), dtype=np.float64)
    rc = lib().blades_kmeans(x.ctypes.data, n, d, k, seed, n_init, max_iter, labels.ctypes.data, centers.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"kmeans failed rc={rc}")
    return labels, centers


def dirichlet_partition(y: np.ndarray, num_classes: int, num_clients: int, alpha: float, seed: int,
                        min_size: int = 10) -> np.ndarray:
    y = np.ascontiguousarray(y, dtype=np.int64)
    out = np.zeros(y.shape[0], dtype=np.int32)
    rc = lib().blades_dirichlet_partition(y.ctypes.data, y.shape[0], num_classes, num_clients, float(alpha),
                                          int(seed), int(min_size), out.ctypes.data)
    if rc < 0:
        raise RuntimeError(f"dirichlet_partition failed rc={rc}")
    return out


def lpt_assign(cost: np.ndarray, world: int) -> np.ndarray:
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    out = np.zeros(cost.shape[0], dtype=np.int32)
    rc = lib().blades_lpt_assign(cost.ctypes.data, cost.shape[0], world, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("lpt_assign failed")
    return out
