from blades_amd.algorithms.registry import ALGORITHMS, _get_algorithm_class, register_all  # noqa: F401
