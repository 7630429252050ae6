from blades_amd.parallel.scaling_config import WorkerGroupScalingConfig  # noqa: F401
