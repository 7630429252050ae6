from blades_amd.parallel.worker_group_config import WorkerGroupConfig  # noqa: F401
