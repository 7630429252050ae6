from blades_amd.parallel.worker_group import WorkerGroup  # noqa: F401
