from blades_amd.parallel.worker import Worker, WorkerSpec  # noqa: F401
