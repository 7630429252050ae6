from blades_amd.parallel.process_workers import ProcessWorkerGroup as WorkerGroup, WorkerMetadata  # noqa: F401
