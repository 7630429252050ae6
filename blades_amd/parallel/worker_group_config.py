"""WorkerGroupConfig (reference ``fllib/core/execution/worker_group_config.py``)."""
from typing import Optional, Type, Union

from blades_amd.types import NotProvided
from .scaling_config import WorkerGroupScalingConfig
from .worker import Worker, WorkerSpec
from .worker_group import WorkerGroup


class WorkerGroupConfig:
    def __init__(self, cls: Type[WorkerGroup] = None) -> None:
        self.worker_group_class = cls or WorkerGroup
        self.task_spec = None
        self.worker_class = Worker
        self.num_remote_workers = 0
        self.num_cpus_per_worker = 1
        self.num_gpus_per_worker = 0
        self.local_gpu_idx = 0
        self.max_remote_requests_in_flight_per_actor = 1

    def validate(self) -> None:
        if self.task_spec is None:
            raise ValueError("Cannot build a WorkerGroup without a task_spec; call .task(task_spec=...).")

    def build(self, device=None) -> WorkerGroup:
        self.validate()
        scaling = WorkerGroupScalingConfig(
            num_workers=self.num_remote_workers, num_cpus_per_worker=self.num_cpus_per_worker,
            num_gpus_per_worker=self.num_gpus_per_worker, local_gpu_idx=self.local_gpu_idx)
        spec = WorkerSpec(task_spec=self.task_spec, worker_scaling_config=scaling)
        return self.worker_group_class(spec, self.max_remote_requests_in_flight_per_actor, device=device)

    def task(self, *, task_spec=NotProvided) -> "WorkerGroupConfig":
        if task_spec is not NotProvided:
            self.task_spec = task_spec
        return self

    def worker(self, *, worker_class=NotProvided) -> "WorkerGroupConfig":
        if worker_class is not NotProvided:
            self.worker_class = worker_class
        return self

    def resources(self, num_remote_workers: Optional[int] = NotProvided,
                  num_gpus_per_worker: Optional[Union[float, int]] = NotProvided,
                  num_cpus_per_worker: Optional[Union[float, int]] = NotProvided,
                  local_gpu_idx: Optional[int] = NotProvided) -> "WorkerGroupConfig":
        if num_remote_workers is not NotProvided:
            self.num_remote_workers = num_remote_workers
        if num_gpus_per_worker is not NotProvided:
            self.num_gpus_per_worker = num_gpus_per_worker
        if num_cpus_per_worker is not NotProvided:
            self.num_cpus_per_worker = num_cpus_per_worker
        if local_gpu_idx is not NotProvided:
            self.local_gpu_idx = local_gpu_idx
        return self
