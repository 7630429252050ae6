"""Worker-group sizing (reference ``fllib/core/execution/scaling_config.py``)."""
from dataclasses import dataclass


@dataclass
class WorkerGroupScalingConfig:
    num_workers: int = 0
    num_cpus_per_worker: float = 1
    num_gpus_per_worker: float = 0
    local_gpu_idx: int = 0
