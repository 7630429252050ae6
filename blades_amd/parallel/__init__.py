from .context import ShardContext, local_context, context_from_env, padded_width
from .runtime import Runtime, init_runtime, get_runtime, set_runtime, shutdown_runtime
from .scaling_config import WorkerGroupScalingConfig
from .worker import Worker, WorkerSpec
from .worker_group import WorkerGroup
from .worker_group_config import WorkerGroupConfig
from .session import get_session

__all__ = ["ShardContext", "local_context", "context_from_env", "padded_width", "Runtime", "init_runtime",
           "get_runtime", "set_runtime", "shutdown_runtime", "WorkerGroupScalingConfig", "Worker", "WorkerSpec",
           "WorkerGroup", "WorkerGroupConfig", "get_session"]
