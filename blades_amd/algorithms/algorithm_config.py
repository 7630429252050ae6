"""AlgorithmConfig: fluent builder + dict shims.

Defaults and key names follow ``fllib/algorithms/algorithm_config.py:54-96``;
YAML/dict configs go through ``update_from_dict`` exactly like the reference's
Tune path.  Engine-specific additions (all optional): ``execution_mode``
(auto | batched | sequential), ``clients_per_chunk``, ``amp_dtype``,
``synthetic_data``.  The Ray resource keys (``num_remote_workers``,
``num_gpus_per_worker`` …) are accepted and recorded but only
``num_gpus_for_driver``/``num_gpus_per_worker`` > 0 influence anything: they
mean "use the GPU".
"""
import copy
import logging
from typing import Callable, Optional

from blades_amd.clients import ClientConfig
from blades_amd.parallel.worker_group_config import WorkerGroupConfig
from blades_amd.tasks import TaskSpec
from blades_amd.types import NotProvided
from blades_amd.utils.from_config import log_once
from blades_amd.utils.registry import get_trainable_cls
from .server_config import ServerConfig

logger = logging.getLogger(__name__)

TRIAL_INFO = "__trial_info__"


class AlgorithmConfig:
    def __init__(self, algo_class=None):
        self.seed = 1234
        self.algo_class = algo_class
        self.global_model = None
        self.num_batch_per_round = 1
        self.local_lr = NotProvided
        self.client_config = {}
        self.server_config = {}
        self.logger_creator = None
        self.learner_class = None
        self.task_spec = TaskSpec(task_class="blades_amd.tasks.MNIST", alg_config=self)
        self._is_frozen = False

        # resources (Ray-era keys; see module docstring)
        self.num_cpus_for_driver = 1
        self.num_gpus_for_driver = 0
        self.num_cpus_per_worker = 1
        self.num_gpus_per_worker = 0
        self.num_remote_workers = 1
        self.num_gpus = 0
        self._fake_gpus = False
        self.custom_resources_per_worker = {}
        self.placement_strategy = "PACK"

        # data
        self.dataset = "cifar10"
        self.dataset_config = {}
        self.num_clients = 0
        self.reuse_actors = True

        # evaluation
        self.evaluation_interval = 10
        self.evaluation_config = None

        # MI355X engine knobs
        self.execution_mode = "auto"
        self.clients_per_chunk = "auto"  # int, or "auto" (engine/trainer.py picks per device and batch size)
        # how the update matrix reaches the aggregator on several ranks: "alltoall" (row shard -> column shard,
        # aggregation sharded over D) or "allgather" (every rank receives the full [K, D] matrix and aggregates
        # replicated — BASELINE config #5; 8x the bytes of the all-to-all on 8 ranks)
        self.update_layout = "alltoall"
        self.amp_dtype = "auto"
        self.device = "auto"
        self.nan_guard = True

    # -- fluent setters -------------------------------------------------
    def client(self, *, client_config: Optional[dict] = NotProvided):
        if client_config is not NotProvided:
            self.client_config = client_config
        return self

    def data(self, *, dataset=NotProvided, num_clients=NotProvided, dataset_config=NotProvided):
        if dataset is not NotProvided:
            self.dataset = dataset
        if num_clients is not NotProvided:
            self.num_clients = num_clients
        if dataset_config is not NotProvided:
            self.dataset_config = dataset_config
        return self

    def training(self, *, global_model=NotProvided, local_lr=NotProvided, num_batch_per_round=NotProvided,
                 server_config=NotProvided):
        if local_lr is not NotProvided:
            self.local_lr = local_lr
        if num_batch_per_round is not NotProvided:
            self.num_batch_per_round = num_batch_per_round
        if global_model is not NotProvided:
            self.global_model = global_model
        if server_config is not NotProvided:
            self.server_config = server_config
        return self

    def resources(self, *, num_cpus_per_worker=NotProvided, num_gpus_per_worker=NotProvided,
                  num_cpus_for_driver=NotProvided, num_gpus_for_driver=NotProvided,
                  num_remote_workers=NotProvided):
        for k, v in (("num_cpus_per_worker", num_cpus_per_worker), ("num_gpus_per_worker", num_gpus_per_worker),
                     ("num_cpus_for_driver", num_cpus_for_driver), ("num_gpus_for_driver", num_gpus_for_driver),
                     ("num_remote_workers", num_remote_workers)):
            if v is not NotProvided:
                setattr(self, k, v)
        return self

    def evaluation(self, *, evaluation_interval=NotProvided, evaluation_config=NotProvided):
        if evaluation_interval is not NotProvided:
            self.evaluation_interval = evaluation_interval
        if evaluation_config is not NotProvided:
            self.evaluation_config = evaluation_config
        return self

    def engine(self, *, execution_mode=NotProvided, clients_per_chunk=NotProvided, amp_dtype=NotProvided,
               device=NotProvided, update_layout=NotProvided):
        if update_layout is not NotProvided and update_layout not in ("alltoall", "allgather"):
            raise ValueError(f"update_layout must be 'alltoall' or 'allgather', got {update_layout!r}")
        for k, v in (("execution_mode", execution_mode), ("clients_per_chunk", clients_per_chunk),
                     ("amp_dtype", amp_dtype), ("device", device), ("update_layout", update_layout)):
            if v is not NotProvided:
                setattr(self, k, v)
        return self

    # -- derived configs -------------------------------------------------
    def _require_frozen(self, what):
        if not self._is_frozen:
            raise ValueError(f"Cannot call `{what}()` on an unfrozen AlgorithmConfig! Please call `freeze()` first.")

    def get_client_config(self) -> ClientConfig:
        self._require_frozen("get_client_config")
        cfg = ClientConfig(class_specifier="blades_amd.clients.BaseClient").training(
            num_batch_per_round=self.num_batch_per_round, lr=self.local_lr)
        return cfg.update_from_dict(self.client_config)

    def get_server_config(self) -> ServerConfig:
        self._require_frozen("get_server_config")
        return ServerConfig(class_specifier="blades_amd.algorithms.Server",
                            task_spec=self.task_spec).update_from_dict(self.server_config)

    def get_worker_group_config(self) -> WorkerGroupConfig:
        self._require_frozen("get_worker_group_config")
        return WorkerGroupConfig().resources(
            num_remote_workers=self.num_remote_workers, num_cpus_per_worker=self.num_cpus_per_worker,
            num_gpus_per_worker=self.num_gpus_per_worker).task(task_spec=self.task_spec)

    def build(self, logger_creator: Optional[Callable] = None, use_copy: bool = True):
        if logger_creator is not None:
            self.logger_creator = logger_creator
        algo_class = self.algo_class
        if isinstance(algo_class, str):
            algo_class = get_trainable_cls(algo_class)
        return algo_class(config=self if not use_copy else copy.deepcopy(self), logger_creator=self.logger_creator)

    # -- validation / freezing ---------------------------------------------
    def validate(self) -> None:
        if self.dataset_config.get("num_clients", None) is None:
            if self.num_clients < 1:
                raise ValueError("You must specify `num_clients` in your `dataset_config` or as `num_clients`!")
            self.dataset_config["num_clients"] = self.num_clients
        else:
            if self.num_clients > 0 and self.num_clients != self.dataset_config["num_clients"]:
                raise ValueError("You cannot specify both `num_clients` and `dataset_config['num_clients']`.")
            self.num_clients = self.dataset_config["num_clients"]
        if self.dataset_config.get("seed", None) is None:
            self.dataset_config["seed"] = self.seed

    def freeze(self) -> None:
        self._is_frozen = True

    # -- dict shims ---------------------------------------------------------
    def __deepcopy__(self, memo):
        new = type(self).__new__(type(self))
        memo[id(self)] = new
        for k, v in vars(self).items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        if isinstance(new.task_spec, TaskSpec) and new.task_spec.config is self:
            new.task_spec.config = new
        return new

    @staticmethod
    def _translate_special_keys(key: str) -> str:
        return {"callbacks": "callbacks_config", "input": "input_", "lambda": "lambda_"}.get(key, key)

    def __getitem__(self, item):
        return getattr(self, self._translate_special_keys(item))

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return hasattr(self, self._translate_special_keys(key))

    def get(self, key, default=None):
        return getattr(self, self._translate_special_keys(key), default)

    def pop(self, key, default=None):
        return self.get(key, default)

    def to_dict(self) -> dict:
        skip = ("algo_class", "_is_frozen", "task_spec", "logger_creator")
        out = {}
        for k, v in vars(self).items():
            if k in skip:
                continue
            out[k] = None if v is NotProvided else copy.deepcopy(v)
        out["task_spec"] = getattr(self.task_spec, "task_class", None)
        return out

    def keys(self):
        return self.to_dict().keys()

    def values(self):
        return self.to_dict().values()

    def items(self):
        return self.to_dict().items()

    @classmethod
    def from_dict(cls, config_dict: dict) -> "AlgorithmConfig":
        return cls().update_from_dict(config_dict)

    def update_from_dict(self, config_dict) -> "AlgorithmConfig":
        eval_call = {}
        for key, value in dict(config_dict).items():
            key = self._translate_special_keys(key)
            if key == TRIAL_INFO:
                continue
            if key.startswith("evaluation_"):
                eval_call[key] = value
            elif key == "task_spec" and not isinstance(value, TaskSpec):
                if value is not None:
                    self.task_spec = TaskSpec(task_class=value, alg_config=self)
            else:
                if not hasattr(self, key) and log_once("unknown_property_in_algo_config_" + key):
                    logger.warning(f"{type(self).__name__}: property `{key}` is not a known setting; set anyway.")
                setattr(self, key, copy.deepcopy(value) if isinstance(value, (dict, list)) else value)
        self.evaluation(**eval_call)
        return self
