"""Algorithm base class — the trainable unit of one FL simulation.

Replaces ``ray.tune.Trainable`` as used by ``fllib/algorithms/algorithm.py``:
``train()`` = ``step()`` + bookkeeping (``training_iteration``, timing),
``step()`` = ``training_step()`` then ``evaluate()`` every
``evaluation_interval`` iterations, ``save_checkpoint`` writes
``algorithm_state.pkl`` with a top-level ``"config"`` key (the reference's whole
checkpoint) **plus** everything needed to really resume, and
``load_checkpoint`` restores it (the reference has none).
"""
import os
import pickle
import time
from typing import Callable, Dict, Optional

from blades_amd.types import ResultDict
from .algorithm_config import AlgorithmConfig
from .client_manager import ClientManager


class Algorithm:
    client_manager_cls = ClientManager

    def __init__(self, config: Optional[AlgorithmConfig] = None, logger_creator: Optional[Callable] = None,
                 **kwargs):
        config = config or self.get_default_config()
        default_config = self.get_default_config()
        if isinstance(config, dict):
            config = default_config.update_from_dict(config)
        elif not isinstance(config, type(default_config)):
            config = default_config.update_from_dict(config.to_dict())
        if config.algo_class is None:
            config.algo_class = type(self)
        config.validate()
        config.freeze()
        self.config = config
        self._iteration = 0
        self._time_total = 0.0
        self._logger_creator = logger_creator
        self._result_logger = logger_creator(config) if logger_creator else None
        self.setup(config)

    # -- Trainable surface -------------------------------------------------
    @property
    def iteration(self) -> int:
        return self._iteration

    @property
    def training_iteration(self) -> int:
        return self._iteration

    def setup(self, config: AlgorithmConfig):
        pass

    def train(self) -> ResultDict:
        t0 = time.perf_counter()
        result = self.step()
        dt = time.perf_counter() - t0
        self._iteration += 1
        self._time_total += dt
        result.setdefault("training_iteration", self._iteration)
        result["time_this_iter_s"] = dt
        result["time_total_s"] = self._time_total
        result["rounds_per_s"] = 1.0 / dt if dt > 0 else float("inf")
        if self._result_logger is not None:
            self._result_logger.on_result(result)
        from blades_amd.parallel.runtime import heartbeat

        heartbeat(f"iteration {self._iteration} done")  # (after evaluation and result logging: still alive)
        return result

    def step(self) -> ResultDict:
        evaluate_this_iter = (self.config.evaluation_interval is not None
                              and self.config.evaluation_interval > 0
                              and (self.iteration + 1) % self.config.evaluation_interval == 0)
        results: ResultDict = self.training_step()
        if evaluate_this_iter:
            results.update(self.evaluate() or {})
        return results

    def training_step(self) -> ResultDict:
        return {}

    def evaluate(self) -> Optional[ResultDict]:
        pass

    def stop(self):
        self.cleanup()

    def cleanup(self):
        pass

    @classmethod
    def get_default_config(cls) -> AlgorithmConfig:
        return AlgorithmConfig()

    @classmethod
    def default_resource_request(cls, config) -> dict:
        """Resource bundles of one trial (reference returns a Tune
        ``PlacementGroupFactory``; here a plain dict the trial scheduler reads)."""
        cf = dict(cls.get_default_config().to_dict(), **dict(config))
        local_worker = {"CPU": cf["num_cpus_for_driver"], "GPU": 0 if cf.get("_fake_gpus") else
                        cf["num_gpus_for_driver"]}
        remote = [{"CPU": cf["num_cpus_per_worker"], "GPU": cf["num_gpus_per_worker"],
                   **cf.get("custom_resources_per_worker", {})} for _ in range(int(cf["num_remote_workers"]))]
        return {"bundles": [local_worker] + remote, "strategy": cf.get("placement_strategy", "PACK")}

    # -- checkpointing -----------------------------------------------------
    def __getstate__(self) -> Dict:
        return {"config": self.config, "iteration": self._iteration, "time_total": self._time_total}

    def __setstate__(self, state: Dict):
        self._iteration = state.get("iteration", 0)
        self._time_total = state.get("time_total", 0.0)

    def save_checkpoint(self, checkpoint_dir: str) -> str:
        os.makedirs(checkpoint_dir, exist_ok=True)
        state = self.__getstate__()
        with open(os.path.join(checkpoint_dir, "algorithm_state.pkl"), "wb") as f:
            pickle.dump(state, f)
        return checkpoint_dir

    def load_checkpoint(self, checkpoint_dir: str) -> None:
        path = checkpoint_dir
        if os.path.isdir(path):
            path = os.path.join(path, "algorithm_state.pkl")
        with open(path, "rb") as f:  # a checkpoint this framework wrote
            state = pickle.load(f)
        self.__setstate__(state)

    save = save_checkpoint
    restore = load_checkpoint
