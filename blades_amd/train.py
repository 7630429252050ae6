"""Command-line launcher:  ``python -m blades_amd.train file tuned_examples/X.yaml``.

Same entry points as ``blades/train.py`` (a Typer app with the ``file``
sub-command and a flag-only ``run`` callback).  The Ray cluster flags of the
reference (``--ray-num-nodes`` …) are accepted for command-line compatibility
and ignored; GPU placement is ``--max-concurrent`` / ``--gpus-per-trial``, and a
multi-GPU *single* trial is started with ``torch.distributed.run``::

    python -m torch.distributed.run --nproc-per-node 8 -m blades_amd.train file X.yaml
"""
import importlib.util
import json
import os
from typing import Optional

import typer
import yaml

from blades_amd.algorithms.registry import ALGORITHMS, register_all
from blades_amd.tune import run_experiments

register_all()

train_app = typer.Typer(add_completion=False, help="Train federated-learning experiments (blades_amd).")


def _parse_stop(stop: Optional[str]) -> dict:
    if not stop:
        return {}
    return json.loads(stop) if stop.strip().startswith("{") else yaml.safe_load(stop)


def load_experiments_from_file(config_file: str, file_type: Optional[str] = None, stop: Optional[str] = None,
                               checkpoint_config: Optional[dict] = None) -> dict:
    """YAML: ``{exp: {run, stop, config}}``.  Python: a module with a ``config`` variable
    (an AlgorithmConfig) and an optional ``stop`` dict (reference ``train.py:60-126``)."""
    file_type = file_type or ("python" if config_file.endswith(".py") else "yaml")
    if file_type == "yaml":
        with open(config_file) as f:
            experiments = yaml.safe_load(f)
        if stop:
            for exp in experiments.values():
                exp.setdefault("stop", {}).update(_parse_stop(stop))
    else:
        spec = importlib.util.spec_from_file_location("blades_user_config", config_file)
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        if not hasattr(module, "config"):
            raise ValueError("Python experiment files must define a `config` variable (an AlgorithmConfig).")
        config = module.config
        algo = config.algo_class
        run = algo if isinstance(algo, str) else f"{algo.__module__}.{algo.__name__}"
        for key, fn in ALGORITHMS.items():
            if fn()[0] is algo:
                run = key
        cfg = {k: v for k, v in config.to_dict().items() if k not in ("task_spec",)}
        experiments = {"default": {"run": run, "config": cfg,
                                   "stop": _parse_stop(stop) or getattr(module, "stop", {})}}
    for key, val in experiments.items():
        val["checkpoint_config"] = checkpoint_config or {}
        if "run" not in val:
            raise ValueError(f"experiment `{key}` has no `run` key")
    return experiments


def run_fllib_experiments(experiments: dict, v: bool = False, vv: bool = False, resume: bool = False,
                          max_concurrent: Optional[int] = None, local_dir: Optional[str] = None,
                          synthetic_data: bool = False, scheduler: Optional[str] = None,
                          scheduler_config: Optional[dict] = None, **ignored):
    if synthetic_data:
        os.environ["BLADES_SYNTHETIC_DATA"] = "1"
    for exp in experiments.values():
        if local_dir:
            exp["local_dir"] = local_dir
        if v or vv:
            exp.setdefault("config", {})["log_level"] = "DEBUG" if vv else "INFO"
    from blades_amd.parallel.runtime import init_runtime, shutdown_runtime

    init_runtime()
    sched = None
    if scheduler and scheduler.lower() != "fifo":
        from blades_amd.tune.schedulers import create_scheduler

        sched = {"type": scheduler, **(scheduler_config or {})}
        create_scheduler(scheduler, **(scheduler_config or {}))  # fail on a bad name / argument before any trial
    trials = run_experiments(experiments, resume=resume, verbose=2 if vv else 1, max_concurrent=max_concurrent,
                             scheduler=sched)
    shutdown_runtime()
    return trials


@train_app.command()
def file(
    config_file: str = typer.Argument(..., help="YAML (or .py) experiment file"),
    stop: Optional[str] = typer.Option(None, "--stop", help='stop criteria, e.g. \'{"training_iteration": 100}\''),
    checkpoint_freq: int = typer.Option(0, "--checkpoint-freq", help="checkpoint every N rounds (0 = never)"),
    checkpoint_at_end: bool = typer.Option(False, "--checkpoint-at-end"),
    keep_checkpoints_num: Optional[int] = typer.Option(None, "--keep-checkpoints-num"),
    checkpoint_score_attr: str = typer.Option("training_iteration", "--checkpoint-score-attr"),
    v: bool = typer.Option(False, "-v"),
    vv: bool = typer.Option(False, "-vv"),
    framework: Optional[str] = typer.Option(None, "--framework", help="ignored (always torch)"),
    trace: bool = typer.Option(False, "--trace", help="ignored"),
    local_mode: bool = typer.Option(False, "--local-mode", help="run all trials in this process"),
    ray_address: Optional[str] = typer.Option(None, "--ray-address", help="ignored (no Ray)"),
    ray_ui: bool = typer.Option(False, "--ray-ui", help="ignored"),
    ray_num_cpus: Optional[int] = typer.Option(None, "--ray-num-cpus", help="ignored"),
    ray_num_gpus: Optional[int] = typer.Option(None, "--ray-num-gpus", help="ignored"),
    ray_num_nodes: Optional[int] = typer.Option(None, "--ray-num-nodes", help="ignored"),
    ray_object_store_memory: Optional[int] = typer.Option(None, "--ray-object-store-memory", help="ignored"),
    resume: bool = typer.Option(False, "--resume", help="continue unfinished trials from their last checkpoint"),
    scheduler: Optional[str] = typer.Option("FIFO", "--scheduler", help="FIFO | median | asha (tune/schedulers.py)"),
    scheduler_config: str = typer.Option("{}", "--scheduler-config",
                                         help='scheduler arguments, e.g. \'{"metric": "acc_top_1", "max_t": 2000}\''),
    max_concurrent: Optional[int] = typer.Option(None, "--max-concurrent", help="trials in flight (default: #GPUs)"),
    local_dir: Optional[str] = typer.Option(None, "--local-dir", help="results root (default ~/blades_results)"),
    synthetic_data: bool = typer.Option(False, "--synthetic-data",
                                        help="use shape-identical synthetic data when dataset files are absent"),
):
    """Run the experiments of a YAML / Python file (grid search included)."""
    checkpoint_config = {
        "checkpoint_frequency": checkpoint_freq,
        "checkpoint_at_end": checkpoint_at_end,
        "num_to_keep": keep_checkpoints_num,
        "checkpoint_score_attribute": checkpoint_score_attr,
    }
    experiments = load_experiments_from_file(config_file, None, stop, checkpoint_config)
    run_fllib_experiments(experiments, v=v, vv=vv, resume=resume,
                          max_concurrent=1 if local_mode else max_concurrent, local_dir=local_dir,
                          synthetic_data=synthetic_data, scheduler=scheduler,
                          scheduler_config=yaml.safe_load(scheduler_config) or {})


@train_app.callback(invoke_without_command=True)
def run(
    ctx: typer.Context,
    algo: Optional[str] = typer.Option(None, "--algo", "--run", help="FEDAVG | FEDAVG_DP | dotted class path"),
    config: str = typer.Option("{}", "--config", help="algorithm config as JSON/YAML"),
    stop: Optional[str] = typer.Option('{"training_iteration": 10}', "--stop"),
    experiment_name: str = typer.Option("default", "--experiment-name"),
    checkpoint_freq: int = typer.Option(0, "--checkpoint-freq"),
    checkpoint_at_end: bool = typer.Option(False, "--checkpoint-at-end"),
    keep_checkpoints_num: Optional[int] = typer.Option(None, "--keep-checkpoints-num"),
    resume: bool = typer.Option(False, "--resume"),
    local_dir: Optional[str] = typer.Option(None, "--local-dir"),
    synthetic_data: bool = typer.Option(False, "--synthetic-data"),
    v: bool = typer.Option(False, "-v"),
    vv: bool = typer.Option(False, "-vv"),
):
    """Without a sub-command: run one algorithm from flags (``--algo FEDAVG --config '{...}'``)."""
    if ctx.invoked_subcommand is not None:
        return
    if not algo:
        typer.echo(ctx.get_help())
        raise typer.Exit(0)
    experiments = {experiment_name: {
        "run": algo, "config": yaml.safe_load(config) or {}, "stop": _parse_stop(stop),
        "checkpoint_config": {"checkpoint_frequency": checkpoint_freq, "checkpoint_at_end": checkpoint_at_end,
                              "num_to_keep": keep_checkpoints_num},
    }}
    run_fllib_experiments(experiments, v=v, vv=vv, resume=resume, local_dir=local_dir,
                          synthetic_data=synthetic_data)


def main():
    train_app()


if __name__ == "__main__":
    main()
