from .grid import expand_grid, count_variants, variant_tag
from .loggers import TrialLogger
from .experiment import run_experiments, run_trial, make_trials, should_stop, TrialScheduler, latest_checkpoint

__all__ = ["expand_grid", "count_variants", "variant_tag", "TrialLogger", "run_experiments", "run_trial",
           "make_trials", "should_stop", "TrialScheduler", "latest_checkpoint"]
