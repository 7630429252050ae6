"""Early-stopping trial schedulers selected with ``--scheduler`` / ``--scheduler-config``.

The reference hands these flags to Ray Tune's ``create_scheduler`` (``train.py:380-386``).  Here trials are
independent processes pinned to GPUs (``TrialScheduler``), possibly several at a time and possibly resumed later,
so the schedulers keep no state of their own: every decision is taken from the ``result.json`` files the trials of
the experiment have written so far.  That makes a decision reproducible, restart-safe and identical whether
trials run one after another or eight at a time.

* ``fifo``                      — never stops a trial early (default).
* ``median`` / ``medianstopping`` — Tune's MedianStoppingRule: stop a trial at time ``t`` when its best result so
  far is worse than the median of the other trials' running averages up to ``t``.
* ``asha`` / ``async_hyperband`` / ``asynchyperband`` — asynchronous successive halving: at every rung
  ``grace_period * reduction_factor**k`` a trial continues only if it is in the top ``1/reduction_factor`` of
  the trials that have reached that rung.
"""
import glob
import json
import os
from typing import Any, Dict, List, Optional


def _read_results(path: str) -> List[Dict[str, Any]]:
    rows = []
    try:
        with open(path) as f:
            for line in f:
                line = line.strip()
                if line:
                    try:
                        rows.append(json.loads(line))
                    except json.JSONDecodeError:  # a line another process is still writing
                        break
    except OSError:
        pass
    return rows


class FIFOScheduler:
    def __init__(self, **_ignored):
        pass

    def should_stop(self, experiment_dir: str, trial_dir_name: str, result: Dict[str, Any]) -> bool:
        return False


class _MetricScheduler(FIFOScheduler):
    def __init__(self, metric: str = "acc_top_1", mode: str = "max", time_attr: str = "training_iteration",
                 grace_period: int = 1, **_ignored):
        if mode not in ("max", "min"):
            raise ValueError("mode must be 'max' or 'min'")
        self.metric, self.sign, self.time_attr, self.grace = metric, (1.0 if mode == "max" else -1.0), time_attr, int(
            grace_period)

    def _series(self, rows):
        """[(t, signed metric)] of the rows that carry the metric (evaluation rounds)."""
        return [(r[self.time_attr], self.sign * float(r[self.metric])) for r in rows
                if self.metric in r and r.get(self.metric) is not None and self.time_attr in r]

    def _others(self, experiment_dir, own):
        out = []
        for path in sorted(glob.glob(os.path.join(experiment_dir, "*", "result.json"))):
            if os.path.basename(os.path.dirname(path)) != own:
                s = self._series(_read_results(path))
                if s:
                    out.append(s)
        return out


class MedianStoppingRule(_MetricScheduler):
    def __init__(self, min_samples_required: int = 3, **kw):
        super().__init__(**kw)
        self.min_samples = int(min_samples_required)

    def should_stop(self, experiment_dir, trial_dir_name, result):
        if self.metric not in result or result.get(self.time_attr, 0) < self.grace:
            return False
        t = result[self.time_attr]
        own = self._series(_read_results(os.path.join(experiment_dir, trial_dir_name, "result.json")))
        own = [v for tt, v in own if tt <= t] + [self.sign * float(result[self.metric])]
        means = []
        for s in self._others(experiment_dir, trial_dir_name):
            upto = [v for tt, v in s if tt <= t]
            if upto:
                means.append(sum(upto) / len(upto))
        if len(means) < self.min_samples:
            return False
        means.sort()
        n = len(means)
        median = means[n // 2] if n % 2 else 0.5 * (means[n // 2 - 1] + means[n // 2])
        return max(own) < median


class AsyncHyperBandScheduler(_MetricScheduler):
    def __init__(self, max_t: int = 100, reduction_factor: float = 4, **kw):
        super().__init__(**kw)
        self.max_t, self.rf = int(max_t), float(reduction_factor)
        if self.rf <= 1:
            raise ValueError("reduction_factor must be > 1")
        self.rungs = []
        r = max(self.grace, 1)
        while r < self.max_t:
            self.rungs.append(int(r))
            r *= self.rf

    @staticmethod
    def _at_rung(series, rung):
        """The first recorded value at or after ``rung`` (metrics only exist on evaluation rounds)."""
        for t, v in series:
            if t >= rung:
                return v
        return None

    def should_stop(self, experiment_dir, trial_dir_name, result):
        if self.metric not in result:
            return False
        t = result.get(self.time_attr, 0)
        if t >= self.max_t:
            return True
        own = self._series(_read_results(os.path.join(experiment_dir, trial_dir_name, "result.json")))
        own = [(tt, v) for tt, v in own if tt < t] + [(t, self.sign * float(result[self.metric]))]
        others = self._others(experiment_dir, trial_dir_name)
        for rung in self.rungs:
            if t < rung:
                break
            mine = self._at_rung(own, rung)
            if mine is None:
                continue
            recorded = [v for v in (self._at_rung(s, rung) for s in others) if v is not None] + [mine]
            keep = max(1, int(len(recorded) / self.rf))
            cutoff = sorted(recorded, reverse=True)[keep - 1]
            if len(recorded) >= self.rf and mine < cutoff:
                return True
        return False


_SCHEDULERS = {
    "fifo": FIFOScheduler, "median": MedianStoppingRule, "medianstopping": MedianStoppingRule,
    "medianstoppingrule": MedianStoppingRule, "asha": AsyncHyperBandScheduler,
    "async_hyperband": AsyncHyperBandScheduler, "asynchyperband": AsyncHyperBandScheduler,
}


def create_scheduler(scheduler: Optional[str] = "FIFO", **config):
    key = (scheduler or "fifo").lower()
    if key not in _SCHEDULERS:
        raise ValueError(f"unknown scheduler {scheduler!r}; available: {sorted(set(_SCHEDULERS))}")
    return _SCHEDULERS[key](**config)
