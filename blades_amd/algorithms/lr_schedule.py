"""Piecewise-linear learning-rate schedule.

Stand-in for RLlib's ``LearningRateSchedule``/``PiecewiseSchedule`` that the
reference server mixes in (``fllib/algorithms/server.py:22,49``):
``lr_schedule = [[t0, v0], [t1, v1], …]``; linear interpolation inside
``[t_i, t_{i+1})``; outside every interval the value of the **last** endpoint.
Without a schedule the lr is constant.
"""
from typing import List, Optional, Sequence


class PiecewiseSchedule:
    def __init__(self, endpoints: Sequence[Sequence[float]], outside_value: Optional[float] = None):
        pts = [(float(t), float(v)) for t, v in endpoints]
        assert pts == sorted(pts, key=lambda p: p[0]), "lr_schedule timesteps must be sorted"
        self.endpoints = pts
        self.outside_value = pts[-1][1] if outside_value is None else outside_value

    def value(self, t: float) -> float:
        for (lt, lv), (rt, rv) in zip(self.endpoints[:-1], self.endpoints[1:]):
            if lt <= t < rt:
                alpha = (t - lt) / (rt - lt)
                return lv + alpha * (rv - lv)
        return self.outside_value

    __call__ = value


class LearningRateSchedule:
    def __init__(self, lr: float, lr_schedule: Optional[List[List[float]]]):
        self._lr_schedule = None
        if lr_schedule is None:
            self.cur_lr = float(lr)
        else:
            self._lr_schedule = PiecewiseSchedule(lr_schedule)
            self.cur_lr = self._lr_schedule.value(0)

    def on_global_var_update(self, global_vars: dict) -> None:
        if self._lr_schedule is not None and global_vars is not None:
            self.cur_lr = self._lr_schedule.value(global_vars["timestep"])
