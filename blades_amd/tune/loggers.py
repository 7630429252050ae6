"""Trial result files, same layout as Tune's default loggers:

``<trial_dir>/params.json``, ``params.pkl``, ``result.json`` (one JSON object per
line), ``progress.csv``.  ``visualize.py`` reads ``params.pkl`` + the last line
of ``result.json`` (reference ``tuned_examples/visualization/visualize.py:8-49``).
"""
import csv
import json
import os
import pickle
import time
from typing import Any, Dict

import numpy as np
import torch


def _jsonable(v):
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, torch.Tensor):
        return v.item() if v.numel() == 1 else v.tolist()
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (str, int, float, bool)) or v is None:
        return v
    return str(v)


def flatten(d: Dict[str, Any], prefix="") -> Dict[str, Any]:
    out = {}
    for k, v in d.items():
        key = f"{prefix}{k}"
        if isinstance(v, dict):
            out.update(flatten(v, key + "/"))
        else:
            out[key] = v
    return out


class TrialLogger:
    def __init__(self, trial_dir: str, params: Dict[str, Any], trial_id: str = "", experiment: str = ""):
        self.dir = trial_dir
        os.makedirs(trial_dir, exist_ok=True)
        self.trial_id, self.experiment = trial_id, experiment
        with open(os.path.join(trial_dir, "params.json"), "w") as f:
            json.dump(_jsonable(params), f, indent=2, sort_keys=True)
        with open(os.path.join(trial_dir, "params.pkl"), "wb") as f:
            pickle.dump(params, f)
        self._json = open(os.path.join(trial_dir, "result.json"), "a")
        self._csv_path = os.path.join(trial_dir, "progress.csv")
        self._csv_fields = None
        self._t0 = time.time()

    def on_result(self, result: Dict[str, Any]):
        row = dict(_jsonable(result))
        row.setdefault("trial_id", self.trial_id)
        row.setdefault("experiment_tag", self.experiment)
        row.setdefault("timestamp", int(time.time()))
        row.setdefault("date", time.strftime("%Y-%m-%d_%H-%M-%S"))
        self._json.write(json.dumps(row) + "\n")
        self._json.flush()
        flat = flatten(row)
        if self._csv_fields is None:
            self._csv_fields = list(flat.keys())
            new = not os.path.exists(self._csv_path) or os.path.getsize(self._csv_path) == 0
            self._csv = open(self._csv_path, "a", newline="")
            self._writer = csv.DictWriter(self._csv, fieldnames=self._csv_fields, extrasaction="ignore")
            if new:
                self._writer.writeheader()
        self._writer.writerow({k: flat.get(k, "") for k in self._csv_fields})
        self._csv.flush()

    def close(self):
        self._json.close()
        if self._csv_fields is not None:
            self._csv.close()
