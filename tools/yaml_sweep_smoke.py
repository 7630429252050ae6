"""Runs a sample of the trials of every shipped experiment YAML for a few rounds on synthetic data (one process, one GPU).

    python tools/yaml_sweep_smoke.py [--per-file 6] [--rounds 2] [files...]

Every grid of blades_amd/tuned_examples/*.yaml expands to tens or hundreds of trials (aggregator x adversary x number of
malicious clients ...); this takes `--per-file` of them, evenly spread over the grid order, trains each for `--rounds`
rounds (with one evaluation at the end) and prints a line per trial.  Exit status 1 if any trial raised or produced a
non-finite loss.  It is a regression net for the less-travelled configurations, not an accuracy check.
"""
import argparse
import glob
import math
import os
import sys
import tempfile
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BLADES_SYNTHETIC_DATA", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--per-file", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    import torch

    from blades_amd.parallel.runtime import init_runtime, shutdown_runtime
    from blades_amd.train import load_experiments_from_file
    from blades_amd.tune.experiment import make_trials, run_trial

    init_runtime(use_gpu=torch.cuda.is_available())
    files = a.files or sorted(glob.glob(os.path.join(ROOT, "blades_amd", "tuned_examples", "*.yaml")))
    out_root = tempfile.mkdtemp(prefix="blades_yaml_smoke_")
    bad = 0
    for f in files:
        exps = load_experiments_from_file(f, None, None, None)
        for name, spec in exps.items():
            spec = dict(spec)
            spec["stop"] = {"training_iteration": a.rounds}
            spec["local_dir"] = out_root
            cfg = spec.setdefault("config", {})
            cfg["evaluation_interval"] = a.rounds
            trials = make_trials(name, spec)
            n = len(trials)
            step = max(1, n // max(a.per_file, 1))
            picked = trials[::step][: a.per_file]
            if trials[-1] not in picked:
                picked.append(trials[-1])
            print(f"== {os.path.basename(f)} / {name}: {n} trials, running {len(picked)}", flush=True)
            for t in picked:
                t0 = time.perf_counter()
                try:
                    res = run_trial(t, resume=False, verbose=0)
                    loss = res.get("train_loss")
                    ok = loss is None or math.isfinite(float(loss))
                    print(f"   {'ok ' if ok else 'BAD'} {t['dir_name'][:110]:110s} rounds {res.get('training_iteration')} "
                          f"loss {loss if loss is None else round(float(loss), 4)} acc {res.get('acc_top_1')} "
                          f"{time.perf_counter() - t0:.1f}s", flush=True)
                    bad += 0 if ok else 1
                except Exception:  # noqa: BLE001 — report and continue with the next trial
                    bad += 1
                    print(f"   EXC {t['dir_name'][:110]}\n" + traceback.format_exc(), flush=True)
    shutdown_runtime()
    print(f"failures: {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
