"""CLI -> experiment runner -> result files -> checkpoint/resume."""
import json
import os
import pickle
import subprocess
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tiny_yaml(path, data_root, iters=4, grid=True):
    exp = {"tiny": {"run": "FEDAVG", "stop": {"training_iteration": iters, "train_loss": 100000}, "config": {
        "seed": {"grid_search": [1, 2]} if grid else 1,
        "dataset_config": {"type": "mnist", "num_clients": 6, "num_classes": 10, "train_batch_size": 8,
                           "n_train": 200, "n_test": 60, "data_root": data_root, "use_cache": False},
        "evaluation_interval": 2, "global_model": "mlp", "num_remote_workers": 0, "num_gpus_per_worker": 0,
        "device": "cpu",
        "client_config": {"lr": 0.5},
        "server_config": {"aggregator": {"grid_search": [{"type": "Median"}, {"type": "Trimmedmean"}]} if grid else
                          {"type": "Centeredclipping"},
                          "optimizer": {"type": "SGD", "lr": 0.5, "lr_schedule": [[0, 0.5], [2, 0.5], [3, 0.1]],
                                        "momentum": 0.9}},
        "num_malicious_clients": 1,
        "adversary_config": {"type": "blades.adversaries.ALIEAdversary"}}}}
    with open(path, "w") as f:
        yaml.safe_dump(exp, f)


def cli(args, env_extra=None):
    env = dict(os.environ, BLADES_SYNTHETIC_DATA="1", PYTHONPATH=ROOT)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-m", "blades_amd.train"] + args, env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r


def test_grid_file_run_writes_tune_layout(tmp_path):
    y = tmp_path / "exp.yaml"
    tiny_yaml(str(y), str(tmp_path / "data"))
    out = tmp_path / "results"
    cli(["file", str(y), "--local-dir", str(out), "--local-mode", "--checkpoint-at-end"])
    trials = sorted(os.listdir(out / "tiny"))
    assert len(trials) == 4  # 2 seeds x 2 aggregators
    for t in trials:
        d = out / "tiny" / t
        for fn in ("params.json", "params.pkl", "result.json", "progress.csv", "DONE"):
            assert (d / fn).exists(), fn
        rows = [json.loads(l) for l in open(d / "result.json")]
        assert [r["training_iteration"] for r in rows] == [1, 2, 3, 4]
        assert "acc_top_1" in rows[1] and "acc_top_1" not in rows[0]  # evaluation_interval = 2
        assert rows[-1]["server lr"] == [0.1]
        params = pickle.load(open(d / "params.pkl", "rb"))
        # the contract of tuned_examples/visualization/visualize.py
        assert params["server_config"]["aggregator"]["type"] in ("Median", "Trimmedmean")
        assert params["num_malicious_clients"] == 1
        ck = [c for c in os.listdir(d) if c.startswith("checkpoint_")]
        assert ck and (d / ck[0] / "algorithm_state.pkl").exists()
        state = pickle.load(open(d / ck[0] / "algorithm_state.pkl", "rb"))
        assert "config" in state  # the reference's whole checkpoint; the rest is ours
        assert state["config"].num_malicious_clients == 1


def test_checkpoint_resume_is_exact(tmp_path):
    """2 rounds + checkpoint + resume for 2 more == 4 rounds straight (stateful aggregator,
    server momentum, LR schedule, sampler state all restored)."""
    data = str(tmp_path / "data")
    y4, y2 = tmp_path / "e4.yaml", tmp_path / "e2.yaml"
    tiny_yaml(str(y4), data, iters=4, grid=False)
    tiny_yaml(str(y2), data, iters=2, grid=False)
    a, b = tmp_path / "A", tmp_path / "B"
    cli(["file", str(y4), "--local-dir", str(a), "--local-mode"])
    cli(["file", str(y2), "--local-dir", str(b), "--local-mode", "--checkpoint-freq", "2"])
    cli(["file", str(y4), "--local-dir", str(b), "--local-mode", "--resume"])
    ta, tb = os.listdir(a / "tiny")[0], os.listdir(b / "tiny")[0]
    ra = [json.loads(l) for l in open(a / "tiny" / ta / "result.json")]
    rb = [json.loads(l) for l in open(b / "tiny" / tb / "result.json")]
    assert [r["training_iteration"] for r in rb] == [1, 2, 3, 4]
    for x, yv in zip(ra, rb):
        assert abs(x["train_loss"] - yv["train_loss"]) < 1e-6
    assert abs(ra[-1]["acc_top_1"] - rb[-1]["acc_top_1"]) < 1e-9
    assert abs(ra[-1]["ce_loss"] - rb[-1]["ce_loss"]) < 1e-6


def test_run_callback_flags(tmp_path):
    cfg = {"dataset_config": {"type": "mnist", "num_clients": 4, "num_classes": 10, "train_batch_size": 8,
                              "n_train": 100, "n_test": 40, "data_root": str(tmp_path / "d"), "use_cache": False},
           "global_model": "mlp", "device": "cpu", "evaluation_interval": 1}
    cli(["--algo", "FEDAVG", "--config", json.dumps(cfg), "--stop", '{"training_iteration": 2}', "--local-dir",
         str(tmp_path / "r")])
    t = os.listdir(tmp_path / "r" / "default")[0]
    assert len(open(tmp_path / "r" / "default" / t / "result.json").read().strip().splitlines()) == 2


def test_stop_on_metric_threshold():
    from blades_amd.tune import should_stop

    assert should_stop({"train_loss": 1e6}, {"train_loss": 100000})
    assert not should_stop({"train_loss": 2.0}, {"train_loss": 100000, "training_iteration": 10})
    assert should_stop({"train_loss": float("nan")}, {"train_loss": 100000})


def test_early_stopping_schedulers(tmp_path):
    """File-based MedianStoppingRule / ASHA (tune/schedulers.py): decisions from the trials' result.json files."""
    import json
    from blades_amd.tune.schedulers import create_scheduler

    exp = tmp_path / "exp"

    def write(name, accs, every=10):
        d = exp / name
        d.mkdir(parents=True)
        with open(d / "result.json", "w") as f:
            for i, a in enumerate(accs):
                f.write(json.dumps({"training_iteration": (i + 1) * every, "acc_top_1": a, "train_loss": 1 - a}) + "\n")

    write("good_a", [0.5, 0.7, 0.8, 0.85])
    write("good_b", [0.4, 0.6, 0.75, 0.8])
    write("good_c", [0.45, 0.65, 0.7, 0.82])
    write("mine", [0.48])
    assert not create_scheduler("FIFO").should_stop(str(exp), "mine", {"training_iteration": 20, "acc_top_1": 0.1})
    med = create_scheduler("median", metric="acc_top_1", mode="max", grace_period=10, min_samples_required=3)
    assert med.should_stop(str(exp), "mine", {"training_iteration": 20, "acc_top_1": 0.12})
    assert not med.should_stop(str(exp), "mine", {"training_iteration": 20, "acc_top_1": 0.9})
    assert not med.should_stop(str(exp), "mine", {"training_iteration": 20, "train_loss": 3.0})  # not an eval round
    lossy = create_scheduler("median", metric="train_loss", mode="min", min_samples_required=3)
    assert lossy.should_stop(str(exp), "mine", {"training_iteration": 20, "train_loss": 0.95})
    assert not lossy.should_stop(str(exp), "mine", {"training_iteration": 20, "train_loss": 0.05})
    asha = create_scheduler("asha", metric="acc_top_1", mode="max", max_t=40, grace_period=10, reduction_factor=2)
    assert asha.rungs == [10, 20]
    assert asha.should_stop(str(exp), "mine", {"training_iteration": 20, "acc_top_1": 0.2})       # bottom half
    assert not asha.should_stop(str(exp), "mine", {"training_iteration": 20, "acc_top_1": 0.72})  # top half
    assert asha.should_stop(str(exp), "mine", {"training_iteration": 40, "acc_top_1": 0.99})      # budget reached
    with pytest.raises(ValueError):
        create_scheduler("pbt")


def test_fractional_gpu_trials_are_packed():
    """``num_gpus_for_driver: 0.25`` (reference YAMLs) = four trial processes per device; whole-GPU and multi-GPU
    requests keep whole devices."""
    from blades_amd.tune.experiment import GpuSlots, TrialScheduler, _gpus_of

    assert _gpus_of({"num_gpus_for_driver": 0.25}) == 0.25
    assert _gpus_of({"num_gpus_for_driver": 0.25, "num_remote_workers": 1, "num_gpus_per_worker": 0.25}) == 0.5
    assert _gpus_of({}) == 1.0 and _gpus_of({"num_gpus_for_driver": 1}) == 1.0 and _gpus_of({"world_size": 4}) == 4.0
    s = GpuSlots(2)
    a = [s.acquire(0.25) for _ in range(4)]
    assert all(t is not None and t[0][0] == a[0][0][0] for t in a)  # packed onto one device
    whole = s.acquire(1.0)
    assert whole is not None and whole[0][0] != a[0][0][0]
    assert s.acquire(0.25) is None and s.acquire(1.0) is None
    s.release(a[0])
    assert s.acquire(0.25) is not None

    class FakeProc:  # finishes on its third poll
        def __init__(self, cmd, env=None):
            self.n, self.env = 0, env

        def poll(self):
            self.n += 1
            return 0 if self.n >= 3 else None

    trials = [{"gpus_per_trial": 0.25, "dir_name": f"t{i}"} for i in range(6)]
    sched = TrialScheduler(num_gpus=1, popen=FakeProc)
    assert sched.run(trials) == [0] * 6
    assert sched.max_in_flight == 4  # four quarter-GPU trials share the one device
