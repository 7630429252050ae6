"""bench.py --gpus N must start N ranks by itself when it is run as a plain script (the driver's call)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", BLADES_SYNTHETIC_DATA="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu", "--steps", "2", "--warmup", "1",
                        "--clients", "8", "--model", "mlp", "--dataset", "mnist", "--batch", "8",
                        "--samples-per-client", "16", *args], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_self_launches_ranks():
    one = _bench("--gpus", "1")
    assert one["n_gpus"] == 1 and one["steps"] == 2 and one["value"] > 0 and one["dtype"] == "fp32"
    two = _bench("--gpus", "2")
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["warmup"] == 1
    assert two["comm_bytes"]["alltoall"] > 0
    assert two["config"]["num_clients"] == 8 and two["scaling"] == "strong"
    # same job, same loss whatever the number of ranks
    assert abs(one["train_loss"] - two["train_loss"]) < 1e-4
    # the line says what the collectives really ran on
    assert one["pg_world_size"] == 1 and "none" in one["backend"]
    assert two["pg_world_size"] == 2 and two["backend"] == "gloo"
    assert [d["rank"] for d in two["rank_devices"]] == [0, 1] and two["exchange_ms"] >= 0.0
    assert two["update_layout"] == "alltoall"


def test_bench_refuses_a_process_group_of_the_wrong_size():
    """--gpus 2 inside a one-rank environment (WORLD_SIZE=1 given, so bench does not launch ranks itself) must fail
    instead of printing a line labelled n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", BLADES_SYNTHETIC_DATA="1", WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu", "--steps", "1", "--warmup", "0",
                        "--clients", "8", "--model", "mlp", "--dataset", "mnist", "--batch", "8",
                        "--samples-per-client", "16", "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_one_rank_forced_collectives_reports_backend():
    env_backup = os.environ.get("BLADES_FORCE_COLLECTIVES")
    os.environ["BLADES_FORCE_COLLECTIVES"] = "1"
    try:
        one = _bench("--gpus", "1")
    finally:
        if env_backup is None:
            os.environ.pop("BLADES_FORCE_COLLECTIVES", None)
        else:
            os.environ["BLADES_FORCE_COLLECTIVES"] = env_backup
    assert one["pg_world_size"] == 1 and one["backend"] == "gloo"  # "nccl" on a GPU box (tests/test_rccl_single_rank.py)
