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
