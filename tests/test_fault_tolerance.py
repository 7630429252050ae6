"""Fault injection, NaN guards and bit-reproducibility (SURVEY §5.2/§5.3 — the reference has no such tests)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from .simple_fixtures import simple_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _algo(agg="Mean", **kw):
    torch.manual_seed(0)
    return simple_config(num_clients=6, agg=agg, **kw)


def test_nan_row_poisons_mean_and_server_guard_skips_the_step():
    cfg = _algo("Mean").fault_tolerance(fault_injection={"round": 2, "kind": "nan_row", "row": 1})
    algo = cfg.build()
    algo.train()
    theta1 = algo.server.theta.clone()
    algo.train()  # aggregate is NaN -> the server must not move
    assert torch.equal(algo.server.theta, theta1)
    assert float(algo.server.skipped_steps) == 1.0
    algo.train()
    assert torch.isfinite(algo.server.theta).all() and not torch.equal(algo.server.theta, theta1)


def test_row_guard_drops_only_the_bad_row():
    ref = _algo("Mean").build()
    ref.train()
    cfg = _algo("Mean").fault_tolerance(guard_rows=True,
                                        fault_injection=[{"round": 1, "kind": "inf_row", "row": 2}])
    algo = cfg.build()
    algo.train()
    assert algo.dropped_rows == 1
    assert float(algo.server.skipped_steps) == 0.0
    U = algo.update_matrix
    assert torch.isfinite(U).all() and float(U[2].abs().sum()) == 0.0
    # the other rows are the ones of the clean run
    Ur = ref.update_matrix
    keep = [i for i in range(U.shape[0]) if i != 2]
    assert torch.equal(U[keep], Ur[keep])


def test_median_survives_a_corrupt_row_without_any_guard():
    cfg = _algo("Median").fault_tolerance(fault_injection={"round": 1, "kind": "scale_row", "row": 0, "value": 1e30})
    algo = cfg.build()
    algo.train()
    assert torch.isfinite(algo.server.theta).all()
    assert float(algo.server.theta.abs().max()) < 1e3


def test_fault_event_from_environment(monkeypatch):
    monkeypatch.setenv("BLADES_INJECT_FAULT", json.dumps({"round": 1, "kind": "nan_row", "row": 0}))
    algo = _algo("Mean").build()
    algo.train()
    assert float(algo.server.skipped_steps) == 1.0


def test_two_runs_are_bitwise_identical():
    """Deterministic mode: the same seed gives the same bits, round after round (no atomics in any
    reduction, counter-based RNG)."""
    outs = []
    for _ in range(2):
        torch.manual_seed(0)
        algo = simple_config(num_clients=6, agg="Median", f=2,
                             adversary={"type": "blades.adversaries.ALIEAdversary"}).build()
        for _ in range(3):
            algo.train()
        outs.append(algo.server.theta.clone())
    assert torch.equal(outs[0], outs[1])


def test_killed_rank_takes_the_job_down(tmp_path):
    """A rank that dies mid-run must not leave its peer hanging in a collective: the launcher tears the job
    down and reports the failing status."""
    script = tmp_path / "job.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from blades_amd.parallel.runtime import init_runtime, shutdown_runtime
        rt = init_runtime(use_gpu=False)
        import torch
        torch.manual_seed(0)
        from tests.simple_fixtures import simple_config
        algo = simple_config(num_clients=6, agg="Mean").fault_tolerance(
            fault_injection={{"round": 2, "kind": "kill_rank", "rank": 1, "code": 17}}).build()
        for _ in range(4):
            algo.train()
        shutdown_runtime()
    """))
    p = subprocess.run([sys.executable, "-m", "blades_amd.parallel.launcher", "--nproc", "2", "--", str(script)],
                       cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert p.returncode == 17, p.stderr[-2000:]
