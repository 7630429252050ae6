"""The distributed code paths on ONE rank with a real process group (``BLADES_FORCE_COLLECTIVES=1``).

The development boxes have one GPU, so a multi-rank RCCL job cannot run there; what can run is a one-rank
``backend="nccl"`` group in which every collective of the framework is really issued through RCCL: padded column
shards, the update exchange (both layouts), ``all_reduce`` of Gram / norm / count packs, ``all_gather_into_tensor``
of theta, ``gather_object`` / ``all_gather_object`` for checkpoints.  That catches what gloo tests cannot: host
tensors, unsupported dtypes or non-contiguous buffers handed to RCCL.  The result must equal the plain
single-process run.  The same test runs with gloo on the CPU."""
import json
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    ({"type": "Median"}, {"type": "blades.adversaries.ALIEAdversary"}, "alltoall"),
    ({"type": "Trimmedmean"}, {"type": "blades.adversaries.IPMAdversary", "scale": 0.1}, "alltoall"),
    ({"type": "Multikrum", "k": 3}, {"type": "blades.adversaries.MinMaxAdversary"}, "alltoall"),
    ({"type": "GeoMed"}, {"type": "blades.adversaries.NoiseAdversary"}, "allgather"),
    ({"type": "Centeredclipping"}, {"type": "blades.adversaries.SignFlipAdversary"}, "allgather"),
    ({"type": "Signguard"}, {"type": "blades.adversaries.ALIEAdversary"}, "alltoall"),
    ({"type": "Clippedclustering"}, {"type": "blades.adversaries.LabelFlipAdversary"}, "alltoall"),
    ({"type": "DnC", "sub_dim": 500}, {"type": "blades.adversaries.AdaptiveAdversary"}, "alltoall"),
    ({"type": "Mean"}, {"type": "blades.adversaries.SignGuardAdversary"}, "alltoall"),
    ({"type": "Clippedclustering", "linkage": "single"},
     {"type": "blades.adversaries.AttackclippedclusteringAdversary"}, "alltoall"),
]


def _run(device, forced, root, out, port, cases=CASES, model="mlp", extra_env=None):
    env = dict(os.environ, OMP_NUM_THREADS="1", BLADES_SYNTHETIC_DATA="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BLADES_FORCE_COLLECTIVES="1" if forced else "0")
    env.update(extra_env or {})
    spec = {"K": 10, "f": 2, "rounds": 3, "root": root, "device": device, "cases": cases, "model": model}
    r = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_worker.py"), json.dumps(spec), out], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    return torch.load(out, weights_only=True)


def _check(device, tmp_path, port, cases=CASES, **kw):
    plain = _run(device, False, str(tmp_path), str(tmp_path / "plain.pt"), port, cases, **kw)
    forced = _run(device, True, str(tmp_path), str(tmp_path / "forced.pt"), port + 1, cases, **kw)
    assert plain["backend"] is None
    assert forced["backend"] == ("nccl" if device == "cuda" else "gloo")
    for i, (agg, _, _) in enumerate(cases):
        a, b = plain[str(i)], forced[str(i)]
        assert torch.isfinite(b["theta"]).all()
        tol = 1e-3 if agg["type"] == "GeoMed" else 1e-4
        torch.testing.assert_close(b["theta"], a["theta"], rtol=1e-3, atol=tol)
        torch.testing.assert_close(b["theta_resumed"], a["theta_resumed"], rtol=1e-3, atol=tol)
        assert abs(a["train_loss"] - b["train_loss"]) < 1e-4
        assert abs(a["acc"] - b["acc"]) < 1e-6
        assert b["comm_bytes"].get("allreduce", 0) >= 0


def test_forced_collectives_gloo(tmp_path):
    _check("cpu", tmp_path, 29871)


@pytest.mark.gpu
def test_forced_collectives_rccl(tmp_path):
    _check("cuda", tmp_path, 29881)


@pytest.mark.gpu
def test_streamed_chunks_ship_pieces_one_chunk_behind_rccl(tmp_path):
    """Eager chunks alternating over two HIP streams (the mode of the ResNet rounds) with the exchange issued one
    chunk behind: three chunks per round, both update layouts, against the plain single-process run."""
    _check("cuda", tmp_path, 29891, cases=[CASES[0], CASES[4], CASES[3]], model="cnn",
           extra_env={"BLADES_HIP_GRAPH": "0", "BLADES_CHUNK_STREAMS": "2"})
