"""Communicator: gather into the server's buffer list, broadcast, update-matrix re-layouts (gloo, 3 ranks)."""
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def test_buffer_list_semantics():
    from blades_amd.parallel.communicator import _BufferList

    b = _BufferList(3, (2,), "cpu")
    assert len(b) == 3 and not b.filled()
    b.take(1)[0].fill_(1.0)
    assert len(b.filled()) == 1 and b.stack().shape == (1, 2)
    b.reset()
    assert not b.filled()


def test_single_process_identity():
    from blades_amd.parallel.communicator import Communicator

    c = Communicator(pkg_size=(3,))
    U = torch.randn(4, 6)
    assert c.all_to_all_rows(U, [4]) is U
    assert torch.equal(c.broadcast_tensor(U.clone()), U)


def test_three_rank_collectives(tmp_path):
    out = tmp_path / "ok"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr",
           "127.0.0.1", "--master-port", "29733", os.path.join(HERE, "_comm_worker.py"), str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert out.read_text() == "ok"
