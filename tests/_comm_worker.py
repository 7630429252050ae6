import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from blades_amd.parallel.communicator import CommunicationSet, Communicator  # noqa: E402
from blades_amd.parallel.runtime import init_runtime, shutdown_runtime  # noqa: E402

rt = init_runtime(use_gpu=False)
G, r = rt.world_size, rt.rank
comm = Communicator(pkg_size=(5,), buffer_len=2 * (G - 1))
got = comm.local_round([None if r == 0 else torch.full((5,), float(r)), None if r == 0 else torch.full((5,), 10.0 + r)])
if r == 0:
    vals = sorted(float(t[0]) for t in got)
    assert vals == sorted([float(i) for i in range(1, G)] + [10.0 + i for i in range(1, G)]), vals
t = torch.arange(4.0) if r == 1 else torch.zeros(4)
CommunicationSet().broadcast(t, root=1)
assert torch.equal(t, torch.arange(4.0))
rows = [2, 3, 1][:G]
Dl = 4
U = torch.arange(rows[r] * G * Dl, dtype=torch.float32).reshape(rows[r], G * Dl) + 1000 * r
col = comm.all_to_all_rows(U, rows)
full = comm.all_gather_rows(U, rows)
assert col.shape == (sum(rows), Dl) and full.shape == (sum(rows), G * Dl)
assert torch.equal(col, full[:, r * Dl:(r + 1) * Dl])
if r == 0:
    open(sys.argv[1], "w").write("ok")
shutdown_runtime()
