"""Device-side per-client batch sampler.

Every local client has a row of a permutation table in HBM and a cursor kept on
the **host** (epoch boundaries are a pure function of shard length and batch
size, so no device->host sync is ever needed).  ``next_indices`` returns the
``[Kc, B]`` global sample indices of one batch per client; the batch itself is
one gather from the concatenated :class:`~blades_amd.datasets.DeviceStore`.

Semantics vs the reference generator (``fldataset.py:230-251``): permutation per
epoch, consecutive slices of ``B``.  The epoch's ragged tail (fewer than ``B``
samples) is **dropped** here because all clients of a chunk must share one
batch shape; clients whose whole shard is smaller than ``B`` wrap around.
"""
import numpy as np
import torch


class DeviceSampler:
    def __init__(self, lengths: np.ndarray, offsets: np.ndarray, batch_size: int, device, seed: int = 0):
        self.n = np.asarray(lengths, dtype=np.int64)
        self.offsets_t = torch.as_tensor(np.asarray(offsets, dtype=np.int64), device=device)
        self.n_t = torch.as_tensor(self.n, device=device)
        self.B = int(batch_size)
        self.device = device
        self.K = len(self.n)
        self.nmax = int(self.n.max()) if self.K else 0
        self.gen = torch.Generator()  # CPU generator: the same permutations on every device type
        self.gen.manual_seed(int(seed))
        self.cursor = np.zeros(self.K, dtype=np.int64)
        self.perm = torch.zeros(self.K, max(self.nmax, 1), dtype=torch.int64, device=device)
        self._ar = torch.arange(self.B, device=device)
        if self.K:
            self._reshuffle(np.arange(self.K))

    def _reshuffle(self, rows: np.ndarray):
        r = torch.as_tensor(rows, device=self.device)
        keys = torch.rand(len(rows), self.nmax, generator=self.gen).to(self.device)
        invalid = torch.arange(self.nmax, device=self.device).unsqueeze(0) >= self.n_t[r].unsqueeze(1)
        keys.masked_fill_(invalid, 2.0)
        self.perm[r] = keys.argsort(dim=1)
        self.cursor[rows] = 0

    def next_indices(self, rows: np.ndarray = None) -> torch.Tensor:
        """Global sample indices ``[len(rows), B]`` (advances those rows' cursors)."""
        if rows is None:
            rows = np.arange(self.K)
        full = self.n[rows] >= self.B
        need = full & (self.cursor[rows] + self.B > self.n[rows])
        if need.any():
            self._reshuffle(rows[need])
        r = torch.as_tensor(rows, device=self.device)
        cur = torch.as_tensor(self.cursor[rows], device=self.device)
        pos = (cur.unsqueeze(1) + self._ar.unsqueeze(0)) % self.n_t[r].unsqueeze(1)
        idx = self.perm[r].gather(1, pos) + self.offsets_t[r].unsqueeze(1)
        self.cursor[rows] += self.B
        small = ~full
        if small.any():  # wrap-around shards reshuffle once a full pass is consumed
            done = small & (self.cursor[rows] >= self.n[rows])
            if done.any():
                self._reshuffle(rows[done])
        return idx

    def state_dict(self):
        return {"cursor": self.cursor.copy(), "perm": self.perm.cpu(), "gen": self.gen.get_state()}

    def load_state_dict(self, s):
        self.cursor = np.asarray(s["cursor"]).copy()
        self.perm.copy_(s["perm"].to(self.device))
        self.gen.set_state(s["gen"])
