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
    def __init__(self, lengths: np.ndarray, offsets: np.ndarray, batch_size: int, device, seed: int = 0,
                 client_uids: np.ndarray = None):
        self.n = np.asarray(lengths, dtype=np.int64)
        self.offsets_t = torch.as_tensor(np.asarray(offsets, dtype=np.int64), device=device)
        self.n_t = torch.as_tensor(self.n, device=device)
        self.B = int(batch_size)
        self.device = device
        self.K = len(self.n)
        self.nmax = int(self.n.max()) if self.K else 0
        # Shuffle keys are a counter-based hash of (seed, client uid, epoch, position): the batches a
        # client sees do not depend on the device type, the rank it runs on, or the number of ranks.
        self.seed = int(seed)
        self.uids = np.arange(self.K, dtype=np.int64) if client_uids is None else np.asarray(client_uids,
                                                                                             dtype=np.int64)
        self.epoch = np.zeros(self.K, dtype=np.int64)
        self.cursor = np.zeros(self.K, dtype=np.int64)
        self.perm = torch.zeros(self.K, max(self.nmax, 1), dtype=torch.int64, device=device)
        # device mirror of the cursors: steady-state batches are indexed without any host->device copy
        self.cursor_t = torch.zeros(self.K, dtype=torch.int64, device=device)
        self._row_cache = {}
        self._ar = torch.arange(self.B, device=device)
        if self.K:
            self._reshuffle(np.arange(self.K))

    def _reshuffle(self, rows: np.ndarray):
        from blades_amd import ops

        r = torch.as_tensor(rows, device=self.device)
        uid = torch.as_tensor(self.uids[rows], device=self.device)
        ep = torch.as_tensor(self.epoch[rows], device=self.device)
        keys = ops.shuffle_keys(uid, ep, self.n_t[r], self.nmax, self.seed)  # invalid positions sort last
        self.epoch[rows] += 1
        self.perm[r] = keys.argsort(dim=1)
        self.cursor[rows] = 0
        self.cursor_t[r] = 0

    def next_indices(self, rows: np.ndarray = None) -> torch.Tensor:
        """Global sample indices ``[len(rows), B]`` (advances those rows' cursors)."""
        if rows is None:
            rows = np.arange(self.K)
        self.prepare(rows)
        idx = self.gather_indices(self.rows_on_device(rows))
        self.finish(rows)
        return idx

    # The three steps of ``next_indices``; a captured HIP graph replays the middle one (pure device work on
    # static tensors) while the host keeps doing the outer two.
    def rows_on_device(self, rows: np.ndarray):
        """(r, n[r], offsets[r]) device tensors of a row set — cached, chunks repeat every round."""
        key = rows.tobytes()
        hit = self._row_cache.get(key)
        if hit is None:
            if len(self._row_cache) > 256:
                self._row_cache.clear()
            r = torch.as_tensor(rows, device=self.device)
            hit = self._row_cache[key] = (r, self.n_t[r].unsqueeze(1), self.offsets_t[r].unsqueeze(1))
        return hit

    def prepare(self, rows: np.ndarray) -> None:
        """Host bookkeeping before a batch: reshuffle the rows whose epoch is used up."""
        full = self.n[rows] >= self.B
        need = full & (self.cursor[rows] + self.B > self.n[rows])
        if need.any():
            self._reshuffle(rows[need])

    def gather_indices(self, dev_rows) -> torch.Tensor:
        """Device part: indices of the next batch of every row, cursors advanced (no host data involved)."""
        r, n_r, off_r = dev_rows
        cur = self.cursor_t[r]
        pos = (cur.unsqueeze(1) + self._ar.unsqueeze(0)) % n_r
        idx = self.perm[r].gather(1, pos) + off_r
        self.cursor_t.index_add_(0, r, torch.full_like(cur, self.B))
        return idx

    def finish(self, rows: np.ndarray) -> None:
        """Host bookkeeping after a batch: mirror the cursor advance; wrap-around shards (smaller than a batch)
        reshuffle once a full pass is consumed."""
        self.cursor[rows] += self.B
        small = self.n[rows] < self.B
        if small.any():
            done = small & (self.cursor[rows] >= self.n[rows])
            if done.any():
                self._reshuffle(rows[done])

    # Checkpoint state is per *client uid*, not per local row: permutations are a pure function of
    # (seed, uid, epoch), so (cursor, epoch) is all there is to save, and a run can resume on a different
    # number of ranks (each rank picks the uids it owns).
    def state_dict(self):
        return {int(u): (int(c), int(e)) for u, c, e in zip(self.uids, self.cursor, self.epoch)}

    def load_state_dict(self, s):
        rows = np.array([r for r, u in enumerate(self.uids) if int(u) in s], dtype=np.int64)
        if len(rows) == 0:
            return
        st = [s[int(self.uids[r])] for r in rows]
        self.epoch[rows] = np.array([e for _, e in st], dtype=np.int64) - 1
        self._reshuffle(rows)  # regenerates the permutation of the saved epoch (and bumps epoch back)
        self.cursor[rows] = np.array([c for c, _ in st], dtype=np.int64)
        self.cursor_t[torch.as_tensor(rows, device=self.device)] = torch.as_tensor(self.cursor[rows],
                                                                                   device=self.device)
