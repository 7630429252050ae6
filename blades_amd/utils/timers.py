"""Per-phase timers reported in every result row.

GPU phases are timed with HIP events recorded on the current stream and
resolved lazily at ``snapshot()`` — timing a phase never adds a host sync to it.
"""
import time
from collections import defaultdict
from contextlib import contextmanager

import torch


class PhaseTimers:
    def __init__(self, device):
        self.gpu = torch.device(device).type == "cuda"
        self._pending = []
        self.totals = defaultdict(float)
        self.counts = defaultdict(int)
        self.last = {}

    @contextmanager
    def __call__(self, name: str):
        if self.gpu:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            try:
                yield
            finally:
                b.record()
                self._pending.append((name, a, b))
        else:
            t0 = time.perf_counter()
            try:
                yield
            finally:
                self._add(name, (time.perf_counter() - t0) * 1e3)

    def _add(self, name, ms):
        self.totals[name] += ms
        self.counts[name] += 1
        self.last[name] = ms

    def _resolve(self, block: bool):
        keep = []
        for name, a, b in self._pending:
            if block:
                b.synchronize()
            if b.query():
                self._add(name, a.elapsed_time(b))
            else:
                keep.append((name, a, b))
        self._pending = keep

    def reset(self):
        """Forget everything measured so far (e.g. warm-up rounds)."""
        if self.gpu:
            self._resolve(True)
        self.totals.clear()
        self.counts.clear()
        self.last.clear()

    def snapshot(self, block: bool = False) -> dict:
        """Last completed duration of every phase, in ms."""
        if self.gpu:
            self._resolve(block)
        return {f"{k}_ms": round(v, 4) for k, v in self.last.items()}

    def means(self) -> dict:
        if self.gpu:
            self._resolve(True)
        return {f"{k}_ms": self.totals[k] / max(self.counts[k], 1) for k in self.totals}
