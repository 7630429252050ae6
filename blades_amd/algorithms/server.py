"""The FL server: global model, robust aggregator, server optimizer.

Reference ``fllib/algorithms/server.py``: aggregate the client updates, hand the
negated aggregate to ``torch.optim.SGD`` as a gradient, step, then advance the
piecewise LR schedule — i.e. ``theta <- theta + lr * agg`` (+ momentum).

Here the global weights are **one flat fp32 vector** ``theta`` (the model's
parameters are views into it) and the step is the fused kernel
``ops.server_step_`` applied to the rank's column slice, followed by an
all-gather of the slices (identity on one rank).  The optimizer ``type`` key is
ignored like in the reference — it is always SGD.
"""
from typing import Dict, Optional, Sequence, Union

import torch
import torch.distributed as dist
from torch import Tensor

from blades_amd import ops
from blades_amd.aggregators import Mean
from blades_amd.parallel.context import ShardContext, local_context, padded_width
from blades_amd.tasks import Task, TaskSpec
from blades_amd.utils.from_config import from_config
from .lr_schedule import LearningRateSchedule


class Server(LearningRateSchedule):
    def __init__(self, task_spec: TaskSpec, device: str, optimizer_config: dict, aggregator_config,
                 ctx: Optional[ShardContext] = None):
        optimizer_config = optimizer_config or {}
        lr = optimizer_config.get("lr", 0.1)
        lr_schedule = optimizer_config.get("lr_schedule", None)
        self.momentum = float(optimizer_config.get("momentum", 0) or 0)
        self.config = {}
        LearningRateSchedule.__init__(self, lr, lr_schedule)
        self.device = device
        self._task = task_spec.build(self.device)
        self._model = self._task.model
        self._aggregator = self._create_aggregator(aggregator_config)
        self._states: Dict[str, Optional[Tensor]] = {}
        self._steps = 0
        self.nan_guard = True
        self._flatten_model(ctx)
        self.skipped_steps = torch.zeros((), device=self.theta.device)

    # -- flat parameter storage -------------------------------------------
    def _flatten_model(self, ctx: Optional[ShardContext]):
        params = [p for p in self._model.parameters() if p.requires_grad]
        D = sum(p.numel() for p in params)
        self.ctx = ctx if ctx is not None else local_context(D)
        self.ctx.d_total = D
        G = self.ctx.world_size
        self.Dl = self.ctx.d_local
        dev = params[0].device if params else torch.device(self.device)
        self.theta = torch.zeros(self.Dl * G, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.theta[off: off + n].copy_(p.data.reshape(-1))
                p.data = self.theta[off: off + n].view_as(p)
                off += n
        self.D = D
        self.momentum_buf = torch.zeros(self.Dl, dtype=torch.float32, device=dev) if self.momentum else None
        if self.ctx.distributed:  # identical start on every rank
            self.ctx.broadcast(self.theta, src=0)

    def _create_aggregator(self, config):
        return from_config(Mean, config)

    @property
    def aggregator(self):
        return self._aggregator

    @property
    def task(self) -> Task:
        return self._task

    @property
    def lr(self) -> float:
        return self.cur_lr

    def register_state(self, name: str, tensor: Optional[Tensor]) -> None:
        self._states[name] = tensor

    def get_state(self, name: str):
        return self._states[name]

    def set_state(self, name: str, tensor: Optional[Tensor]) -> None:
        self._states[name] = tensor

    def zero_grad(self, set_to_none: bool = False):
        for p in self._model.parameters():
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def get_global_model(self):
        return self._model

    # -- the step -----------------------------------------------------------
    def step(self, local_updates: Union[Tensor, Sequence[Tensor]], global_vars: Optional[dict] = None,
             owned: bool = False) -> dict:
        """``owned=True``: ``local_updates`` is the engine's resident update matrix, which the aggregator may consume
        in place (norm clipping rescales rows) — no protective copy of a [K, D] matrix (45 GB at K = 1000, ResNet-18)."""
        self.update_global_model(local_updates, owned=owned)
        self.on_global_var_update(global_vars)
        return {"server lr": [self.cur_lr]}

    def _as_matrix(self, local_updates) -> Tensor:
        if isinstance(local_updates, Tensor):
            return local_updates
        ups = [u for u in local_updates]
        if any(u is None for u in ups):
            raise ValueError("a client produced no update (None) and no adversary filled it in")
        U = torch.stack([u.reshape(-1).to(self.theta.device, torch.float32) for u in ups], dim=0)
        if self.ctx.distributed:
            lo = self.ctx.col_offset
            pad = torch.zeros(U.shape[0], self.Dl * self.ctx.world_size - U.shape[1], device=U.device)
            U = torch.cat([U, pad], dim=1)[:, lo: lo + self.Dl].contiguous()
        return U

    def update_global_model(self, local_updates, owned: bool = False) -> None:
        U = self._as_matrix(local_updates)
        if (owned and isinstance(local_updates, Tensor) and U.dtype == torch.float32 and U.dim() == 2
                and U.stride(1) == 1 and hasattr(self._aggregator, "aggregate")):
            agg = self._aggregator.aggregate(U, self.ctx)  # in place on the engine's matrix
        else:
            agg = self._aggregator(U, self.ctx)  # [Dl]; __call__ copies a caller-owned tensor before clipping rows
        bad = None
        if self.nan_guard:
            # A non-finite aggregate (diverged clients, an attack the defence let
            # through) would poison theta for good: skip the step instead.  Decided
            # on the device, no host sync; all ranks agree through the all-reduce.
            lo = self.ctx.col_offset
            if ops.guarded_step_available(self.theta[lo: lo + self.Dl], agg) and self.skipped_steps.dtype == torch.float32:
                # one flag kernel, and the step kernel reads the flag (ops/csrc/optim.hip)
                bad = ops.nonfinite_flag(agg)
                if self.ctx.distributed:
                    self.ctx.all_reduce(bad)
            else:
                flag = (~torch.isfinite(agg).all()).float()
                if self.ctx.distributed:
                    self.ctx.all_reduce(flag)
                agg = torch.where(flag > 0, torch.zeros_like(agg), agg)
                self.skipped_steps = self.skipped_steps + flag.clamp(max=1)
        self.apply_aggregate(agg, bad)

    def apply_aggregate(self, agg: Tensor, bad: Optional[Tensor] = None) -> None:
        """``bad``: fp32 device scalar, > 0 = take this step with a zero aggregate and count it as skipped."""
        lo = self.ctx.col_offset
        sl = self.theta[lo: lo + self.Dl]
        buf = self.momentum_buf if self.momentum_buf is not None else sl
        if bad is not None:
            ops.server_step_guarded_(sl, buf, agg, self.cur_lr, self.momentum, self._steps == 0, bad,
                                     self.skipped_steps)
        else:
            ops.server_step_(sl, buf, agg.to(sl.dtype), self.cur_lr, self.momentum, self._steps == 0)
        self._steps += 1
        if self.ctx.distributed:
            dist.all_gather_into_tensor(self.theta, sl.clone(), group=self.ctx.group)

    # -- checkpoint ---------------------------------------------------------
    def _aggregator_state(self) -> dict:
        """Aggregator state with column-sharded vectors (Centeredclipping's momentum) gathered to full width, so
        a checkpoint written by rank 0 holds every rank's slice and can be resumed at any world size."""
        st = self._aggregator.state_dict() if hasattr(self._aggregator, "state_dict") else {}
        mom = getattr(self._aggregator, "momentum", None)
        if "momentum" in st and isinstance(mom, Tensor):
            full = self.ctx.all_gather_cols(mom) if self.ctx.distributed else mom
            st = dict(st)
            st["momentum"] = full[: self.D].detach().cpu()
        return st

    def state_dict(self) -> dict:
        return {
            "theta": self.theta[: self.D].detach().cpu(),
            "momentum_buf": None if self.momentum_buf is None else self.ctx.all_gather_cols(
                self.momentum_buf).detach().cpu(),
            "cur_lr": self.cur_lr,
            "steps": self._steps,
            "aggregator": self._aggregator_state(),
        }

    def load_state_dict(self, s: dict) -> None:
        self.theta[: self.D].copy_(s["theta"].to(self.theta.device))
        if self.momentum_buf is not None and s.get("momentum_buf") is not None:
            full = torch.zeros(self.Dl * self.ctx.world_size, device=self.theta.device)
            full[: self.D] = s["momentum_buf"].to(self.theta.device)
            lo = self.ctx.col_offset
            self.momentum_buf.copy_(full[lo: lo + self.Dl])
        self.cur_lr = s["cur_lr"]
        self._steps = s["steps"]
        if hasattr(self._aggregator, "load_state_dict"):
            st = s.get("aggregator", {})
            if isinstance(st.get("momentum", None), Tensor):
                # stored full-width: take this rank's column slice (zero-padded up to the shard width)
                full = torch.zeros(self.Dl * self.ctx.world_size, device=self.theta.device)
                m = st["momentum"].to(self.theta.device)
                full[: m.numel()] = m
                lo = self.ctx.col_offset
                st = dict(st)
                st["momentum"] = full[lo: lo + self.Dl].clone()
            self._aggregator.load_state_dict(st)
