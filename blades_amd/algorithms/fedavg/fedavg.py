"""FedAvg / FedSGD under attack — the FL round loop.

Same contract as ``blades/algorithms/fedavg/fedavg.py`` (config keys,
``training_step`` / ``evaluate`` results, first ``num_malicious_clients``
clients are the adversary's), different machine mapping:

====================  ======================================================
reference (Ray)       here (one process per MI355X)
====================  ======================================================
weights -> W actors   nothing to send: ``theta`` is replicated, refreshed by
(pickled closure)     the all-gather that ends the server step
K RPCs, one client    ``ClientEngine.train_round``: all clients of the rank
at a time per actor   batched through shared/per-client-weight kernels
K result dicts ->     update matrix stays in HBM; one RCCL all-to-all turns
driver (plasma)       the row shard into the column shard
adversary on a        adversary forges rows of the resident column shard
``vstack``            (fused mean/std kernels + tiny all-reduces)
aggregator on the     aggregator kernels on the column shard; fused server
CPU, ``opt.step()``   step on the slice; all-gather of ``theta``
====================  ======================================================
"""
import time
from collections import defaultdict
from typing import Dict, List, Optional

import numpy as np
import torch

from blades_amd.adversaries import AdversaryConfig
from blades_amd.algorithms import Algorithm, AlgorithmConfig
from blades_amd.clients import ClientConfig
from blades_amd.constants import CLIENT_UPDATE, NUM_GLOBAL_STEPS
from blades_amd.datasets.catalog import DatasetCatalog
from blades_amd.engine.trainer import ClientEngine
from blades_amd.parallel.runtime import get_runtime
from blades_amd.parallel.worker_group_config import WorkerGroupConfig  # noqa: F401  (re-exported like the reference module)
from blades_amd.types import NotProvided
from blades_amd.utils.from_config import deep_update
from blades_amd.utils.timers import PhaseTimers
from blades_amd.utils.torch_utils import set_torch_seed


class FedavgConfig(AlgorithmConfig):
    def __init__(self, algo_class=None):
        super().__init__(algo_class=algo_class or Fedavg)
        self.adversary_config = {}
        self.num_malicious_clients = 0
        # fault tolerance (the reference has none of its own)
        self.guard_rows = False        # zero client rows that contain NaN/Inf before attack + aggregation
        self.fault_injection = None    # test hook: dict or list of dicts, see Fedavg._inject_faults

    def fault_tolerance(self, *, guard_rows: Optional[bool] = NotProvided, fault_injection=NotProvided):
        """``guard_rows``: drop (zero) update rows with non-finite entries.  ``fault_injection``: one or more
        ``{"round": r, "kind": "nan_row" | "inf_row" | "scale_row" | "kill_rank", "row": i, "rank": k,
        "value": v}`` events, also settable through the ``BLADES_INJECT_FAULT`` environment variable (JSON)."""
        if guard_rows is not NotProvided:
            self.guard_rows = bool(guard_rows)
        if fault_injection is not NotProvided:
            self.fault_injection = fault_injection
        return self

    def adversary(self, *, num_malicious_clients: Optional[int] = NotProvided,
                  adversary_config: Optional[Dict] = NotProvided):
        if adversary_config is not NotProvided:
            deep_update(self.adversary_config, adversary_config, True)
        if num_malicious_clients is not NotProvided:
            self.num_malicious_clients = num_malicious_clients
        return self

    def get_adversary_config(self) -> AdversaryConfig:
        self._require_frozen("get_adversary_config")
        return AdversaryConfig(adversary_cls=self.adversary_config.get("type", "blades_amd.adversaries.Adversary"),
                               config=self).update_from_dict(self.adversary_config)

    def get_client_config(self) -> ClientConfig:
        return ClientConfig(class_specifier="blades_amd.clients.Client").update_from_dict(self.client_config)

    def validate(self) -> None:
        super().validate()
        # DnC / Trimmedmean / Multikrum learn the number of attackers from the
        # experiment (reference fedavg.py:94-107).
        agg = self.server_config.get("aggregator", None) if isinstance(self.server_config, dict) else None
        if isinstance(agg, dict) and agg.get("type") is not None:
            t = str(agg["type"])
            if t.rpartition(".")[2] == "DnC" or "Trimmedmean" in t or "Multikrum" in t:
                agg["num_byzantine"] = self.num_malicious_clients
        if self.num_malicious_clients > self.num_clients:
            raise ValueError("`num_malicious_clients` must be smaller than or equal `num_clients`! "
                             "Simulation makes no sense otherwise.")


class _ResultRow(dict):
    """Reference-style per-client result dict backed by a row of the update matrix."""


class Fedavg(Algorithm):
    def __init__(self, config=None, logger_creator=None, **kwargs):
        self.local_results: List[dict] = []
        super().__init__(config, logger_creator, **kwargs)

    @classmethod
    def get_default_config(cls) -> AlgorithmConfig:
        return FedavgConfig()

    # ------------------------------------------------------------------
    def setup(self, config: AlgorithmConfig):
        self.config = config
        set_torch_seed(config.seed)
        self.rt = get_runtime()
        want_gpu = (config.num_gpus_for_driver or 0) > 0 or (config.num_gpus_per_worker or 0) > 0
        if config.device != "auto":
            self.device = torch.device(config.device)
        else:
            self.device = self.rt.device if (want_gpu or self.rt.device.type == "cuda") else torch.device("cpu")
        self._timers = PhaseTimers(self.device)
        self._counters = defaultdict(int)
        self._allreduce_bytes_seen = 0

        from blades_amd.parallel.context import ShardContext

        # "allgather" layout: every rank holds the whole update matrix and the whole theta and aggregates
        # replicated, so the server / aggregators / adversaries see a non-distributed context
        self.update_layout = getattr(config, "update_layout", "alltoall") if self.rt.distributed else "alltoall"
        replicated = self.update_layout == "allgather"
        ctx = ShardContext(d_total=0, world_size=1 if replicated else self.rt.world_size,
                           rank=0 if replicated else self.rt.rank,
                           force=self.rt.force_collectives and not replicated)
        self.server = config.get_server_config().build(self.device, ctx=ctx)
        self.ctx = self.server.ctx
        if replicated:
            import torch.distributed as dist

            dist.broadcast(self.server.theta, src=0)  # identical start on every rank

        self.worker_group = self._setup_worker_group()
        self._setup_dataset()
        self.client_manager = self.client_manager_cls(self._dataset.client_ids,
                                                      client_config=config.get_client_config())
        clients = self.client_manager.clients
        self.worker_group.local_execute(lambda _, client: client.setup(), clients)

        self.adversary = config.get_adversary_config().build(clients[: config.num_malicious_clients])
        self.adversary.on_algorithm_start(self)

        ccfg = config.get_client_config()
        dc = config.dataset_config
        self.engine = ClientEngine(
            task=self.server.task, dataset=self._dataset, clients=clients, runtime=self._engine_runtime(),
            client_lr=ccfg.lr, client_momentum=ccfg.momentum, num_batch_per_round=ccfg.num_batch_per_round,
            batch_size=self._dataset.train_bs, seed=config.seed, clients_per_chunk=config.clients_per_chunk,
            amp_dtype=config.amp_dtype, execution_mode=config.execution_mode,
            num_classes=dc.get("num_classes"), update_layout=self.update_layout)
        if self.engine.D != self.server.D:
            raise RuntimeError(f"engine and server disagree on the flat model width ({self.engine.D} vs "
                               f"{self.server.D}): both must enumerate utils.torch_utils.named_trainable_parameters")
        self._malicious_flags = [bool(c.is_malicious) for c in clients]
        self._benign_mask = self.engine.row_mask([not m for m in self._malicious_flags])
        self.num_benign = int(sum(1 for m in self._malicious_flags if not m))  # host copy: no sync to ask
        self._malicious_rows = torch.nonzero(self._benign_mask == 0).flatten()
        self._malicious_row_clients = torch.as_tensor(
            self.engine.plan.client_of_row[self._malicious_rows.cpu().numpy()], dtype=torch.int64)
        self.update_matrix: Optional[torch.Tensor] = None

    def _engine_runtime(self):
        from blades_amd.parallel.runtime import Runtime

        return Runtime(self.rt.rank, self.rt.world_size, self.rt.local_rank, self.device, self.rt.backend,
                       force_collectives=self.rt.force_collectives)

    def _setup_worker_group(self):
        wg = self.config.get_worker_group_config().build(device=str(self.device))
        wg.local_worker().bind_task(self.server.task)  # sequential path shares the server's model
        return wg

    def _setup_dataset(self):
        self._dataset = DatasetCatalog.get_dataset(self.config.dataset_config)
        self.worker_group.local_worker().dataset = self._dataset

    @property
    def dataset(self):
        return self._dataset

    # -- row-order helpers for adversaries ------------------------------------
    @property
    def benign_mask(self) -> torch.Tensor:
        """[K] float mask (global row order): 1 for non-malicious clients."""
        return self._benign_mask

    @property
    def malicious_rows(self) -> torch.Tensor:
        return self._malicious_rows

    @property
    def row_clients(self) -> torch.Tensor:
        """[K] global client index of every row of the update matrix."""
        return torch.as_tensor(self.engine.plan.client_of_row, dtype=torch.int64)

    @property
    def malicious_row_clients(self) -> torch.Tensor:
        """Global client index of each entry of ``malicious_rows`` (rows are rank-major, so the
        row number is not the client number when several ranks take part)."""
        return self._malicious_row_clients

    def _build_local_results(self, U: torch.Tensor):
        """Reference-style ``local_results`` view (single rank only): a list of
        ``{"id", CLIENT_UPDATE}`` dicts whose tensors alias rows of ``U``."""
        res = []
        for i, c in enumerate(self.client_manager.clients):
            row = int(self.engine.plan.row_of_client[i])
            d = _ResultRow(id=c.client_id)
            if getattr(c, "local_training", True):
                d[CLIENT_UPDATE] = U[row, : self.engine.D]
            res.append(d)
        return res

    def _writeback_local_results(self, U: torch.Tensor):
        for i, d in enumerate(self.local_results):
            row = int(self.engine.plan.row_of_client[i])
            v = d.get(CLIENT_UPDATE, None)
            if v is None:
                continue
            tgt = U[row, : self.engine.D]
            if v.data_ptr() != tgt.data_ptr():
                tgt.copy_(v.to(tgt.device))

    # ------------------------------------------------------------------
    def training_step(self):
        from blades_amd.parallel.runtime import heartbeat

        heartbeat(f"round {self.engine.round}")  # stall watchdog of the launcher; a no-op without one
        T = self._timers
        with T("local_train"):
            U, loss_sum, loss_cnt = self.engine.train_round(self.server.theta)
        self.update_matrix = U
        self._inject_faults(U)
        if getattr(self.config, "guard_rows", False):
            with T("guard_rows"):
                self._guard_rows(U)
        with T("attack"):
            if self.adversary.uses_local_results:
                if self.ctx.distributed:
                    raise RuntimeError(f"{type(self.adversary).__name__} only implements the reference-style "
                                       "`local_results` interface, which needs world_size == 1.")
                self.local_results = self._build_local_results(U)
                self.adversary.on_local_round_end(self)
                self._writeback_local_results(U)
            else:
                self.adversary.on_local_round_end(self)
        self._counters[NUM_GLOBAL_STEPS] += 1
        global_vars = {"timestep": self._counters[NUM_GLOBAL_STEPS]}
        with T("aggregate_and_step"):
            server_return = self.server.step(U, global_vars, owned=True)
        self._check_replicas()
        pack = torch.stack([loss_sum, loss_cnt])
        if self._metrics_lag() and pack.is_cuda:
            stats = self._lagged_stats(pack)  # the previous round's numbers: the host never waits for this round
        else:
            stats = pack.tolist()  # the round's single host sync
        results = {"train_loss": stats[0] / stats[1] if stats[1] > 0 else float("nan")}
        if self._metrics_lag() and pack.is_cuda:
            results["train_loss_lag"] = 1
        results.update(server_return)
        results["timers"] = T.snapshot()
        results["timers"].update({k: v for k, v in self.engine.timers.items()})
        # bytes this rank moved over RCCL this round (update-matrix exchange / small all-reduces), SURVEY §5.5
        cs = self.ctx.stats
        results["comm_bytes"] = {"alltoall": int(self.engine.ctx.stats.get("alltoall_bytes", 0)),
                                 "allgather": int(self.engine.ctx.stats.get("allgather_bytes", 0)),
                                 "allreduce": int(cs.get("allreduce_bytes", 0)) - self._allreduce_bytes_seen}
        self._allreduce_bytes_seen = int(cs.get("allreduce_bytes", 0))
        info = getattr(self.server.aggregator, "last_info", None)
        if info and "num_selected" in info:
            results["num_selected_clients"] = int(info["num_selected"])
        return results

    # -- deferred loss read-back (BLADES_METRICS_LAG=1) ---------------------------------------------------------
    # Reading the round's loss is the one host sync of a round: the host then starts the next round's launches on an
    # idle GPU.  On 0.6-3 ms rounds (60 clients, MLP / CNN) that bubble is 0.15-0.2 ms per round (timeline:
    # profiles/r3_small_round_timeline_mlp.txt).  With the switch on, ``train()`` reports the loss of the PREVIOUS
    # round (``train_loss_lag: 1`` in the result row; NaN in the first row) from a pinned buffer whose copy finished a
    # round ago, the host runs at most one round ahead of the GPU, and ``flush_metrics()`` returns the last round's
    # loss.  Off by default: the reference's rows carry their own round's loss.
    def _metrics_lag(self) -> bool:
        import os

        return os.environ.get("BLADES_METRICS_LAG", "0") == "1"

    def _lagged_stats(self, pack):
        st = self.__dict__.setdefault("_lag", {"slots": None, "i": 0, "prev": None})
        if st["slots"] is None:
            st["slots"] = [(torch.zeros(2, dtype=pack.dtype).pin_memory(), torch.cuda.Event()) for _ in range(2)]
        buf, ev = st["slots"][st["i"]]
        st["i"] ^= 1
        buf.copy_(pack, non_blocking=True)
        ev.record()
        prev, st["prev"] = st["prev"], (buf, ev)
        if prev is None:
            return [float("nan"), 0.0]
        prev[1].synchronize()
        return prev[0].tolist()

    def flush_metrics(self):
        """Loss of the last trained round when losses are read back one round late (else None)."""
        st = self.__dict__.get("_lag")
        if not st or st["prev"] is None:
            return None
        buf, ev = st["prev"]
        ev.synchronize()
        s, c = buf.tolist()
        return s / c if c > 0 else float("nan")

    def _check_replicas(self):
        """"allgather" layout: every rank aggregates the full matrix on its own and steps its own copy of theta, so
        the replicas stay equal only while every adversary / aggregator path is bitwise identical on all ranks (host
        clustering, library solver choices).  Every ``BLADES_REPLICA_CHECK_EVERY`` rounds (default 16; 0 = never)
        the copies are compared (max - min of theta over ranks, one all-reduce of D floats each) and a fork raises
        instead of silently training different models."""
        if self.update_layout != "allgather" or not self.rt.distributed:
            return
        import os

        every = int(os.environ.get("BLADES_REPLICA_CHECK_EVERY", "16"))
        n = self._counters[NUM_GLOBAL_STEPS]
        if every <= 0 or n % every:
            return
        import torch.distributed as dist

        th = self.server.theta.detach()
        hi, lo = th.clone(), th.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        spread = float((hi - lo).abs().max())
        if not spread == 0.0:  # (NaN on some rank only also lands here)
            raise RuntimeError(f"replicated theta diverged across ranks after round {n}: max |difference| = {spread}")

    # -- fault injection / detection ---------------------------
    def _fault_events(self):
        ev = getattr(self.config, "fault_injection", None)
        import json
        import os

        env = os.environ.get("BLADES_INJECT_FAULT")
        evs = [] if ev is None else (list(ev) if isinstance(ev, (list, tuple)) else [ev])
        if env:
            e = json.loads(env)
            evs += e if isinstance(e, list) else [e]
        return evs

    def _inject_faults(self, U: torch.Tensor):
        """Test hook: corrupt a row of the update matrix or kill a rank at a given round (1-based, the round
        that is being computed).  Row indices are global rows of the column-sharded matrix, so every rank
        applies the corruption to its own column slice."""
        events = self._fault_events()
        if not events:
            return
        rnd = self._counters[NUM_GLOBAL_STEPS] + 1
        for e in events:
            if int(e.get("round", -1)) != rnd:
                continue
            kind = e.get("kind", "nan_row")
            if kind == "kill_rank":
                if self.rt.rank == int(e.get("rank", 0)):
                    import os

                    print(f"[fault-injection] rank {self.rt.rank} exits at round {rnd}", flush=True)
                    os._exit(int(e.get("code", 17)))
                continue
            row = int(e.get("row", 0))
            if kind == "nan_row":
                U[row].fill_(float("nan"))
            elif kind == "inf_row":
                U[row].fill_(float("inf"))
            elif kind == "scale_row":
                U[row].mul_(float(e.get("value", 1e30)))
            else:
                raise ValueError(f"unknown fault kind {kind!r}")

    def _guard_rows(self, U: torch.Tensor):
        """Zero every client row that holds a non-finite value.  The verdict comes from the row norms (one
        streaming pass over ``U`` + one ``[K]`` all-reduce — NaN/Inf propagate through the sum, so every rank
        agrees); only flagged rows are rewritten.  Costs one host sync per round, which is why it is opt-in.
        ``self.dropped_rows`` keeps the count of the last round."""
        from blades_amd import ops

        sq = ops.row_sqnorm(U)
        if self.ctx.distributed:
            self.ctx.all_reduce(sq)
        bad = (~torch.isfinite(sq)).nonzero().flatten()
        self.dropped_rows = int(bad.numel())
        if self.dropped_rows:
            U[bad] = 0.0

    def evaluate(self):
        from blades_amd.parallel.runtime import heartbeat

        # progress marks for the launcher's stall watchdog around the phases that are not a training round (the
        # engine marks every evaluation chunk as well): a long evaluation or checkpoint is work, not a hang
        heartbeat(f"evaluate {self.engine.round}")
        with self._timers("evaluate"):
            r = self.engine.evaluate(self.server.theta)
        heartbeat(f"evaluated {self.engine.round}")
        # reference returns ce_loss + acc_top_1 (fedavg.py:268-277); the other top-k accuracies of its Task
        # metric collection (task.py:104-121) come out of the same fused kernel for free
        return {k: v for k, v in r.items() if k == "ce_loss" or k.startswith("acc_top_")}

    # -- checkpoint -----------------------------------------------------
    def __getstate__(self) -> Dict:
        state = super().__getstate__()
        state.update({
            "server": self.server.state_dict(),
            "engine": self.engine.state_dict(),
            "adversary": self.adversary.state_dict(),
            "counters": dict(self._counters),
            "torch_rng": torch.get_rng_state(),
            "numpy_rng": np.random.get_state(),
            "world_size": self.rt.world_size,
        })
        return state

    def __setstate__(self, state: Dict):
        super().__setstate__(state)
        if "server" in state:
            self.server.load_state_dict(state["server"])
            self.engine.load_state_dict(state["engine"])  # keyed by client uid: any world size
            self.adversary.load_state_dict(state.get("adversary", {}))
            self._counters.update(state.get("counters", {}))
            torch.set_rng_state(state["torch_rng"])
            np.random.set_state(state["numpy_rng"])

    def save_checkpoint(self, checkpoint_dir: str) -> str:
        from blades_amd.parallel.runtime import heartbeat

        heartbeat(f"checkpoint {self.engine.round}")
        try:
            if self.rt.is_main:
                return super().save_checkpoint(checkpoint_dir)
            self.__getstate__()  # collectives inside state_dict must run on every rank
            return checkpoint_dir
        finally:
            heartbeat(f"checkpointed {self.engine.round}")
