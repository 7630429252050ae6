"""ClientEngine: trains all clients of a rank and produces the update matrix.

Replaces the reference's hot loop — one Ray RPC and one sequential
forward/backward per client (``blades/algorithms/fedavg/fedavg.py:203-224``,
``fllib/clients/client.py:46-74``) — with batched execution:

* **FedSGD path** (``num_batch_per_round == 1``, every shipped "fedsgd/fedavg"
  YAML): all clients start from the same weights, so one vectorised
  forward/backward over ``[Kc, B, …]`` with *shared* parameters yields the
  per-client gradients; the update row is ``-lr * g`` (momentum buffer and
  sign/clip coefficients fused in).  No per-client weight copies exist.
* **General path** (``num_batch_per_round > 1``): per-client weight replicas
  ``W[Kc, D]`` live in HBM for the duration of a chunk; every local step is a
  vectorised forward/backward with *per-client* parameters followed by the
  fused ``client_sgd_step_`` kernel; the update row is ``W_c - theta``.
* **Sequential path**: reference semantics through ``Client.train_one_round``
  for clients whose callbacks have Python-only behaviour.

BatchNorm layers use batch statistics (``track_running_stats=False``); under
vectorisation the statistics are per (client, channel), as they must be.

Afterwards the row shard ``[K_r, Dpad]`` is re-laid-out into the column shard
``[K, Dl]`` with one RCCL all-to-all (identity when ``world_size == 1``).
"""
import contextlib
import os
import time
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist
from torch.func import functional_call, grad_and_value, vmap

from blades_amd import ops
from blades_amd.constants import CLIENT_UPDATE
from blades_amd.parallel.context import ShardContext, padded_width
from .layers import DropContext, make_engine_model, swap_dropout
from .plan import RowPlan, make_plan
from .sampler import DeviceSampler


def store_hw(store):
    """(H, W) of the samples of a DeviceStore."""
    return int(store.x.shape[-2]), int(store.x.shape[-1])


def _autocast(device: torch.device, dtype: Optional[torch.dtype]):
    if dtype is None or dtype == torch.float32:
        return torch.autocast(device_type=device.type, enabled=False)
    return torch.autocast(device_type=device.type, dtype=dtype)


# "fp32": fp32 activations; the hand-written convolution kernels split them exactly into 3 bf16 planes (6 MFMA
# products, fp32 grade).  "bf16x3": fp32 activations, 2 planes / 3 products (~2^-17).  "bf16": bf16 activations.
_DTYPES = {None: None, "fp32": None, "float32": None, "bf16x3": None, "bf16": torch.bfloat16,
           "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16}


class ClientEngine:
    def __init__(self, *, task, dataset, clients: list, runtime, client_lr: float, client_momentum: float,
                 num_batch_per_round: int, batch_size: int, seed: int = 0, clients_per_chunk="auto",
                 amp_dtype="auto", execution_mode: str = "auto", num_classes: Optional[int] = None,
                 eval_clients_per_chunk: Optional[int] = None, update_layout: str = "alltoall"):
        self.task = task
        self.drop_ctx = DropContext(seed, runtime.device)
        self.model = swap_dropout(make_engine_model(task.model), self.drop_ctx)
        self.dataset = dataset
        self.clients = clients
        self.rt = runtime
        self.device = runtime.device
        self.lr = float(client_lr)
        self.momentum = float(client_momentum)
        self.nb = int(num_batch_per_round)
        self.B = int(batch_size)
        self.seed = int(seed)
        if clients_per_chunk in (None, "auto"):
            # MI355X default: 200 clients x batch 64 per launch is where the per-client kernels saturate the
            # chip on ResNet-18 (measured: 64 -> 100 -> 200 clients/chunk each gain ~3 %); larger batches take
            # proportionally fewer clients so that activation memory stays put.  CPU runs keep chunks small.
            on_gpu = self.device.type == "cuda"
            clients_per_chunk = max(8, min(200, (200 * 64) // max(int(batch_size), 1))) if on_gpu else 64
            # fp32 planes on a large convolutional model: the MFMA-bound kernels are full at 32-40 x 64 samples and
            # the smaller working set is worth 5 % (profiles/r2_chunk_stream_sweep_fp32.txt, ..._v6.txt)
            n_par = sum(p.numel() for p in task.model.parameters())
            if on_gpu and amp_dtype in ("auto", "fp32", "bf16x3", None) and n_par > 8_000_000:
                clients_per_chunk = max(8, (32 * 64) // max(int(batch_size), 1))
        self.chunk = max(1, int(clients_per_chunk))
        self.eval_chunk = eval_clients_per_chunk or self.chunk
        self.num_classes = num_classes
        self.mode = execution_mode
        if amp_dtype == "auto":
            amp_dtype = "fp32"  # the reference trains in fp32 (fllib/tasks/task.py:170-186); bf16 is opt-in
        self.conv_planes = 2 if amp_dtype == "bf16x3" else 3
        self.amp_dtype = _DTYPES[amp_dtype] if not isinstance(amp_dtype, torch.dtype) else amp_dtype
        self.compute_dtype = ("bf16x3" if self.conv_planes == 2 else
                              {None: "fp32", torch.bfloat16: "bf16", torch.float16: "fp16"}[self.amp_dtype])
        self.timers: Dict[str, float] = {}
        self.round = 0
        if self.device.type == "cuda":
            # Only the library fallbacks use this (BLADES_CONV_IMPL=miopen, the grouped/vmap path of models the fused
            # executor does not cover): the default training paths run the engine's own convolution kernels.
            torch.backends.cudnn.benchmark = True

        # ---- parameter layout (named_parameters order == reference flattening)
        # trainable parameters only: frozen ones are constants of the model, same enumeration as the server's
        from blades_amd.utils.torch_utils import named_trainable_parameters

        self.names = [n for n, _ in named_trainable_parameters(task.model)]
        self.shapes = [tuple(p.shape) for _, p in named_trainable_parameters(task.model)]
        self.numels = [int(np.prod(s)) if len(s) else 1 for s in self.shapes]
        self.offsets = np.concatenate([[0], np.cumsum(self.numels)]).astype(np.int64)
        self.D = int(self.offsets[-1])
        G = runtime.world_size
        self.ctx = ShardContext(d_total=self.D, world_size=G, rank=runtime.rank, force=runtime.force_collectives)
        self.Dl = self.ctx.d_local
        self.Dpad = self.Dl * G

        # ---- plan rows, collect per-client fused specs
        self.K = len(clients)
        trains = [bool(getattr(c, "local_training", True)) for c in clients]
        self.plan: RowPlan = make_plan(trains, G)
        self.local_ids = self.plan.local_clients(runtime.rank)
        self.Kloc = len(self.local_ids)
        self.refresh_client_specs()

        # ---- resident data
        ids = [clients[i].client_id for i in self.local_ids]
        self._train_rows = np.array([r for r, i in enumerate(self.local_ids) if self._trains[r]], dtype=np.int64)
        if self.ctx.distributed:
            # at least two chunks per rank: the all-to-all piece of the first travels while the second trains
            # (all ranks must agree on the piece count, so the rule uses the largest per-rank training load)
            most = max(sum(1 for c in rc if trains[c]) for rc in self.plan.rank_clients)
            if most >= 2:
                self.chunk = max(1, min(self.chunk, -(-most // 2)))
        train_ids = [ids[r] for r in self._train_rows]
        self.train_store = dataset.device_store(train_ids, self.device, "train")
        self._train_uids = np.array([self.local_ids[r] for r in self._train_rows], dtype=np.int64)
        self.sampler = DeviceSampler(self.train_store.lengths, self.train_store.offsets, self.B, self.device,
                                     seed=self.seed * 7919 + 1, client_uids=self._train_uids)
        self.test_store = dataset.device_store(ids, self.device, "test")

        # ---- persistent buffers
        # Row stride is always a multiple of COL_ALIGN floats, also on one GPU where the logical width is the
        # unpadded D: rows that start float4-aligned are what the streaming kernels' vector paths need
        # (Mean over 1000 x 11.2 M: 7.5 ms with aligned rows, 31 ms without).
        ld = padded_width(self.Dpad, 1)
        self.layout = update_layout if self.ctx.distributed else "alltoall"
        if self.layout == "allgather":
            # the full [K, D] matrix lives on every rank; this rank's clients write straight into their rows of
            # it, peers' rows arrive there: no staging copy on either side
            self._U_store = torch.zeros(self.K, ld, dtype=torch.float32, device=self.device)
            r0 = int(np.sum(self.plan.rows_per_rank()[: runtime.rank]))
            self.U_row = self._U_store[r0: r0 + self.Kloc, : self.Dpad]
            self.U_col = self._U_store[:, : self.D]
        else:
            self._U_store = torch.zeros(self.Kloc, ld, dtype=torch.float32, device=self.device)
            self.U_row = self._U_store[:, : self.Dpad]
            self.U_col = self.U_row if not self.ctx.distributed else torch.zeros(
                self.K, self.Dl, dtype=torch.float32, device=self.device)
        # Client chunks alternate over HIP streams: +3 % on the headline round with two streams, another 1 % with
        # three (four are slower; profiles/r2_chunk_stream_sweep_v6.txt).  What the streams buy is closed launch
        # gaps and tails — a kernel's last, partly filled wave of workgroups runs beside the next chunk's kernel —
        # not bandwidth-bound kernels hidden under matrix-core ones: a convolution launch fills every CU, and the
        # serial kernel sum of a round equals its wall clock (profiles/r3_stream_overlap_probe_bn_under_conv.log).
        # Multi-GPU runs do the same; their exchange pieces are issued one chunk behind (train_round).
        default_streams = "3" if self.device.type == "cuda" else "1"
        self.n_streams = max(1, int(os.environ.get("BLADES_CHUNK_STREAMS", default_streams)))
        # HIP-graph capture of a FedSGD chunk (see _graphed_chunk): "auto" turns it on for launch-bound rounds
        self.hip_graphs = os.environ.get("BLADES_HIP_GRAPH", "auto").lower()
        self._chunk_graphs = {}
        self._graph_failed = False
        self.graph_replays = 0
        self._streams = None
        self.M = None  # per-client momentum buffers [Kloc_train, D], lazily
        self._mom_init = False
        from . import fused

        self.fused = None
        if self.mode in ("auto", "batched", "fused") and fused.supports(task.model):
            self.fused = fused.FusedFedSGD(task, self.names, self.offsets, self.numels, self.amp_dtype, self.device,
                                           conv_planes=self.conv_planes)
            self.fused.template = swap_dropout(self.fused.template, self.drop_ctx)
        # several local steps (per-client weights), GPU: the fused executor itself, with the hand-written kernels
        # picking each client's weight set (engine/fused.py::run_per_client); first NotImplementedError (a layer
        # without a per-client kernel) drops to the grouped / vmap executors below for good
        self.fused_pc = (self.fused is not None and self.nb > 1 and self.device.type == "cuda"
                         and os.environ.get("BLADES_DISABLE_FUSED_PC") != "1")
        self.fused_pc_error = None
        # several local steps: per-client weights -> grouped wide-network executor, if the model passes its
        # layout self-check (otherwise the torch.vmap path).  Where the fused executor runs with per-client weights
        # the grouped one is only its fallback: it is then built (and self-checked, which runs the library
        # convolutions once) when a layer first declines, not at setup.
        self.grouped = None
        self._grouped_ready = False
        self._grouped_args = (task, dataset)
        if not self.fused_pc:
            self._ensure_grouped()

    def _ensure_grouped(self):
        if self._grouped_ready:
            return
        self._grouped_ready = True
        task, dataset = self._grouped_args
        if self.nb > 1 and self.mode in ("auto", "batched", "fused") and self.Kloc > 0:
            from . import grouped

            if grouped.supports(task.model):
                ge = grouped.GroupedLocalSGD(task, self.names, self.shapes, self.offsets[:-1], self.numels,
                                             self.amp_dtype, self.device)
                ge.template = swap_dropout(ge.template, self.drop_ctx)

                def ref(W, X, Y):
                    was = self.model.training
                    self.model.eval()
                    try:
                        amp, self.amp_dtype = self.amp_dtype, None
                        g, _ = self._grads(self._params_from_flat(W, batched=True), X, Y, per_client_params=True)
                        self.amp_dtype = amp
                        R = torch.empty_like(W)
                        self._flatten_into(g, R)
                        return R
                    finally:
                        self.model.train(was)

                ncls = int(getattr(dataset, "num_classes", 10) or 10)
                if ge.self_check(ref, self.D, tuple(self.train_store.x.shape[1:]), ncls):
                    self.grouped = ge
                self.grouped_check = getattr(ge, "check_error", None)  # diagnostics: error or exception text

    # ------------------------------------------------------------------
    def refresh_client_specs(self):
        """(Re)reads each local client's callbacks -> fused per-row settings."""
        specs, trains = [], []
        for i in self.local_ids:
            c = self.clients[i]
            trains.append(bool(getattr(c, "local_training", True)))
            cbs = c.active_callbacks() if hasattr(c, "active_callbacks") else getattr(c, "callbacks", None)
            spec = cbs.batched_spec() if cbs is not None and hasattr(cbs, "batched_spec") else {}
            if self.mode == "sequential":
                spec = None
            specs.append(spec)
        self._specs, self._trains = specs, trains
        self._row_vec_cache = {}  # per-chunk device vectors derive from the specs
        self._spec_version = getattr(self, "_spec_version", 0) + 1
        self._malicious = [bool(getattr(self.clients[i], "is_malicious", False)) for i in self.local_ids]

    def _row_vectors(self, rows: np.ndarray):
        """Per-row fused settings for local rows ``rows`` (all batched)."""
        key = rows.tobytes()
        hit = self._row_vec_cache.get(key)
        if hit is not None:
            gscale, flip, gclip, self._any_flip, self._any_clip = hit
            return gscale, flip, gclip
        sp = [self._specs[r] for r in rows]
        flips = [s.get("label_flip", 0) for s in sp]
        clips = [s.get("grad_clip", float("inf")) for s in sp]
        gscale = torch.tensor([s.get("grad_scale", 1.0) for s in sp], device=self.device)
        flip = torch.tensor(flips, device=self.device)
        gclip = torch.tensor(clips, device=self.device)
        # "is there anything to do" is decided on the host lists: asking the device tensors would be a
        # stream-draining sync at the start of every chunk
        self._any_flip = any(f > 0 for f in flips)
        self._any_clip = any(np.isfinite(c) for c in clips)
        self._row_vec_cache[key] = (gscale, flip, gclip, self._any_flip, self._any_clip)
        return gscale, flip, gclip

    def _chunk_consts(self, rows: np.ndarray):
        """(row index tensor, benign flags) of a chunk on the device — cached like the row vectors."""
        key = (b"c", rows.tobytes())
        hit = self._row_vec_cache.get(key)
        if hit is None:
            hit = self._row_vec_cache[key] = (
                torch.as_tensor(rows, device=self.device),
                torch.tensor([not self._malicious[r] for r in rows], device=self.device))
        return hit

    # ------------------------------------------------------------------
    def _params_from_flat(self, flat: torch.Tensor, batched: bool = False) -> Dict[str, torch.Tensor]:
        out = {}
        for n, s, o, m in zip(self.names, self.shapes, self.offsets[:-1], self.numels):
            v = flat[..., int(o): int(o) + m]
            out[n] = v.reshape(*flat.shape[:-1], *s) if batched else v.reshape(s)
        return out

    def _client_loss(self, params, x, y, uid=None):
        if uid is not None:
            self.drop_ctx.set(uid, "single")  # under vmap: a 0-dim batched tensor, one client per slice

        def m(inp):
            return functional_call(self.model, params, (inp,))

        with _autocast(self.device, self.amp_dtype):
            return self.task.loss(m, x, y)

    def _grads(self, params, X, Y, per_client_params: bool, uids: Optional[torch.Tensor] = None):
        """Vectorised per-client (grads dict [Kc,…], losses [Kc])."""
        if uids is None:  # numerical self-checks: no dropout keys
            fn = vmap(grad_and_value(self._client_loss), in_dims=(0 if per_client_params else None, 0, 0),
                      randomness="different")
            return fn(params, X, Y)
        fn = vmap(grad_and_value(self._client_loss), in_dims=(0 if per_client_params else None, 0, 0, 0),
                  randomness="different")
        return fn(params, X, Y, uids)

    def _chunk_uids(self, srows: np.ndarray) -> torch.Tensor:
        key = (b"u", srows.tobytes())
        hit = self._row_vec_cache.get(key)
        if hit is None:
            hit = self._row_vec_cache[key] = torch.as_tensor(self._train_uids[srows], dtype=torch.int64,
                                                             device=self.device)
        return hit

    def _drop_tag(self) -> int:
        return self.round * self.nb + getattr(self, "_local_step", 0) + 1

    def _fetch_batch(self, srows: np.ndarray, flip: torch.Tensor):
        """One training batch per client of sampler rows ``srows`` -> X[Kc,B,…], Y[Kc,B]."""
        idx = self.sampler.next_indices(srows)
        Kc = len(srows)
        X = self.train_store.x[idx.reshape(-1)].reshape(Kc, self.B, *self.train_store.x.shape[1:])
        Y = self.train_store.y[idx.reshape(-1)].reshape(Kc, self.B)
        tf = self.dataset.train_transform
        if tf is not None:
            if hasattr(tf, "draw"):
                X = tf.apply(X, tf.draw(Kc, None, X.shape[-2], X.shape[-1], uniforms=self._aug_uniforms(srows, tf)))
            else:
                X = torch.stack([tf(X[c]) for c in range(Kc)])
        if self._any_flip:
            Y = torch.where(flip.unsqueeze(1) > 0, flip.unsqueeze(1) - 1 - Y, Y)
        return X, Y

    def _aug_uniforms(self, srows: np.ndarray, tf) -> torch.Tensor:
        """[Kc, N_UNIFORMS] uniforms keyed by (seed, client uid, augmentation call counter)."""
        from blades_amd.ops.reference import _philox4x32

        # keyed by (seed, client uid, round, local step): independent of chunking and of the number of ranks
        batch_no = self.round * self.nb + getattr(self, "_local_step", 0) + 1
        n = tf.N_UNIFORMS
        uid = torch.as_tensor(self._train_uids[srows], dtype=torch.int64).unsqueeze(1).expand(-1, n)
        j = torch.arange(n, dtype=torch.int64).unsqueeze(0).expand(len(srows), -1)
        call = torch.full_like(j, batch_no & 0xFFFFFFFF)
        sd = self.seed * 104729 + 3
        w = _philox4x32(j, uid & 0xFFFFFFFF, call, torch.full_like(j, 11), sd & 0xFFFFFFFF, (sd >> 32) & 0xFFFFFFFF)
        return (w[0] >> 8).float() * (1.0 / 16777216.0)

    @staticmethod
    def _clip_coef(sqnorm: torch.Tensor, thr: torch.Tensor) -> torch.Tensor:
        return (thr / (sqnorm.sqrt() + 1e-6)).clamp(max=1.0)

    def _flatten_into(self, grads: Dict[str, torch.Tensor], out: torch.Tensor):
        Kc = out.shape[0]
        for n, o, m in zip(self.names, self.offsets[:-1], self.numels):
            out[:, int(o): int(o) + m].copy_(grads[n].reshape(Kc, m))

    # ------------------------------------------------------------------
    # -- HIP-graph path ---------------------------------------------------------------------------------------
    # Small rounds (the reference's 60-client experiments on an MLP / small CNN / ResNet-10) are bound by
    # kernel-launch latency, not by the GPU: a chunk is a few hundred launches of microsecond kernels.  Such a
    # chunk is captured once — batch gather from the device-resident sampler, augmentation, the fused
    # forward/backward with its per-client gradient kernels writing into the update-matrix rows — and replayed
    # as one hipGraphLaunch per round.  Everything the graph reads lives at fixed addresses (theta, the sample
    # store, permutation table and cursors, U rows); the only per-round host input is the augmentation table.
    def _graphs_wanted(self) -> bool:
        if self.hip_graphs in ("0", "off", "false") or self._graph_failed:
            return False
        if self.device.type != "cuda" or self.fused is None or self.nb != 1 or self.momentum != 0.0:
            return False
        if self.hip_graphs in ("1", "on", "true"):
            return True
        # auto: per-round work small enough that launches dominate (ResNet-10 x 60 clients yes, the 1000-client
        # ResNet-18 headline no — there the GPU is saturated and a graph would only pin activation memory)
        return len(self._train_rows) * self.B * self.D <= 5e10

    def _begin_round_planes(self):
        """``fused.begin_round()`` — one weight-plane launch per convolution / Linear layer — as a HIP graph where the
        round itself is graphed: on the MLP these six eager launches with their gaps were 0.2 ms of a 0.65 ms round
        (timeline in profiles/r3_small_round_timeline_mlp_dnc.txt).  The graph reads the weights and writes the planes
        by address: it is re-captured when a weight tensor has moved."""
        if not self._graphs_wanted() or os.environ.get("BLADES_ROUND_PLANES", "1") == "0":
            self.fused.begin_round()
            return
        ent = self.__dict__.setdefault("_planes_graph", {"visits": 0, "graph": None, "key": None})
        key = tuple(p.data_ptr() for p in self.fused.template.parameters()) + (self.fused.sink.npl,)
        if ent["graph"] is not None and ent["key"] == key:
            ent["graph"].replay()
            self.fused.sink.planes_valid = True
            return
        ent["visits"] += 1
        self.fused.begin_round()  # eager: allocates the plane buffers on the first visits
        if ent["visits"] <= 2 or ent.get("failed"):
            return
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                self.fused.begin_round()
        except Exception as e:  # noqa: BLE001 — not capturable: stay eager
            ent["failed"] = repr(e)
            torch.cuda.synchronize()
            self.fused.begin_round()
            return
        ent.update(graph=g, key=key)  # (capture does not execute: the eager call above produced this round's planes)

    def _graphed_chunk(self, rows, srows):
        """Replays (capturing on the third visit) the HIP graph of this chunk; ``None`` = not graphable."""
        tf = self.dataset.train_transform
        aug = tf is not None and hasattr(tf, "draw")
        if tf is not None and not aug:
            return None
        if int(rows[-1] - rows[0]) != len(rows) - 1:
            return None
        gscale, flip, gclip = self._row_vectors(rows)
        if self._any_clip:
            return None
        key = (rows.tobytes(), srows.tobytes(), self._spec_version)
        ent = self._chunk_graphs.get(key)
        if ent is None:
            if len(self._chunk_graphs) > 64:
                self._chunk_graphs.clear()
            ent = self._chunk_graphs[key] = {"visits": 0, "graph": None}
        ent["visits"] += 1
        if ent["graph"] is None and ent["visits"] <= 2:
            return None  # eager first: wgrad plan tables (host->device copies) and library kernel searches happen
            # outside capture; the split-K scratch is a per-graph temporary (ops/_hip.py::_scratch)
        Kc = len(rows)
        # the chunk's own dropout tag (captured by address, filled before every replay): engine/layers.py DropContext.bind
        self.drop_ctx.bind((b"f",) + key)
        ent["tag"] = self.drop_ctx.tag
        if ent["graph"] is None:
            any_flip = self._any_flip
            dev_rows = self.sampler.rows_on_device(srows)
            block = self.U_row[int(rows[0]): int(rows[-1]) + 1]
            scale = (-self.lr * gscale).float()
            store = self.train_store
            params = torch.zeros(Kc, 9, device=self.device) if aug else None  # transforms.N_PARAMS columns

            uids_t = self._chunk_uids(srows)

            def body():
                self.drop_ctx.set(uids_t, "major")
                idx = self.sampler.gather_indices(dev_rows).reshape(-1)
                X = store.x[idx].reshape(Kc, self.B, *store.x.shape[1:])
                Y = store.y[idx].reshape(Kc, self.B)
                if aug:
                    X = tf.apply(X, params)
                if any_flip:
                    Y = torch.where(flip.unsqueeze(1) > 0, flip.unsqueeze(1) - 1 - Y, Y)
                losses = self.fused.run(X, Y, block, scale)
                # the round's loss bookkeeping for this chunk rides in the graph (four eager launches otherwise)
                return losses, (losses.double() * benign).sum(), benign.sum()

            benign = self._chunk_consts(rows)[1]
            g = torch.cuda.CUDAGraph()
            try:
                # Every chunk graph has a memory pool of its own (torch's default).  Round 1 saw sporadic
                # non-finite rows when chunk graphs shared a pool.  Round-2 audit (tools/probes/poison_probe.py,
                # multi_graph_probe.py; profiles/r2_graph_pool_audit.txt): with every kernel workspace / output
                # pre-filled with NaN, eager, private-pool and shared-pool rounds are finite and bitwise equal
                # (2 dtypes x 12 trials x 8 rounds), i.e. no kernel reads memory nothing wrote; what was left
                # shared across graphs was the module-global split-K scratch slab, first allocated inside the
                # first graph's capture (so owned by that graph's pool) and then written by every later graph.
                # It is a per-graph temporary now.  Private pools stay: they cost nothing for these small rounds.
                pool = None
                if os.environ.get("BLADES_GRAPH_SHARED_POOL") == "1":  # diagnostics (tools/probes/poison_probe.py)
                    pool = self.__dict__.setdefault("_graph_pool", torch.cuda.graph_pool_handle())
                with torch.cuda.graph(g, pool=pool):
                    losses, loss_part, cnt_part = body()
            except Exception as e:  # noqa: BLE001 — anything not capturable: stay eager for good
                self._graph_failed = True
                self._graph_error = repr(e)
                self._chunk_graphs.clear()
                torch.cuda.synchronize()
                return None
            # everything the graph reads by address must outlive it: keep the per-chunk tensors with the graph
            ent.update(graph=g, losses=losses, params=params, keep=(scale, flip, gscale, block, dev_rows, benign),
                       loss_pair=(loss_part, cnt_part))
        self.sampler.prepare(srows)
        self.drop_ctx.set_tag(self._drop_tag())
        if aug:
            table = tf.draw(Kc, None, store_hw(self.train_store)[0], store_hw(self.train_store)[1],
                            uniforms=self._aug_uniforms(srows, tf))
            tf.check(table, *store_hw(self.train_store))
            ent["params"].copy_(table)
        ent["graph"].replay()
        self.graph_replays += 1
        self.sampler.finish(srows)
        self._graph_loss_pair = ent["loss_pair"]
        return ent["losses"]

    def _train_chunk_fedsgd(self, theta, rows, srows):
        self._local_step = 0
        if self._graphs_wanted():
            losses = self._graphed_chunk(rows, srows)
            if losses is not None:
                return losses
        gscale, flip, gclip = self._row_vectors(rows)
        X, Y = self._fetch_batch(srows, flip)
        uids_t = self._chunk_uids(srows)
        self.drop_ctx.bind((b"e", rows.tobytes()))
        self.drop_ctx.set_tag(self._drop_tag())
        self.drop_ctx.set(uids_t, "major")
        rows_t = self._chunk_consts(rows)[0]
        contiguous = int(rows[-1] - rows[0]) == len(rows) - 1
        plain = self.momentum == 0.0 and not self._any_clip
        if self.fused is not None:
            # per-client gradients land in the rows of U directly, pre-scaled
            block = self.U_row[int(rows[0]): int(rows[-1]) + 1] if contiguous else torch.empty(
                len(rows), self.D, device=self.device)
            scale = (-self.lr * gscale) if plain else torch.ones_like(gscale)
            losses = self.fused.run(X, Y, block, scale.float())
            if plain:
                if not contiguous:
                    self.U_row[rows_t, : self.D] = block
                return losses
            G = block[:, : self.D]
        else:
            grads, losses = self._grads(self._params_from_flat(theta), X, Y, per_client_params=False, uids=uids_t)
            G = torch.empty(len(rows), self.D, device=self.device)
            self._flatten_into(grads, G)
            del grads
            contiguous = False
        if self._any_clip:
            gscale = gscale * self._clip_coef(ops.row_sqnorm(G), gclip)
        if self.momentum != 0.0:
            if self.M is None:
                self.M = torch.zeros(len(self._train_rows), self.D, device=self.device)
            srows_t = torch.as_tensor(srows, device=self.device)
            Mv = self.M[srows_t]
            W = torch.zeros(len(rows), self.D, device=self.device)  # W - 0 = update after one step from zero
            ops.client_sgd_step_(W, G, Mv, self.lr, self.momentum, not self._mom_init, gscale)
            self.M[srows_t] = Mv
            self.U_row[rows_t, : self.D] = W
        else:
            ops.row_scale_(G, -self.lr * gscale)
            if not (self.fused is not None and contiguous):
                self.U_row[rows_t, : self.D] = G
        return losses

    def _general_graphs_wanted(self) -> bool:
        if self.hip_graphs in ("0", "off", "false") or self._graph_failed:
            return False
        if self.device.type != "cuda" or not self.fused_pc or self.nb <= 1:
            return False
        if self.hip_graphs in ("1", "on", "true"):
            return True
        return len(self._train_rows) * self.B * self.D <= 5e10  # launch-bound rounds only, as for FedSGD chunks

    def _graphed_local_steps(self, theta, rows, srows):
        """Several local steps per round on launch-bound models: ONE local step of the chunk (batch gather,
        augmentation, per-client-weight forward / backward on the fused executor, client SGD step) is captured as a
        HIP graph on persistent per-chunk weight / gradient / momentum buffers and replayed ``nb`` times per round;
        the host only advances the sampler bookkeeping, the dropout tag and the augmentation table in between.
        Returns the mean losses, or ``None`` when the chunk is not graphable (the eager loop runs instead)."""
        tf = self.dataset.train_transform
        aug = tf is not None and hasattr(tf, "draw")
        if tf is not None and not aug:
            return None
        gscale0, flip, gclip = self._row_vectors(rows)
        if self._any_clip:
            return None
        key = (b"g", rows.tobytes(), srows.tobytes(), self._spec_version)
        ent = self._chunk_graphs.get(key)
        if ent is None:
            if len(self._chunk_graphs) > 64:
                self._chunk_graphs.clear()
            ent = self._chunk_graphs[key] = {"visits": 0, "graph": None}
        ent["visits"] += 1
        if ent["graph"] is None and ent["visits"] <= 2:
            return None  # eager first (plan tables, first-use allocations, momentum initialisation)
        Kc = len(rows)
        srows_t = torch.as_tensor(srows, device=self.device)
        rows_t = torch.as_tensor(rows, device=self.device)
        self.drop_ctx.bind(key)  # the chunk's own dropout tag: captured by address, filled before every replay
        ent["tag"] = self.drop_ctx.tag
        if ent["graph"] is None:
            any_flip = self._any_flip
            dev_rows = self.sampler.rows_on_device(srows)
            store = self.train_store
            W = torch.empty(Kc, self.D, device=self.device)
            inplace = self._inplace_client_step()
            G = None if inplace else torch.empty_like(W)
            step_coef = (gscale0 * (-self.lr)).contiguous() if inplace else None
            Mv = torch.zeros_like(W) if self.momentum != 0.0 else W
            loss_acc = torch.zeros(Kc, device=self.device)
            params = torch.zeros(Kc, 9, device=self.device) if aug else None
            uids_t = self._chunk_uids(srows)
            W.copy_(theta.unsqueeze(0).expand(Kc, -1))  # valid weights while capturing

            def body():
                self.drop_ctx.set(uids_t, "major")
                idx = self.sampler.gather_indices(dev_rows).reshape(-1)
                X = store.x[idx].reshape(Kc, self.B, *store.x.shape[1:])
                Y = store.y[idx].reshape(Kc, self.B)
                if aug:
                    X = tf.apply(X, params)
                if any_flip:
                    Y = torch.where(flip.unsqueeze(1) > 0, flip.unsqueeze(1) - 1 - Y, Y)
                if inplace:
                    losses = self.fused.run_per_client(W, X, Y, None, step_coef=step_coef)
                else:
                    losses = self.fused.run_per_client(W, X, Y, G)
                    ops.client_sgd_step_(W, G, Mv, self.lr, self.momentum, False, gscale0)
                loss_acc.add_(losses)

            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    body()
            except Exception as e:  # noqa: BLE001 — not capturable: eager for good
                self._graph_failed = True
                self._graph_error = repr(e)
                self._chunk_graphs.clear()
                torch.cuda.synchronize()
                return None
            ent.update(graph=g, W=W, G=G, Mv=Mv, loss=loss_acc, params=params,
                       keep=(flip, gscale0, dev_rows, uids_t, step_coef))
        W, Mv, loss_acc = ent["W"], ent["Mv"], ent["loss"]
        W.copy_(theta.unsqueeze(0).expand(Kc, -1))
        loss_acc.zero_()
        if self.momentum != 0.0:
            if self.M is None:
                self.M = torch.zeros(len(self._train_rows), self.D, device=self.device)
            Mv.copy_(self.M[srows_t])
        hw = store_hw(self.train_store)
        for step in range(self.nb):
            self._local_step = step
            self.sampler.prepare(srows)
            self.drop_ctx.set_tag(self._drop_tag())
            if aug:
                table = tf.draw(Kc, None, hw[0], hw[1], uniforms=self._aug_uniforms(srows, tf))
                tf.check(table, *hw)
                ent["params"].copy_(table)
            ent["graph"].replay()
            self.graph_replays += 1
            self.sampler.finish(srows)
        if self.momentum != 0.0:
            self.M[srows_t] = Mv
        self.U_row[rows_t, : self.D] = W - theta.unsqueeze(0)
        return loss_acc / self.nb

    def _inplace_client_step(self) -> bool:
        """Plain client SGD (no momentum, no gradient clipping) on the fused per-client executor of a model without
        normalisation layers: the gradient kernels add ``-lr * gscale * dW`` onto the weight rows themselves
        (engine/fused.py::run_per_client(step_coef=...)); taken once a regular step has gone through."""
        return (self.fused_pc and self.fused is not None and self.momentum == 0.0 and not self._any_clip
                and self.fused.inplace_step_ok and self.fused.pc_steps_ok > 0)

    def _train_chunk_general(self, theta, rows, srows):
        if self._general_graphs_wanted() and self._mom_init:
            losses = self._graphed_local_steps(theta, rows, srows)
            if losses is not None:
                return losses
        gscale0, flip, gclip = self._row_vectors(rows)
        Kc = len(rows)
        W = theta.unsqueeze(0).repeat(Kc, 1)  # a real copy even for Kc == 1 (expand().contiguous() would alias theta)
        inplace = self._inplace_client_step()
        G = None if inplace else torch.empty_like(W)
        step_coef = (gscale0 * (-self.lr)).contiguous() if inplace else None
        srows_t = torch.as_tensor(srows, device=self.device)
        if self.momentum != 0.0:
            if self.M is None:
                self.M = torch.zeros(len(self._train_rows), self.D, device=self.device)
            Mv = self.M[srows_t].contiguous()
        else:
            Mv = W  # unused
        loss_acc = torch.zeros(Kc, device=self.device)
        for step in range(self.nb):
            self._local_step = step
            X, Y = self._fetch_batch(srows, flip)
            uids_t = self._chunk_uids(srows)
            self.drop_ctx.bind((b"e", rows.tobytes()))
            self.drop_ctx.set_tag(self._drop_tag())
            losses = None
            if self.fused_pc:
                try:
                    self.drop_ctx.set(uids_t, "major")
                    if inplace:
                        losses = self.fused.run_per_client(W, X, Y, None, step_coef=step_coef)
                        loss_acc += losses
                        continue  # the gradient kernels have taken the SGD step
                    losses = self.fused.run_per_client(W, X, Y, G)
                except NotImplementedError as e:
                    self.fused_pc, self.fused_pc_error = False, str(e)
                    self._ensure_grouped()
                    if G is None:
                        inplace, G = False, torch.empty_like(W)
            if losses is not None:
                pass
            elif self.grouped is not None:
                self.drop_ctx.set(uids_t, "inner")
                losses = self.grouped.run(W, X, Y, G)
            else:
                grads, losses = self._grads(self._params_from_flat(W, batched=True), X, Y, per_client_params=True,
                                            uids=uids_t)
                self._flatten_into(grads, G)
                del grads
            gs = gscale0
            if self._any_clip:
                gs = gs * self._clip_coef(ops.row_sqnorm(G), gclip)
            first = (not self._mom_init) and step == 0
            ops.client_sgd_step_(W, G, Mv, self.lr, self.momentum, first, gs)
            loss_acc += losses
        if self.momentum != 0.0:
            self.M[srows_t] = Mv
        rows_t = torch.as_tensor(rows, device=self.device)
        self.U_row[rows_t, : self.D] = W - theta.unsqueeze(0)
        return loss_acc / self.nb

    def _train_sequential(self, theta, r):
        """Reference-semantics round for local row ``r`` (Python callbacks)."""
        from blades_amd.parallel.worker import current_worker

        client = self.clients[self.local_ids[r]]
        w = current_worker()
        with torch.no_grad():
            from blades_amd.utils.torch_utils import vector_to_parameters

            vector_to_parameters(theta.to(next(w.task.model.parameters()).device), w.task.model.parameters())
        res = client.train_one_round(self.dataset.get_train_loader(client.client_id))
        upd = res.get(CLIENT_UPDATE)
        if upd is not None:
            self.U_row[r, : self.D] = upd.to(self.device)
        return res.get("avg_loss", float("nan"))

    # ------------------------------------------------------------------
    def train_round(self, theta: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Returns (U_col [K, Dl], sum of benign losses, number of benign clients) —
        the last two already reduced over ranks."""
        t0 = time.perf_counter()
        dev = self.device
        theta = theta.detach()[: self.D]
        self.model.train()
        loss_sum = torch.zeros((), device=dev, dtype=torch.float64)
        loss_cnt = torch.zeros((), device=dev, dtype=torch.float64)
        batched = [k for k, r in enumerate(self._train_rows) if self._specs[r] is not None]
        seq = [k for k, r in enumerate(self._train_rows) if self._specs[r] is None]
        nontrain = [r for r in range(self.Kloc) if not self._trains[r]]
        if nontrain:
            if nontrain[-1] - nontrain[0] == len(nontrain) - 1:
                self.U_row[nontrain[0]: nontrain[-1] + 1].zero_()  # one memset instead of an index_put
            else:
                key = (b"n", self._spec_version)
                nt = self._row_vec_cache.get(key)
                if nt is None:
                    nt = self._row_vec_cache[key] = torch.as_tensor(nontrain, device=dev)
                self.U_row[nt] = 0
        # rows whose content is final only at the end of the round (sequential clients, round-end hooks):
        # with any of those the exchange cannot start early
        late = bool(seq) or self._has_round_end_hooks()
        self._exchange_begin()
        if self.fused is not None and self.nb == 1:
            # theta does not change until the server step: the convolution weights are split into their operand
            # planes once, here, instead of once per chunk and launch
            self._begin_round_planes()
        # Chunks alternate between `n_streams` HIP streams (closes launch gaps and kernel tails, see __init__).
        n_streams = self.n_streams if (dev.type == "cuda" and not late) else 1
        if n_streams > 1 and self._graphs_wanted():
            n_streams = 1  # launch-bound rounds gain nothing from a second stream; keep replays in sequence
        streams = self._chunk_streams(n_streams)
        pending = []
        done = []  # per chunk: event at its end on its stream (streamed mode, when pieces are shipped early)
        early = self.ctx.distributed and not late
        with torch.enable_grad():
            # equal-sized chunks (100 clients at 40 per chunk: 34 + 33 + 33, not 40 + 40 + 20)
            n_chunks = max(1, -(-len(batched) // self.chunk))
            per = max(1, -(-len(batched) // n_chunks))
            for ci, c0 in enumerate(range(0, len(batched), per)):
                ks = np.array(batched[c0: c0 + per], dtype=np.int64)
                rows = self._train_rows[ks]
                st = streams[ci % len(streams)] if streams else None
                if st is not None:
                    st.wait_stream(torch.cuda.current_stream())
                with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                    if self.nb == 1:
                        losses = self._train_chunk_fedsgd(theta, rows, ks)
                    else:
                        losses = self._train_chunk_general(theta, rows, ks)
                    pair = self.__dict__.pop("_graph_loss_pair", None)  # (a replayed chunk graph computed it)
                    if pair is None:
                        benign = self._chunk_consts(rows)[1]
                        pair = ((losses.double() * benign).sum(), benign.sum())
                    pending.append(pair)
                    if st is not None and early:
                        done.append(st.record_event())
                if not early:
                    continue
                if st is None:
                    if c0 + per < len(batched):
                        # everything below the next untrained row is final: ship it while the next chunk trains
                        self._exchange_upto(int(self._train_rows[batched[c0 + per]]))
                elif ci >= 1:
                    # Streamed chunks: ship one chunk behind.  The main stream (which orders the pack copy and the
                    # sends) waits for chunk ci-1 only — it has waited for all earlier ones in earlier iterations
                    # — so chunk ci, just launched on another stream, keeps the GPU busy meanwhile, and chunk
                    # ci+1 (same stream as ci-1 with two streams) inherits no dependency it did not have.
                    torch.cuda.current_stream().wait_event(done[ci - 1])
                    self._exchange_upto(int(self._train_rows[batched[c0]]))
        for st in streams or []:
            torch.cuda.current_stream().wait_stream(st)
        if self.fused is not None:
            self.fused.end_round()
        for ls, cnt in pending:
            loss_sum += ls
            loss_cnt += cnt
        for k in seq:
            r = int(self._train_rows[k])
            l = self._train_sequential(theta, r)
            if not self._malicious[r]:
                loss_sum += l
                loss_cnt += 1
        self._mom_init = True
        self._post_round_hooks()
        self.timers["local_train_s"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        self._exchange_upto(self.Kloc)
        U = self._exchange_finish()
        if self.ctx.distributed:
            pack = torch.stack([loss_sum, loss_cnt])
            self.ctx.all_reduce(pack)
            loss_sum, loss_cnt = pack[0], pack[1]
        self.timers["exchange_s"] = time.perf_counter() - t1
        self.round += 1
        return U, loss_sum, loss_cnt

    def _chunk_streams(self, n: int):
        if n <= 1:
            return None
        if self._streams is None or len(self._streams) != n:
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(n)]
        return self._streams

    def _round_end_rows(self):
        return [r for r in self._train_rows if self._specs[r] and (
            "update_clip" in self._specs[r] or self._specs[r].get("update_noise_std", 0) > 0)]

    def _has_round_end_hooks(self) -> bool:
        return bool(self._round_end_rows())

    def _post_round_hooks(self):
        """Fused ``on_train_round_end`` effects: update clip + DP noise."""
        rows = self._round_end_rows()
        if not rows:
            return
        rows = [int(r) for r in rows]
        inplace = rows[-1] - rows[0] == len(rows) - 1  # consecutive rows: work on the matrix itself, no copies
        if inplace:
            blk = self.U_row[rows[0]: rows[-1] + 1, : self.D]
        else:
            rt = torch.as_tensor(rows, device=self.device)
            blk = self.U_row[rt][:, : self.D].contiguous()
        clips = [self._specs[r].get("update_clip", float("inf")) for r in rows]
        clip = torch.tensor(clips, device=self.device)
        if any(np.isfinite(c) for c in clips):
            ops.row_scale_(blk, self._clip_coef(ops.row_sqnorm(blk), clip))
        stds = {self._specs[r].get("update_noise_std", 0.0) for r in rows}
        for std in stds:
            if std > 0:
                sel = [k for k, r in enumerate(rows) if self._specs[r].get("update_noise_std", 0.0) == std]
                pairs = torch.tensor([[k, self.local_ids[rows[k]]] for k in sel], dtype=torch.int64)
                ops.add_normal_rows_(blk, pairs, self.seed, 1000 + self.round, 0, self.D, 0.0, std, False)
        if not inplace:
            self.U_row[rt, : self.D] = blk

    # ------------------------------------------------------------------
    def exchange(self) -> torch.Tensor:
        """Row shard -> column shard (all-to-all over RCCL / xGMI), in one go."""
        self._exchange_begin()
        self._exchange_upto(self.Kloc)
        return self._exchange_finish()

    # The all-to-all is issued in ``P`` pieces of consecutive local rows so that finished rows travel while
    # later chunks still train ("a dedicated comm stream overlapping the all-to-all of finished
    # client chunks with the training of the next chunk").  Every piece is one grouped batch of point-to-point
    # sends/receives to all peers — xGMI is a full mesh, so all 7 links of a GPU carry data at once — and
    # lands directly in the receiver's rows of ``U_col``.  All ranks issue the same ``P`` batches in the same
    # order; which rows a piece covers follows from the (globally known) rows-per-rank table alone.
    # Rows of clients that do not train (update-forging attackers: a fifth of the matrix in the headline
    # configuration) are all-zero until the adversary overwrites them *after* the exchange, so they are not
    # sent: each rank ships only the span [lo, hi) of its block that holds its training rows, and every
    # receiver zeroes the rest of the sender's block locally.  The spans follow from the plan and the clients'
    # ``local_training`` flags, which every rank knows for all clients.
    def _piece_bounds(self, span, j: int):
        lo, hi = span
        P = self._n_pieces
        return lo + (j * (hi - lo)) // P, lo + ((j + 1) * (hi - lo)) // P

    def _send_spans(self):
        key = ("spans", self._spec_version)
        hit = self._row_vec_cache.get(key)
        if hit is None:
            spans = []
            for q in range(self.rt.world_size):
                flags = [bool(getattr(self.clients[c], "local_training", True)) for c in self.plan.rank_clients[q]]
                idx = [i for i, f in enumerate(flags) if f]
                spans.append((idx[0], idx[-1] + 1) if idx else (0, 0))
            hit = self._row_vec_cache[key] = spans
        return hit

    def _exchange_begin(self):
        if not self.ctx.distributed:
            return
        rows = self.plan.rows_per_rank()
        self._spans = self._send_spans()
        self._n_pieces = max(1, -(-max(hi - lo for lo, hi in self._spans) // self.chunk))
        self._next_piece = 0
        self._works = []
        self._send_keep = []
        self.ctx.stats["alltoall_bytes"] = 0
        self.ctx.stats["allgather_bytes"] = 0
        row_off = np.concatenate([[0], np.cumsum(rows)])
        for q, (lo, hi) in enumerate(self._spans):  # rows nobody sends: zero (all ranks agree on which)
            if lo > 0:
                self.U_col[row_off[q]: row_off[q] + lo].zero_()
            if hi < rows[q]:
                self.U_col[row_off[q] + hi: row_off[q + 1]].zero_()

    def _exchange_upto(self, ready_rows: int):
        """Issue (asynchronously) every not-yet-sent piece that lies entirely below local row ``ready_rows``."""
        if not self.ctx.distributed:
            return
        G, me, Dl = self.rt.world_size, self.rt.rank, self.Dl
        rows = self.plan.rows_per_rank()
        row_off = np.concatenate([[0], np.cumsum(rows)])
        while self._next_piece < self._n_pieces:
            j = self._next_piece
            a, b = self._piece_bounds(self._spans[me], j)
            if b > ready_rows:
                break
            self._next_piece += 1
            ops_ = []
            if self.layout == "allgather":
                # whole rows to every peer (contiguous slabs of the resident matrix on both sides)
                if b > a:
                    blk = self._U_store[row_off[me] + a: row_off[me] + b]
                    for q in range(G):
                        if q != me:
                            ops_.append(dist.P2POp(dist.isend, blk, q))
                for q in range(G):
                    if q == me:
                        continue
                    qa, qb = self._piece_bounds(self._spans[q], j)
                    if qb > qa:
                        ops_.append(dist.P2POp(dist.irecv, self._U_store[row_off[q] + qa: row_off[q] + qb], q))
                        self.ctx.stats["allgather_bytes"] += (qb - qa) * self._U_store.shape[1] * 4
                if ops_:
                    self._works.extend(dist.batch_isend_irecv(ops_))
                continue
            if b > a:
                # pack [rows, G*Dl] -> [G, rows, Dl]: one contiguous message per peer (RCCL point-to-point sends take
                # contiguous buffers).  The copy reads and writes this rank's rows once: 125 clients x 11.2 M x 4 B
                # = 5.6 GB each way at 8 GPUs, ~3 ms of a ~150 ms round, and it runs on the compute stream between
                # chunks while the previous piece is on the wire.  Writing the update matrix column-blocked from the
                # start would remove it, but parameters straddle block boundaries and every gradient kernel's
                # epilogue (one contiguous run per parameter and client) would have to split its stores.
                send = self.U_row[a:b].view(b - a, G, Dl).transpose(0, 1).contiguous()
                self._send_keep.append(send)
                self.U_col[row_off[me] + a: row_off[me] + b].copy_(send[me])
                for q in range(G):
                    if q != me:
                        ops_.append(dist.P2POp(dist.isend, send[q], q))
                self.ctx.stats["alltoall_bytes"] += (G - 1) * (b - a) * Dl * 4
            for q in range(G):
                if q == me:
                    continue
                qa, qb = self._piece_bounds(self._spans[q], j)
                if qb > qa:
                    ops_.append(dist.P2POp(dist.irecv, self.U_col[row_off[q] + qa: row_off[q] + qb], q))
            if ops_:
                self._works.extend(dist.batch_isend_irecv(ops_))

    def _exchange_finish(self) -> torch.Tensor:
        if not self.ctx.distributed:
            return self.U_row
        assert self._next_piece == self._n_pieces, "exchange finished before every piece was issued"
        for w in self._works:
            w.wait()
        self._works, self._send_keep = [], []
        return self.U_col

    def gather_full(self) -> torch.Tensor:
        """One-off all-gather of the update matrix (every rank ends up with ``[K, D]``).  The per-round
        "allgather" layout (``update_layout="allgather"``, BASELINE config #5) does not come through here: it
        sends finished row pieces while later chunks train, see ``_exchange_upto``."""
        if not self.ctx.distributed:
            return self.U_row[:, : self.D]
        if self.layout == "allgather":
            return self.U_col
        from blades_amd.parallel.communicator import all_gather_uneven_rows

        return all_gather_uneven_rows(self.U_row, self.plan.rows_per_rank())[:, : self.D]

    # -- masks in row order ------------------------------------------------
    def row_mask(self, flags: List[bool]) -> torch.Tensor:
        """[K] float mask in global row order from a per-client flag list."""
        f = np.asarray(flags, dtype=np.float32)[self.plan.client_of_row]
        return torch.from_numpy(f).to(self.device)

    # ------------------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, theta: torch.Tensor) -> Dict[str, float]:
        from blades_amd.parallel.runtime import heartbeat as _heartbeat

        """All clients' test shards (malicious ones too, like the reference),
        batches of ``test_bs`` with batch-statistics BN; length-weighted means."""
        theta = theta.detach()[: self.D]
        self.model.eval()
        params = self._params_from_flat(theta)
        bs = self.dataset.test_bs
        tf = self.dataset.test_transform
        st = self.test_store
        ncls = int(getattr(self.dataset, "num_classes", 10) or 10)
        ks = tuple(k for k in (1, 3, 5, 7, 9) if k < ncls) or (1,)  # reference: top-1/3/5/7/9 for 10 classes
        acc = torch.zeros(1 + len(ks), device=self.device, dtype=torch.float64)  # ce_sum, hits@k...
        count = 0

        def fwd(x):
            with _autocast(self.device, self.amp_dtype):
                return functional_call(self.model, params, (x,)).float()

        vf = self.fused.forward_only if self.fused is not None else vmap(fwd, in_dims=0, randomness="different")
        by_len: Dict[int, List[int]] = {}
        for r, n in enumerate(st.lengths):
            by_len.setdefault(int(n), []).append(r)
        if self.fused is not None:
            self.fused.begin_round()  # one weight-plane split for all evaluation batches
        for n, rs in by_len.items():
            if n == 0:
                continue
            for c0 in range(0, len(rs), self.eval_chunk):
                _heartbeat(f"evaluate chunk {c0}")
                rr = rs[c0: c0 + self.eval_chunk]
                offs = torch.as_tensor(st.offsets[rr], device=self.device)
                for b0 in range(0, n, bs):
                    m = min(bs, n - b0)
                    idx = (offs.unsqueeze(1) + torch.arange(b0, b0 + m, device=self.device).unsqueeze(0))
                    X = st.x[idx.reshape(-1)].reshape(len(rr), m, *st.x.shape[1:])
                    Y = st.y[idx.reshape(-1)].reshape(len(rr), m)
                    if tf is not None:
                        X = tf(X)
                    logits = vf(X)
                    acc += ops.ce_topk(logits.reshape(-1, logits.shape[-1]), Y.reshape(-1), ks)  # fused CE + top-k
                    count += Y.numel()
        if self.fused is not None:
            self.fused.end_round()
        acc = torch.cat([acc, torch.tensor([float(count)], device=self.device, dtype=torch.float64)])
        self.ctx.all_reduce(acc)
        vals = acc.tolist()
        n = max(vals[-1], 1.0)
        out = {"ce_loss": vals[0] / n, "num_test_samples": int(vals[-1])}
        for k, h in zip(ks, vals[1:-1]):
            out[f"acc_top_{k}"] = h / n
        return out

    # -- checkpoint -------------------------------------------------------
    # Engine state is global and keyed by client uid (sampler cursors/epochs, per-client momentum rows): rank 0
    # collects every rank's share, and on load each rank takes the uids it owns — also when the run resumes on
    # a different number of ranks.  (Collective: call on every rank.)
    def state_dict(self):
        local = {"sampler": self.sampler.state_dict(),
                 "M": None if self.M is None else {int(u): self.M[k].cpu() for k, u in enumerate(self._train_uids)}}
        shares = [local]
        if self.ctx.distributed:
            shares = [None] * self.rt.world_size if self.rt.rank == 0 else None
            dist.gather_object(local, shares, dst=0)
            shares = shares or [local]
        sampler, M = {}, None
        for sh in shares:
            sampler.update(sh["sampler"])
            if sh["M"] is not None:
                M = M or {}
                M.update(sh["M"])
        return {"sampler": sampler, "M": M, "mom_init": self._mom_init, "round": self.round}

    def load_state_dict(self, s):
        self.sampler.load_state_dict(s["sampler"])
        self.M = None
        if s.get("M") is not None and len(self._train_uids):
            self.M = torch.zeros(len(self._train_rows), self.D, device=self.device)
            for k, u in enumerate(self._train_uids):
                row = s["M"].get(int(u))
                if row is not None:
                    self.M[k].copy_(row.to(self.device))
        self._mom_init = s["mom_init"]
        self.round = s["round"]
