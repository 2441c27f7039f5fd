"""Thin runner that replaces the PopTorch runtime of the reference.

The reference hands its module to `poptorch.inferenceModel / trainingModel`,
which (a) place slice `r` of `entity_embedding` on replica `r`
(`replicaGrouping(NoGrouping, 0, OnePerGroup)`, reference
`tests/test_bess.py:146-150`), (b) split every input along dim 0 over
`device_iterations x replicas` and (c) concatenate the replicas' outputs
(reference `tests/test_bess.py:122-135,181-196`).  This module does the same
three things for HIP devices:

    runner = besskge.runtime.inference_model(model)            # all shards, 1 process
    runner = besskge.runtime.inference_model(model, group=DistributedGroup())
    res = runner(**{k: v.flatten(end_dim=1) for k, v in batch.items()})

so a notebook's `poptorch.X` lines become `besskge.runtime.X`.
"""

import dataclasses
import sys
from typing import Any, Dict, List, Optional

import torch

from besskge.bess import BessKGE
from besskge.collectives import DistributedGroup, MultiDeviceGroup, NativeGroup, ReplicaGroup, SingleProcessGroup

_MULTI_PROCESS = (DistributedGroup, NativeGroup)

_BATCH_KEYS = ("head", "relation", "tail", "negative", "triple_mask", "triple_weight", "negative_mask", "step",
               "rank_truth", "rank_filter")  # (the last two: AllScoresBESS in rank-counting mode)


@dataclasses.dataclass
class Options:
    """The subset of `poptorch.Options` the BESS notebooks use."""

    #: micro-batches consumed per call (poptorch `deviceIterations`)
    device_iterations: int = 1
    #: "all": outputs of every micro-batch; "final": only the last one; None = PopTorch's defaults
    #: (`OutputMode.Final` for a training model - the notebooks read "the loss of the last batch",
    #: `1_biokg_training_inference.ipynb` training loop - and `OutputMode.All` for an inference model)
    output_mode: Optional[str] = None
    #: distributed runs: all-gather outputs so every rank sees all replicas
    gather_outputs: bool = False
    #: inference only: micro-batches of one call are issued round-robin on this
    #: many HIP streams, so the collectives / small kernels of micro-batch i+1
    #: overlap the scoring of micro-batch i (they are independent: the tables
    #: are read-only).  Training steps depend on each other and stay in order.
    pipeline_streams: int = 2
    #: capture one micro-batch step (all kernels of forward / forward+backward+
    #: update, single process) into a hipGraph and replay it: the notebook-sized
    #: micro-batches (S = 512, K = 32) are launch-bound, a replay costs one
    #: launch.  Inputs are copied into static buffers.  Stateful optimisers
    #: work too (their state tables are updated in place by the recorded
    #: kernels, Adam's step count is kept on the device; learning rate etc. are
    #: recorded by value).  One process per GPU: with `NativeGroup` (collectives on
    #: the kernels' stream, captured with them), not with c10d's `DistributedGroup`.
    use_graphs: bool = False
    #: record one call of the runner as a STEP PLAN - the list of the library's own calls the step consists of
    #: (`besskge._native.record_plan`, csrc/plan.hip) - and replay it from C: no Python between the launches (an
    #: eager notebook-size step spends 0.24 ms of host time on 0.07 ms of kernels), collectives included
    #: (`NativeGroup`: bess_pack_exchange / bess_alltoall / bess_allreduce_sum_f32 are calls like any other - no
    #: RCCL inside a hipGraph), and runnable from one host thread per device (`MultiDeviceGroup`).  The first call
    #: with an input signature records, then replays the plan from the same start and checks that it reproduces the
    #: recorded step (a step that is not made of library calls only is refused: RuntimeError).  Inputs are copied
    #: into static buffers, as with `use_graphs`.
    use_plans: bool = False
    #: `use_graphs`: keep the recorded hipGraph_t next to its executable form, so that
    #: `Runner.graph_node_counts()` can say what a recorded step holds (kernel / memset / memcpy nodes)
    keep_graph: bool = False
    #: training: micro-batches whose gradients are summed before ONE optimiser step (poptorch
    #: `Training.gradientAccumulation`; reference `notebooks/1_biokg_training_inference.ipynb:408-417,
    #: 470-477`, `2_yago_topk_prediction.ipynb:240-280`).  A call consumes `device_iterations *
    #: gradient_accumulation` micro-batches (`batches_per_step` of the batch sampler) and makes
    #: `device_iterations` weight updates; every micro-batch of an update sees the same tables.
    gradient_accumulation: int = 1
    #: how the gradients of accumulated micro-batches and of the replicas (replicated tables) are combined -
    #: PopTorch's `Training.accumulationAndReplicationReductionType`: "sum" = gradient of the summed loss, "mean" =
    #: divided by `gradient_accumulation` (and, for the replicated tables, by n_shard unless the optimiser names
    #: its own `replica_reduction`).  None = not set by the caller: `training_model` then takes "sum" (the gradient
    #: of the summed loss, what the reference-generated fixtures pin), the PopTorch-spelled `trainingModel` takes
    #: PopTorch's own documented default, Mean - so a recipe ported line by line (`from besskge import runtime as
    #: poptorch`) steps as far as the upstream run.  The reference never sets it: DESIGN.md section 4.
    accumulation_reduction: Optional[str] = None

    def deviceIterations(self, n: int) -> "Options":  # noqa: N802 - poptorch spelling
        self.device_iterations = int(n)
        return self

    def outputMode(self, mode: Any) -> "Options":  # noqa: N802 - poptorch spelling
        name = str(getattr(mode, "name", mode)).lower()
        if name not in ("all", "final"):
            raise ValueError("output mode must be 'all' or 'final'")
        self.output_mode = name
        return self

    @property
    def Training(self) -> "_TrainingOptions":  # noqa: N802 - poptorch spelling
        """`options.Training.gradientAccumulation(k)` as the notebooks write it."""
        return _TrainingOptions(self)

    @property
    def _popart(self) -> "_Ignored":
        """`options._popart.setPatterns(dict(RemoveAllReducePattern=True))` of the notebooks: the shard
        gradient is never all-reduced here by construction - accepted, nothing to do."""
        return _Ignored()

    @property
    def batches_per_call(self) -> int:
        """Micro-batches one call of a training runner consumes (per replica)."""
        return self.device_iterations * max(1, self.gradient_accumulation)


class _Ignored:
    def __getattr__(self, name: str) -> Any:
        return lambda *a, **k: None


class _TrainingOptions:
    def __init__(self, options: Options) -> None:
        self._options = options

    def gradientAccumulation(self, k: int) -> Options:  # noqa: N802 - poptorch spelling
        if int(k) < 1:
            raise ValueError("gradientAccumulation needs a factor >= 1")
        self._options.gradient_accumulation = int(k)
        return self._options

    def accumulationAndReplicationReductionType(self, kind: Any) -> Options:  # noqa: N802
        name = str(getattr(kind, "name", kind)).lower()
        if name not in ("sum", "mean"):
            raise ValueError("reduction type must be 'sum' or 'mean'")
        self._options.accumulation_reduction = name
        return self._options


@dataclasses.dataclass
class SGD:
    """SGD (optionally with momentum / weight decay), applied sparsely to the
    rows a step touched.  Plain SGD (momentum = weight_decay = 0) takes the
    atomic / fused fast paths; otherwise contributions are first coalesced per
    unique row (K9) and the update is "lazy": untouched rows and their momentum
    buffers do not move."""

    lr: float = 0.01
    momentum: float = 0.0
    weight_decay: float = 0.0
    #: paged state: keep at most this many rows of momentum per shard (rows get one when first stepped);
    #: None = a state table of the shard's own size
    state_rows: Optional[int] = None
    #: exact DENSE semantics (torch.optim / poptorch.optim on a dense gradient): EVERY row of the shard is stepped in
    #: every update - weight decay on all rows, momentum of untouched rows keeps decaying - for shards whose fp32
    #: image fits next to them (the step's gradient rows are summed into a dense [M, W] accumulator).  The default
    #: (False) is row-lazy: only touched rows and their state move - the only affordable form for a 128 GB shard.
    dense: bool = False
    #: how replicated-parameter gradients are combined over replicas: "sum" (d of the summed replica losses),
    #: "mean", or None = as the runner's `Options.accumulation_reduction` resolves (PopTorch has ONE setting for
    #: both); a model stepped without a runner (`train_step_replicas`) sums
    replica_reduction: Optional[str] = None

    kind = 0  # BESS_OPT_SGD

    @property
    def is_plain_sgd(self) -> bool:
        return self.momentum == 0.0 and self.weight_decay == 0.0


@dataclasses.dataclass
class Adagrad:
    """Row-sparse Adagrad (semantics of torch.optim.Adagrad on sparse gradients)."""

    lr: float = 0.01
    eps: float = 1e-10
    weight_decay: float = 0.0
    replica_reduction: Optional[str] = None
    #: paged state (see :class:`SGD`)
    state_rows: Optional[int] = None
    #: exact dense semantics (see :class:`SGD`)
    dense: bool = False
    kind = 1  # BESS_OPT_ADAGRAD
    is_plain_sgd = False


@dataclasses.dataclass
class Adam:
    """Lazy Adam (torch.optim.SparseAdam semantics: moments of untouched rows are
    not decayed); `weight_decay` is decoupled (AdamW, as the notebooks' poptorch
    AdamW) and also applied to touched rows only."""

    lr: float = 0.001
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 0.0
    replica_reduction: Optional[str] = None
    #: paged state: the two moment tables hold at most this many rows per shard, a row gets its pair the
    #: first time it is stepped (BASELINE configs[4]: Adam state of a 128 GB shard's own size would be
    #: 256 GB); `BessKGE.optimizer_state_rows_used()` tells when the pool is exhausted.  None = full tables
    state_rows: Optional[int] = None
    #: exact dense semantics (see :class:`SGD`): torch.optim.AdamW / poptorch.optim.AdamW on a dense gradient - all
    #: rows decayed, the moments of untouched rows keep decaying
    dense: bool = False
    kind = 2  # BESS_OPT_ADAM
    is_plain_sgd = False


def place_shards(model: BessKGE, group: ReplicaGroup, device: torch.device,
                 dtype: Optional[torch.dtype] = None) -> None:
    """Move the hosted shard(s) and the relation table to `device`."""
    fn = model.score_fn
    emb = fn.entity_embedding.data
    n = model.sharding.n_shard
    if emb.dim() != 3:
        raise ValueError("entity_embedding must be [n_shard, max_entity_per_shard, W]")
    if emb.shape[0] == n and len(group.local_shards) != n:
        emb = emb[group.local_shards]  # keep only what this process hosts
    elif emb.shape[0] != len(group.local_shards):
        raise ValueError(
            f"entity_embedding holds {emb.shape[0]} shards, process hosts {len(group.local_shards)}"
        )
    dt = dtype or emb.dtype
    fn.entity_embedding = torch.nn.Parameter(emb.to(device=device, dtype=dt).contiguous(), requires_grad=False)
    fn.relation_embedding = torch.nn.Parameter(
        fn.relation_embedding.data.to(device=device, dtype=dt).contiguous(), requires_grad=False)
    model.entity_embedding = fn.entity_embedding
    # whatever else the scorer owns (ConvE's query network, offset buffers): replicated on the device
    for child in fn.children():
        child.to(device)
    for name, buf in list(fn.named_buffers(recurse=False)):
        setattr(fn, name, buf.to(device))
    model.attach(group, {s: i for i, s in enumerate(group.local_shards)})


class Runner:
    """Callable that steps a :class:`BessKGE` module like a PopTorch model."""

    def __init__(self, model: BessKGE, options: Optional[Options], group: Optional[ReplicaGroup],
                 device: Optional[torch.device], optimizer: Optional[Any],
                 dtype: Optional[torch.dtype] = None, default_reduction: str = "sum") -> None:
        self.model = model
        self.options = options or Options()
        self.default_reduction = default_reduction
        if self.reduction not in ("sum", "mean"):
            raise ValueError("Options.accumulation_reduction must be 'sum' or 'mean'")
        if optimizer is not None and getattr(optimizer, "replica_reduction", "") is None:
            optimizer = dataclasses.replace(optimizer, replica_reduction=self.reduction)
        n = model.sharding.n_shard
        if group is None:
            group = SingleProcessGroup(n)
        self.group = group
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.optimizer = optimizer
        place_shards(model, group, device, dtype)

    def _side_streams(self, n: int) -> List[torch.cuda.Stream]:
        have = getattr(self, "_streams", [])
        while len(have) < n:
            have.append(torch.cuda.Stream(device=self.device))
        self._streams = have
        return have[:n]

    # ------------------------------------------------------------ hipGraph path
    def _step(self, reps: List[Dict[str, torch.Tensor]]) -> List[Any]:
        if self.optimizer is not None:
            return self.model.train_step_replicas(reps, self.optimizer)  # type: ignore
        with torch.no_grad():
            return self.model.forward_replicas(reps)

    @property
    def reduction(self) -> str:
        """"sum" | "mean": the options' `accumulation_reduction`, or the default of the entry point that made the
        runner (`training_model`: sum; `trainingModel`, PopTorch's spelling: PopTorch's default, mean)."""
        return self.options.accumulation_reduction or self.default_reduction

    @property
    def _accum(self) -> int:
        return max(1, int(self.options.gradient_accumulation)) if self.optimizer is not None else 1

    def _iteration(self, batch: Dict[str, torch.Tensor], it: int) -> List[List[Any]]:
        """One device iteration = one weight update: the results of its micro-batches, in order.
        With gradient accumulation every micro-batch is differentiated against the same tables, their
        gradients are summed per row and the optimiser runs once (`BessKGE.apply_accumulated`)."""
        k = self._accum
        if k == 1:
            return [self._step(self._split(batch, it))]
        mean = self.reduction == "mean"
        pending: List[Any] = []
        outs = []
        if mean:
            self.model.__dict__["_grad_scale"] = 1.0 / k
        try:
            for j in range(k):
                outs.append(self.model.train_step_replicas(self._split(batch, it * k + j), self.optimizer,
                                                           pending=pending))  # type: ignore
        finally:
            self.model.__dict__["_grad_scale"] = 1.0
        self.model.apply_accumulated(pending, self.optimizer)  # type: ignore
        return outs

    def _call_with_graphs(self, batch: Dict[str, torch.Tensor], iters: int) -> Dict[str, torch.Tensor]:
        if isinstance(self.group, DistributedGroup):
            raise NotImplementedError(
                "use_graphs needs the collectives on the kernels' own stream: SingleProcessGroup, or NativeGroup "
                "(bess_comm_* / RCCL through the C ABI) for one process per GPU - not c10d's DistributedGroup")
        stateful = self.optimizer is not None and not getattr(self.optimizer, "is_plain_sgd", True)
        if stateful:
            # the optimiser state lives in device tables the recorded kernels update in place; Adam's step
            # count moves to the device (BessKGE._opt_desc).  Hyper-parameters are recorded by value.
            self.model._device_step = True
        # ONE graph per call: all `device_iterations` micro-batches are recorded back to back (as PopTorch runs
        # its device iterations inside one compiled program), reading their inputs from a static copy of the
        # whole call's batch - so a call costs one copy per input tensor and one graph launch, not one of each
        # per micro-batch (at the notebooks' micro-batch the four input copies, the replay and the output
        # clones were a third of the step).
        sig = (iters,) + tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items()))
        cache = self.__dict__.setdefault("_graphs", {})
        # the group destroys these graphs before its communicator goes (ReplicaGroup.release_graphs)
        self.group.register_graph_cache(cache)
        generation = self.model.__dict__.get("_state_generation", 0)
        if self.__dict__.get("_graphs_generation", generation) != generation:
            cache.clear()  # optimiser state tensors were replaced (checkpoint load): recorded addresses are stale
        self.__dict__["_graphs_generation"] = generation
        if sig not in cache:
            static = {k: torch.empty(v.shape, dtype=v.dtype, device=self.device) for k, v in batch.items()}
            for k, v in batch.items():
                static[k].copy_(v)
            training = self.optimizer is not None
            # The warm-up steps (index maps, allocator pools) and the capture must not train: everything a
            # training step writes is saved here and put back afterwards - the tables, the optimiser state
            # accumulated so far (a new input signature may turn up in the middle of a run: a last, shorter
            # batch), and the scorer's dense parameters / buffers (ConvE's network and batch-norm statistics).
            snapshot = self._training_snapshot() if training else None
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._iteration(static, 0)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph(keep_graph=True) if self.options.keep_graph else torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = [o for it in range(iters) for o in self._iteration(static, it)]
                stacked = self._stack_outputs(outs)  # the stacking of the outputs is part of the recording too
            if snapshot is not None:
                self._restore_training_snapshot(snapshot)
            cache[sig] = (graph, static, stacked)
        graph, static, stacked = cache[sig]
        for k, v in batch.items():
            # (a feed that writes its micro-batches straight into `static_inputs()` costs no copy at all)
            if v.data_ptr() != static[k].data_ptr():
                static[k].copy_(v, non_blocking=True)
        graph.replay()
        if isinstance(stacked, dict):
            return {k: v.clone() for k, v in stacked.items()}
        return stacked.clone()

    # ------------------------------------------------------------ step plans
    def _call_with_plans(self, batch: Dict[str, torch.Tensor], iters: int) -> Dict[str, torch.Tensor]:
        from besskge import _native as nat

        if isinstance(self.group, DistributedGroup):
            raise NotImplementedError(
                "use_plans needs the collectives to be calls of the library: SingleProcessGroup with one shard, or "
                "NativeGroup / MultiDeviceGroup (bess_comm_* / RCCL through the C ABI) - not c10d's DistributedGroup")
        stateful = self.optimizer is not None and not getattr(self.optimizer, "is_plain_sgd", True)
        if stateful:
            self.model._device_step = True  # Adam's step count lives on the device (as under use_graphs)
        sig = (iters,) + tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items()))
        cache = self.__dict__.setdefault("_plans", {})
        generation = self.model.__dict__.get("_state_generation", 0)
        if self.__dict__.get("_plans_generation", generation) != generation:
            cache.clear()  # optimiser state tensors were replaced (checkpoint load): recorded addresses are stale
        self.__dict__["_plans_generation"] = generation
        if sig not in cache:
            cache[sig] = self._record_plan(batch, iters)
        plan, static, stacked, _pool = cache[sig]
        for k, v in batch.items():
            if v.data_ptr() != static[k].data_ptr():
                static[k].copy_(v, non_blocking=True)
        plan.run()
        if isinstance(stacked, dict):
            return {k: v.clone() for k, v in stacked.items()}
        return stacked.clone()

    def _record_plan(self, batch: Dict[str, torch.Tensor], iters: int) -> Any:
        """Record one call (all its device iterations) as a plan and prove it: replayed from the same start it must
        leave the tables and outputs the recorded call left."""
        from besskge import _native as nat

        static = {k: torch.empty(v.shape, dtype=v.dtype, device=self.device) for k, v in batch.items()}
        for k, v in batch.items():
            static[k].copy_(v)
        training = self.optimizer is not None
        snapshot = self._training_snapshot() if training else None
        for _ in range(2):  # static index maps, allocator pools, per-stream counters: made outside the recording
            self._iteration(static, 0)
        if snapshot is not None:
            self._restore_training_snapshot(snapshot)
        torch.cuda.synchronize(self.device)
        pool = torch.cuda.MemPool()
        # Everything the step enqueues must be a call of the library.  A plan that lacks a producer - an index tensor
        # made by a torch operator - would read whatever the recording left in that buffer's (recycled) memory: row
        # ids that are not row ids.  So the recording runs under the profiler and is REFUSED, before anything is
        # replayed, if the device saw work that is not the library's (or RCCL's).
        acts = [torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]
        with torch.profiler.profile(activities=acts) as prof:
            with torch.cuda.use_mem_pool(pool, device=self.device), nat.record_plan(self.device) as plan:
                outs = [o for it in range(iters) for o in self._iteration(static, it)]
                stacked = self._stack_outputs(outs)
            torch.cuda.synchronize(self.device)
        events = prof.events()
        device_work = [e.name for e in events if e.device_type == torch.autograd.DeviceType.CUDA]
        # work the device did for a torch operator: the operator's host-side event carries the kernels / copies it
        # launched (the library's own calls - rocPRIM's scan inside bess_build_segment_index included - are no
        # torch operators); and, as a second net, device kernels that are torch's by name
        foreign = {e.name for e in events if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")
                   and len(getattr(e, "kernels", ())) > 0}
        foreign |= {n for n in device_work if "at::native" in n or n.startswith("void at::") or "at_cuda_detail" in n}
        foreign = sorted(foreign)
        if snapshot is not None:
            self._restore_training_snapshot(snapshot)
        # (every decision below is taken by all ranks together: a rank that gave up alone would leave its peers in the
        # collectives of the proof runs)
        peers_ok = self.group.all_agree(not foreign and bool(device_work) and len(plan) > 0)
        if not foreign and device_work and len(plan) > 0 and not peers_ok:
            del outs, stacked
            raise RuntimeError("use_plans: another rank could not record its step as a plan")
        if foreign or not device_work:
            del outs, stacked
            raise RuntimeError(
                "use_plans: the step is not made of library calls only - the device also ran "
                f"{[n[:80] for n in foreign[:6]] if foreign else 'nothing the profiler saw'} (a torch operator between the "
                "calls: an index tensor derived from the inputs, a concatenation, a dtype conversion of the outputs) - "
                "run it eagerly or with use_graphs")
        if len(plan) == 0:
            raise RuntimeError("use_plans: the step made no library call")
        fn = self.model.score_fn
        # The proof runs on OTHER inputs than the recording saw (every input rotated by one position along its last
        # axis: still a valid batch): whatever the step derived from its inputs with a torch operator - an index
        # tensor made by `idx[sel]`, a concatenation - sits in the plan's buffers with the recording batch's values.
        # The eager step on the rotated inputs is the reference.
        rotated = {k: v.roll(1, dims=-1) if v.dim() > 1 else v.clone() for k, v in static.items()}
        ref_outs = [o for it in range(iters) for o in self._iteration(rotated, it)]
        ref_stacked = self._stack_outputs(ref_outs)
        torch.cuda.synchronize(self.device)
        want_out = {k: v.clone() for k, v in ref_stacked.items()} if isinstance(ref_stacked, dict) else ref_stacked.clone()
        want_tables = (fn.entity_embedding.data.clone(), fn.relation_embedding.data.clone()) if training else None
        del ref_outs, ref_stacked
        if snapshot is not None:
            self._restore_training_snapshot(snapshot)
        for k, v in rotated.items():
            static[k].copy_(v)
        plan.run()
        torch.cuda.synchronize(self.device)

        def same(a: torch.Tensor, b: torch.Tensor, start: Optional[torch.Tensor] = None) -> bool:
            """Do replay (a) and eager step (b) agree?  Judged on the elements the step moved (`start`: the table
            before it): sums of fp32 atomics differ in their last bits from run to run, and where a gradient
            cancels to ~0 Adam turns that into a +-lr step - a few per cent of the moved elements; a plan that is
            missing work moves other rows, or the same rows somewhere else entirely."""
            a, b = a.float(), b.float()
            ok = torch.isclose(a, b, rtol=1e-3, atol=1e-3 * float(b.abs().max().clamp(min=1e-6)), equal_nan=True)
            if start is None:
                return float((~ok).float().mean()) <= 1e-3
            moved = (a != start.float()) | (b != start.float())
            n_moved = int(moved.sum())
            return n_moved == 0 or float((~ok & moved).sum()) <= 0.25 * n_moved

        got_out = stacked if isinstance(stacked, dict) else {"out": stacked}
        ref_out = want_out if isinstance(want_out, dict) else {"out": want_out}
        bad = [k for k in ref_out if not same(got_out[k], ref_out[k])]
        if training and not bad:
            if not same(fn.entity_embedding.data, want_tables[0], snapshot["entity"]):
                bad.append("entity_embedding")
            if not same(fn.relation_embedding.data, want_tables[1], snapshot["relation"]):
                bad.append("relation_embedding")
        if not self.group.all_agree(not bad) and not bad:
            bad = ["(on another rank)"]
        if bad:
            raise RuntimeError(
                f"use_plans: replaying the recorded calls {plan.names} does not reproduce the step ({bad} differ): "
                "something the step computes on the host changes from call to call and is not part of the plan - "
                "run it eagerly or with use_graphs")
        if snapshot is not None:
            self._restore_training_snapshot(snapshot)  # the call that recorded takes its step by replaying, below
        for k, v in batch.items():
            static[k].copy_(v)
        return plan, static, stacked, pool

    def plan_calls(self) -> Dict[Any, List[str]]:
        """`use_plans`: the entry points of every recorded step, in order, by input signature."""
        return {sig: list(entry[0].names) for sig, entry in self.__dict__.get("_plans", {}).items()}

    def graph_node_counts(self) -> Dict[Any, Dict[str, int]]:
        """`Options(use_graphs=True, keep_graph=True)`: node types of every recorded step, by input signature
        (`bess_graph_node_counts`): {"kernel": n, "memset": n, "memcpy": n, ...}."""
        from besskge import _native as nat

        if not self.options.keep_graph:
            raise RuntimeError("graph_node_counts() needs Options(use_graphs=True, keep_graph=True)")
        return {sig: nat.graph_node_counts(entry[0]) for sig, entry in self.__dict__.get("_graphs", {}).items()}

    def reset_graphs(self) -> None:
        """Destroy this runner's recorded steps (the next call records again)."""
        cache = self.__dict__.get("_graphs")
        if cache:
            for entry in list(cache.values()):
                entry[0].reset()
            cache.clear()

    def __del__(self, _finalizing: Any = sys.is_finalizing) -> None:  # pragma: no cover - collection order
        cache = self.__dict__.get("_graphs")
        if cache is None or _finalizing():
            return
        try:
            # the graphs go before anything else of the runner does (a communicator they recorded waits for them),
            # and the group stops holding the cache
            self.reset_graphs()
            self.group.unregister_graph_cache(cache)
        except Exception:
            pass

    def static_inputs(self, **batch: torch.Tensor) -> Dict[str, torch.Tensor]:
        """`use_graphs`: the device-resident input buffers the recorded step reads for inputs of these shapes
        (recording it if need be).  A feed that fills them in place - `DeviceBatchSampler.sample(out=...)`, or
        any kernel of the host program - and then calls the runner with them pays no input copy: at the
        notebooks' micro-batch the four copies were 10 % of a replayed step."""
        if not (self.options.use_graphs or self.options.use_plans):
            raise RuntimeError("static_inputs() belongs to use_graphs=True / use_plans=True")
        self(**batch)  # records the step for this signature (a training runner also takes this one step)
        sig = (self.options.device_iterations,) + tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items()))
        return dict((self._graphs if self.options.use_graphs else self._plans)[sig][1])

    def _training_snapshot(self) -> Dict[str, Any]:
        fn = self.model.score_fn
        snap: Dict[str, Any] = dict(
            entity=fn.entity_embedding.data.clone(), relation=fn.relation_embedding.data.clone(),
            dense=[(p, p.data.clone()) for p in fn.dense_parameters()],
            buffers=[(b, b.clone()) for b in fn.buffers()], opt={})
        for key, st in getattr(self.model, "_optimizer_state", {}).items():
            snap["opt"][key] = dict(step=st["step"], s=[t.clone() for t in st["s"]],
                                    step_dev=st["step_dev"].clone() if "step_dev" in st else None,
                                    paging=(st["slot_map"].clone(), st["slot_counter"].clone())
                                    if "slot_map" in st else None)
        return snap

    def _restore_training_snapshot(self, snap: Dict[str, Any]) -> None:
        fn = self.model.score_fn
        fn.entity_embedding.data.copy_(snap["entity"])
        fn.relation_embedding.data.copy_(snap["relation"])
        for p, v in snap["dense"]:
            p.data.copy_(v)
        for b, v in snap["buffers"]:
            b.copy_(v)
        for key, st in getattr(self.model, "_optimizer_state", {}).items():
            old = snap["opt"].get(key)
            # state tensors the warm-up created (or grew) go back to what they were: zero
            st["step"] = old["step"] if old else 0
            for i, t in enumerate(st["s"]):
                if old and i < len(old["s"]):
                    t.copy_(old["s"][i])
                else:
                    t.zero_()
            if "slot_map" in st:  # paged state: which row owns which state row
                if old and old.get("paging") is not None:
                    st["slot_map"].copy_(old["paging"][0])
                    st["slot_counter"].copy_(old["paging"][1])
                else:
                    st["slot_map"].fill_(-1)
                    st["slot_counter"].zero_()
            if "step_dev" in st:
                if old and old["step_dev"] is not None:
                    st["step_dev"].copy_(old["step_dev"])
                else:
                    st["step_dev"].fill_(st["step"])

    def _split(self, batch: Dict[str, torch.Tensor], it: int) -> List[Dict[str, torch.Tensor]]:
        n = self.group.n_shard
        out = []
        local = list(self.group.local_shards)
        own_rows_only = batch["relation"].shape[0] != self.options.device_iterations * self._accum * n
        for j, shard in enumerate(local):
            row = it * len(local) + j if own_rows_only else it * n + shard
            out.append({k: v[row: row + 1].to(self.device, non_blocking=True) for k, v in batch.items()})
        return out

    def __call__(self, **batch: torch.Tensor) -> Dict[str, torch.Tensor]:
        n = self.group.n_shard
        unknown = set(batch) - set(_BATCH_KEYS)
        if unknown:
            raise TypeError(f"unexpected inputs {sorted(unknown)}")
        rows = batch["relation"].shape[0]
        iters = self.options.device_iterations
        n_local = len(list(self.group.local_shards))
        micro = iters * self._accum
        if rows != micro * n and rows != micro * n_local:
            raise ValueError(
                f"inputs have {rows} rows; expected device_iterations * gradient_accumulation * n_shard = "
                f"{iters} * {self._accum} * {n}"
                " (flatten [batches_per_step, n_shard, ...] with .flatten(end_dim=1)), or"
                f" {micro} * {n_local} rows holding only this process's shards"
                " (DeviceBatchSampler(..., shards=...))"
            )
        if self.options.use_graphs:
            return self._call_with_graphs(batch, iters)
        if self.options.use_plans:
            return self._call_with_plans(batch, iters)
        collected: List[List[Dict[str, Any]]] = []
        n_streams = 1 if (self.optimizer is not None or iters == 1) else max(1, self.options.pipeline_streams)
        if isinstance(self.group, NativeGroup):
            # ONE stream for the collectives of the one RCCL communicator: two of its collectives in flight on
            # different streams may be started in different orders on different ranks (each waits for all ranks).
            # c10d serialises its collectives on its own stream; the library's run where they are issued - so all
            # micro-batches of a call are issued on the current stream (their gathers are still queued one
            # micro-batch ahead of the scoring: forward_begin / forward_finish)
            n_streams = 1
        main = torch.cuda.current_stream(self.device)
        streams = [main] if n_streams == 1 else self._side_streams(n_streams)
        for st in streams:
            if st is not main:
                st.wait_stream(main)
        pipelined = (self.optimizer is None and isinstance(self.group, _MULTI_PROCESS) and iters > 1
                     and hasattr(self.model, "forward_begin"))
        if pipelined:
            # one process per GPU: issue the gathers / all-gathers of micro-batch it + 1 before the
            # scoring of micro-batch it (BessKGE.forward_begin / forward_finish)
            with torch.no_grad():
                with torch.cuda.stream(streams[0]):
                    ctx = self.model.forward_begin(self._split(batch, 0))
                for it in range(iters):
                    nxt = None
                    if it + 1 < iters:
                        with torch.cuda.stream(streams[(it + 1) % len(streams)]):
                            nxt = self.model.forward_begin(self._split(batch, it + 1))
                    with torch.cuda.stream(streams[it % len(streams)]):
                        collected.append(self.model.forward_finish(ctx))
                    ctx = nxt
        for it in range(0 if not pipelined else iters, iters):
            with torch.cuda.stream(streams[it % len(streams)]):
                collected.extend(self._iteration(batch, it))
        for st in streams:
            if st is not main:
                main.wait_stream(st)
        return self._stack_outputs(collected)

    def _stack_outputs(self, collected: List[List[Any]]) -> Dict[str, torch.Tensor]:
        mode = self.options.output_mode or ("final" if self.optimizer is not None else "all")
        if mode == "final":
            collected = collected[-1:]
        bare = not isinstance(collected[0][0], dict)  # modules returning one tensor (AllScoresBESS)
        if bare:
            collected = [[{"out": r} for r in res] for res in collected]
        keys = collected[0][0].keys()
        out: Dict[str, torch.Tensor] = {}
        for k in keys:
            per_it = []
            for res in collected:
                vals = [r[k] if r[k].dim() > 0 else r[k].reshape(1) for r in res]
                x = torch.cat(vals, dim=0) if len(vals) > 1 else vals[0]  # (a one-piece cat is a copy launch)
                if self.options.gather_outputs and isinstance(self.group, _MULTI_PROCESS):
                    x = self.group.all_gather([x])[0].flatten(end_dim=1)
                per_it.append(x)
            out[k] = torch.cat(per_it, dim=0) if len(per_it) > 1 else per_it[0]
        return out["out"] if bare else out  # type: ignore[return-value]


class MultiDeviceRunner:
    """`Runner` for ONE process that drives n GPUs (`MultiDeviceGroup`): the call shape of the reference -
    `runner(**{k: v.flatten(end_dim=1) for k, v in batch.items()})` with the full `[bps * n_shard, ...]` batch, stacked
    outputs back (reference `tests/test_bess.py:146-150`, `pipeline.py:129-144`) - on n devices.

    Shard r of `model.score_fn.entity_embedding` goes to `devices[r]` with a copy of the relation table and of the
    scorer's dense parts; replica r is stepped by its own `Runner` over rank r's `NativeGroup`, from its own host
    thread (the library calls release the GIL; with `Options.use_plans` a step is ONE call per device).  The
    replicas' outputs come back concatenated in rank order on `devices[0]`.  `sync_to_model()` writes the trained
    shards / relation table back into the module that was handed in."""

    def __init__(self, model: BessKGE, options: Optional[Options], group: MultiDeviceGroup, optimizer: Optional[Any],
                 dtype: Optional[torch.dtype] = None, default_reduction: str = "sum") -> None:
        import copy
        from concurrent.futures import ThreadPoolExecutor

        n = group.n_shard
        if model.sharding.n_shard != n:
            raise ValueError(f"MultiDeviceGroup has {n} devices, the sharding {model.sharding.n_shard} shards")
        emb = model.score_fn.entity_embedding.data
        if emb.dim() != 3 or emb.shape[0] != n:
            raise ValueError("entity_embedding must be [n_shard, max_entity_per_shard, W]")
        self.model, self.group = model, group
        self.options = options or Options()
        self.optimizer = optimizer
        self.replicas: List[BessKGE] = []
        self.runners: List[Runner] = []
        for r, rank_group in enumerate(group.ranks):
            # a replica of the module that shares nothing mutable with the others: its own scorer (shallow copy with
            # its own parameter / buffer / submodule tables; submodules - ConvE's network - copied), its own shard
            rep = copy.copy(model)
            rep.__dict__ = dict(model.__dict__)
            rep._modules = dict(model._modules)
            rep._parameters = dict(model._parameters)
            rep._buffers = dict(model._buffers)
            fn = copy.copy(model.score_fn)
            fn.__dict__ = dict(model.score_fn.__dict__)
            fn._parameters = dict(model.score_fn._parameters)
            fn._buffers = dict(model.score_fn._buffers)
            fn._modules = {k: copy.deepcopy(m) for k, m in model.score_fn._modules.items()}
            fn.entity_embedding = torch.nn.Parameter(emb[r: r + 1], requires_grad=False)  # (placed by the Runner)
            fn.relation_embedding = torch.nn.Parameter(model.score_fn.relation_embedding.data.clone(), requires_grad=False)
            rep._modules["score_fn"] = fn
            rep.entity_embedding = fn.entity_embedding
            rep.replica_group = None
            self.replicas.append(rep)
            with torch.cuda.device(rank_group.device):
                self.runners.append(Runner(rep, copy.copy(self.options), rank_group, rank_group.device, optimizer, dtype,
                                           default_reduction=default_reduction))
        self._pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix="bess-device")

    @property
    def device(self) -> torch.device:
        return self.group.devices[0]

    def _rows_of(self, batch: Dict[str, torch.Tensor], r: int) -> Dict[str, torch.Tensor]:
        """Rank r's rows of a `[micro-batches * n, ...]` batch (micro-batch-major, as the reference flattens it)."""
        n = self.group.n_shard
        return {k: v[r::n] for k, v in batch.items()}

    def __call__(self, **batch: torch.Tensor) -> Dict[str, torch.Tensor]:
        n = self.group.n_shard
        rows = batch["relation"].shape[0]
        micro = self.options.device_iterations * (max(1, self.options.gradient_accumulation) if self.optimizer is not None else 1)
        if rows != micro * n:
            raise ValueError(f"inputs have {rows} rows; expected device_iterations * gradient_accumulation * n_shard = "
                             f"{micro} * {n} (flatten [batches_per_step, n_shard, ...] with .flatten(end_dim=1))")

        def one(r: int) -> Dict[str, torch.Tensor]:
            dev = self.group.devices[r]
            torch.cuda.set_device(dev)  # (the current device is per host thread)
            mine = {k: v.to(dev, non_blocking=True) for k, v in self._rows_of(batch, r).items()}
            out = self.runners[r](**mine)
            torch.cuda.current_stream(dev).synchronize()
            return out

        outs = [f.result() for f in [self._pool.submit(one, r) for r in range(n)]]
        if not isinstance(outs[0], dict):
            outs = [{"out": o} for o in outs]
            bare = True
        else:
            bare = False
        dev0 = self.device
        merged: Dict[str, torch.Tensor] = {}
        for k in outs[0]:
            # rank r returned its micro-batches in order: interleave to the reference's micro-batch-major stacking
            per_rank = [o[k].to(dev0) for o in outs]
            if per_rank[0].dim() == 0:
                per_rank = [x.reshape(1) for x in per_rank]
            m = per_rank[0].shape[0]
            if m % max(1, self._outs_per_rank()) == 0 and self._outs_per_rank() > 1:
                parts = [x.reshape(self._outs_per_rank(), -1, *x.shape[1:]) for x in per_rank]
                merged[k] = torch.stack(parts, dim=1).flatten(end_dim=2)
            else:
                merged[k] = torch.cat(per_rank, dim=0)
        return merged["out"] if bare else merged

    def _outs_per_rank(self) -> int:
        mode = self.options.output_mode or ("final" if self.optimizer is not None else "all")
        return 1 if mode == "final" else self.options.device_iterations

    def sync_to_model(self) -> None:
        """Write the replicas' shards (and rank 0's replicated tables) back into the module handed to the constructor."""
        with torch.no_grad():
            emb = self.model.score_fn.entity_embedding.data
            for r, rep in enumerate(self.replicas):
                emb[r].copy_(rep.score_fn.entity_embedding.data[0].to(emb.device, emb.dtype))
            self.model.score_fn.relation_embedding.data.copy_(
                self.replicas[0].score_fn.relation_embedding.data.to(self.model.score_fn.relation_embedding.device,
                                                                      self.model.score_fn.relation_embedding.dtype))

    def close(self) -> None:
        self._pool.shutdown(wait=True)
        for r in self.runners:
            r.reset_graphs()
        self.group.close()


def _runner(model: BessKGE, options: Optional[Options], group: Any, device: Optional[torch.device], optimizer: Optional[Any],
            dtype: Optional[torch.dtype], default_reduction: str = "sum") -> Any:
    if isinstance(group, MultiDeviceGroup):
        return MultiDeviceRunner(model, options, group, optimizer, dtype, default_reduction)
    return Runner(model, options, group, device, optimizer, dtype, default_reduction=default_reduction)


def inference_model(model: BessKGE, options: Optional[Options] = None, group: Optional[ReplicaGroup] = None,
                    device: Optional[torch.device] = None, dtype: Optional[torch.dtype] = None) -> Runner:
    """`poptorch.inferenceModel` analogue.  `group=MultiDeviceGroup(devices)`: one process, n GPUs."""
    model.eval()
    return _runner(model, options, group, device, None, dtype)


def training_model(model: BessKGE, options: Optional[Options] = None, optimizer: Optional[Any] = None,
                   group: Optional[ReplicaGroup] = None, device: Optional[torch.device] = None,
                   dtype: Optional[torch.dtype] = None) -> Runner:
    """`poptorch.trainingModel` analogue (forward + backward + sparse update per call).
    `group=MultiDeviceGroup(devices)`: one process, n GPUs."""
    model.train()
    return _runner(model, options, group, device, optimizer or SGD(), dtype)


# PopTorch's spellings, so that `from besskge import runtime as poptorch` keeps a notebook's lines
# (`poptorch.trainingModel(model, options=options, optimizer=opt)`; the `replicaGrouping` call that follows
# in the notebooks has no counterpart: shard r always lives with replica r)
def trainingModel(model: BessKGE, options: Optional[Options] = None, optimizer: Optional[Any] = None,  # noqa: N802
                  group: Optional[ReplicaGroup] = None, device: Optional[torch.device] = None,
                  dtype: Optional[torch.dtype] = None) -> Runner:
    """`poptorch.trainingModel` under its own name AND with its own default reduction: gradients of accumulated
    micro-batches and of the replicas are averaged (PopTorch's documented default for
    `accumulationAndReplicationReductionType` is Mean; the notebooks rely on it with `gradientAccumulation(6)` and
    four replicas) unless the options / the optimiser say otherwise."""
    model.train()
    return _runner(model, options, group, device, optimizer or SGD(), dtype, default_reduction="mean")


inferenceModel = inference_model  # noqa: N816
