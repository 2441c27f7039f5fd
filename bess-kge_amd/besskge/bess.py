"""BESS distribution schemes on MI355X: gather -> exchange -> score -> loss.

Host-side mirror of the reference interface (`besskge/bess.py:34-603`):
`BessKGE`, `EmbeddingMovingBessKGE`, `ScoreMovingBessKGE`,
`BAD_NEGATIVE_SCORE`, with the same constructor and
`forward(head, relation, tail, negative, triple_mask, triple_weight,
negative_mask) -> dict` contract (every tensor carries a leading replica dim of
1, reference `bess.py:142-176`).

How one replica's step maps to the device:

  K1  rows leaving the shard are packed by `bess_gather_rows` into the send
      buffer `[n_shard, ppp + B*K, W]`; rows that stay (heads, and everything
      when n_shard == 1) are *not* copied - kernels read them from the shard
      through an index ("row source").
  C1  one balanced all-to-all of that buffer (`ReplicaGroup.all_to_all`; RCCL).
  K2-K6  positive score, query transform, negative scores: kernels read tails /
      negatives from the receive buffer through static index maps, so the
      reference's reshape/transpose/concat of embeddings (`bess.py:356-466`)
      becomes index arithmetic on a few KiB of int32.
  K7  mask / augment kill (in place).   K8  fused loss (+ score gradients).

Training (`BessKGE.train_step`, EmbeddingMoving): backward kernels produce the
gradient of every *gathered row*; rows that came through the all-to-all travel
back through one more all-to-all (C8) and are applied to the owning shard with
a sparse atomic SGD update (K9+K10).  The shard gradient is never dense and
never all-reduced; only the replicated relation table is (C9).
"""

from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional, Tuple, Union

import torch

from besskge import _native as nat
from besskge._native import RowSource
from besskge.collectives import ReplicaGroup, SingleProcessGroup
from besskge.loss import BaseLossFunction, SampledSoftmaxCrossEntropyLoss
from besskge.negative_sampler import (
    ShardedNegativeSampler,
    TripleBasedShardedNegativeSampler,
)
from besskge.scoring import BaseScoreFunction

BAD_NEGATIVE_SCORE = -50000.0

_Batch = Dict[str, torch.Tensor]

# the inference-against-many-candidates variants live in besskge.query and are
# re-exported at the end of this module, where the reference defines them


def _i32(x: torch.Tensor) -> torch.Tensor:
    x = x if x.dtype == torch.int32 else x.to(torch.int32)
    return x.contiguous()


class _PlainSGD:
    """`table -= lr * grad` as an optimiser object (what a bare learning rate means), for the code
    paths that coalesce all contributions per unique row before the update."""

    kind = nat.OPT_SGD
    momentum = 0.0
    weight_decay = 0.0
    is_plain_sgd = True

    def __init__(self, lr: float) -> None:
        self.lr = float(lr)


class _NegGroup:
    """One (queries, corrupted side, candidate rows) scoring problem."""

    def __init__(self, side: int, sel: Optional[torch.Tensor], ent: RowSource, rel_idx: torch.Tensor,
                 neg: RowSource, shared: bool, n_per_query: int) -> None:
        self.side = side          # nat.CORRUPT_HEAD | nat.CORRUPT_TAIL
        self.sel = sel            # triple slots scored by this group (None = all)
        self.ent = ent            # the entity kept in the query (tail | head rows)
        self.rel_idx = rel_idx    # relation ids of the group's triples
        self.neg = neg            # candidate rows (shared list, or [Q * n_per_query])
        self.shared = shared
        self.n_per_query = n_per_query
        # filled by forward, consumed by backward
        self.query: Optional[torch.Tensor] = None
        self.query_ctx: Any = None  # what the scorer's query_bwd needs
        self.out: Optional[torch.Tensor] = None
        self.dq: Optional[torch.Tensor] = None  # d loss / d query from the fused training forward
        self.partials: Any = None  # ScoreMoving: online-softmax partials of this shard's negatives (fused forward)
        self.kill: Any = None  # K7 applied together with the scores (shared negatives, one group)
        self.neg_parts: Optional[List[torch.Tensor]] = None  # neg.idx = their concatenation, written by the prologue
        self.bwd_buf: Optional[torch.Tensor] = None  # targets of the shared backward, cleared by the prologue


class _SmallPlan(list):
    """Row-id lists of a shard's small update lists, in hand-over order (`BessKGE._small_plan`).  `concats`:
    {position in the plan: (address of the tensor that is the concatenation of the next k lists, k)} - the only
    update lists `_match_ahead` accepts as "several planned lists in one"."""

    def __init__(self) -> None:
        super().__init__()
        self.concats: Dict[int, Tuple[int, int]] = {}


class _ReplicaStep:
    """Per-replica state of one micro-batch (forward products + backward ctx)."""

    def __init__(self) -> None:
        self.table: torch.Tensor = None  # type: ignore  # local shard [M, W]
        self.triple_ctx: Any = None  # what the scorer's triple_bwd needs
        self.head_idx: torch.Tensor = None  # type: ignore  # [S] rows of the shard
        self.rel_idx: torch.Tensor = None  # type: ignore  # [S]
        self.tail: RowSource = None  # type: ignore  # tails of my triples
        self.groups: List[_NegGroup] = []
        self.send_idx: Optional[torch.Tensor] = None  # rows packed for the exchange
        self.recv: Optional[torch.Tensor] = None  # [n*L (+ext), W]
        self.recv_rows: int = 0  # n*L: rows that came through the all-to-all
        self.ext_src: Optional[RowSource] = None  # provenance of the rows appended to recv
        self.n: int = 1
        self.ppp: int = 0
        self.neg_shape: Tuple[int, ...] = ()  # (n, B, K) of the negative index tensor
        self.local_neg: torch.Tensor = None  # type: ignore
        self.local_tail: torch.Tensor = None  # type: ignore
        self.positive_score: torch.Tensor = None  # type: ignore
        self.negative_score: torch.Tensor = None  # type: ignore
        self.loss_norm: Optional[torch.Tensor] = None  # [S, 2] (m, L / C) of each triple's softmax (ScoreMoving, fused)
        self.kill_applied = False  # the scoring call already applied K7 (mask / augment kill)
        self.fused_qt = False  # query + positive score came out of one launch (so will their backwards)
        self.tail_pre: Optional[Tuple[torch.Tensor, torch.Tensor]] = None  # (d_head, d_tail) out of `pertriple_tail`
        # training: copy / fill jobs (dst, src | None, fill word) run by ONE launch in front of the step's kernels
        # (`nat.step_prologue`), together with the index of the step's small update lists
        self.jobs: Optional[List[Tuple[torch.Tensor, Optional[torch.Tensor], int]]] = None
        self.d_recv: Optional[torch.Tensor] = None  # [rows of recv, W] f32, cleared by the prologue (n_shard > 1)
        # training, shared negatives: (loss, d_pos, d_neg) that came out of the scoring call (K4 + K7 + K8 behind
        # `bess_neg_score_shared_fwd_loss`: one launch where the scoring kernel finishes the loss rows itself)
        self.loss_pre: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None
        # training: the shard's update goes through `bess_direct_update` - the backward kernels add their gradient
        # rows into `direct.acc` at the rows' ids, `direct_lists` names the rows (no index, no dense row gradients)
        self.direct: Any = None
        self.direct_lists: List[torch.Tensor] = []
        self.jobs_ride = False  # the step's copy / fill jobs go with the query / positive-score launch


class _PendingUpdate:
    """Gradients of one micro-batch that have not been applied yet (gradient accumulation)."""

    __slots__ = ("steps", "local_updates", "deferred", "d_rel")

    def __init__(self, steps: List[_ReplicaStep], local_updates: List[List[Tuple[torch.Tensor, torch.Tensor]]],
                 deferred: List[Tuple[torch.Tensor, _NegGroup, torch.Tensor]], d_rel: torch.Tensor) -> None:
        self.steps = steps                  # one per hosted replica
        self.local_updates = local_updates  # per replica: (rows of the shard, gradient rows) lists
        self.deferred = deferred            # (shard, per-triple group, d loss / d scores): reduced by the segmented K9
        self.d_rel = d_rel                  # relation-table gradient (shared by the micro-batches of one update)


class _MergedGroup:
    """The per-triple groups of several micro-batches seen as one scoring problem: queries, score
    gradients and row references stacked along the query axis (what the segmented K9 needs of a group)."""

    __slots__ = ("query", "n_per_query")

    def __init__(self, query: torch.Tensor, n_per_query: int) -> None:
        self.query, self.n_per_query = query, n_per_query


class BessKGE(torch.nn.Module, ABC):
    """Base class of the distributed KGE step (see module docstring)."""

    def __init__(
        self,
        negative_sampler: ShardedNegativeSampler,
        score_fn: BaseScoreFunction,
        loss_fn: Optional[BaseLossFunction] = None,
        evaluation: Optional[Any] = None,
        return_scores: bool = False,
        augment_negative: bool = False,
    ) -> None:
        """
        :param negative_sampler: sampler of corrupting entities (only its
            `flat_negative_format`, `local_sampling`, `corruption_scheme`
            attributes are read here).
        :param score_fn: scoring function (owns the embedding tables).
        :param loss_fn: loss, required for training.
        :param evaluation: object with `ranks_from_scores(pos, neg)`,
            `stacked_metrics_from_ranks(ranks, mask)` and `return_ranks`.
        :param return_scores: return positive / negative scores.
        :param augment_negative: also use the other positives of the
            micro-batch as negatives (needs negative sample sharing).
        """
        super().__init__()
        self.sharding = score_fn.sharding
        self.negative_sampler = negative_sampler
        self.score_fn = score_fn
        self.loss_fn = loss_fn
        self.evaluation = evaluation
        self.return_scores = return_scores
        self.augment_negative = augment_negative
        if not (loss_fn or evaluation or return_scores):
            raise ValueError(
                "Nothing to return. At least one of loss_fn,"
                " evaluation or return_scores needs to be != None"
            )
        if self.augment_negative:
            assert (
                score_fn.negative_sample_sharing
            ), "Negative augmentation requires negative sample sharing"
            assert not isinstance(
                self, ScoreMovingBessKGE
            ), "ScoreMovingBessKGE does not support negative augmentation"
        if negative_sampler.flat_negative_format:
            assert (
                score_fn.negative_sample_sharing
            ), "Using flat negative format requires negative sample sharing"
        elif score_fn.negative_sample_sharing and isinstance(
            self.negative_sampler, TripleBasedShardedNegativeSampler
        ):
            raise ValueError(
                "Negative sample sharing cannot be used"
                " with non-flat triple-specific negatives"
            )
        self.entity_embedding = self.score_fn.entity_embedding
        self.entity_embedding_size: int = self.score_fn.entity_embedding.shape[-1]
        #: replica group; None = a private SingleProcessGroup(n_shard) on first use
        self.replica_group: Optional[ReplicaGroup] = None
        #: position of each hosted shard inside `entity_embedding` (dim 0)
        self._shard_slot: Optional[Dict[int, int]] = None
        self._map_cache: Dict[Any, torch.Tensor] = {}

    # ------------------------------------------------------------------ setup
    @property
    def n_embedding_parameters(self) -> int:
        """Trainable parameters in the embedding tables."""
        return self.score_fn.entity_embedding.numel() + self.score_fn.relation_embedding.numel()

    def attach(self, group: ReplicaGroup, shard_slot: Optional[Dict[int, int]] = None) -> None:
        """Bind the module to a replica group.

        :param shard_slot: {shard id: index along dim 0 of `entity_embedding`}
            for the shards hosted here; default: the table holds all shards.
        """
        if group.n_shard != self.sharding.n_shard:
            raise ValueError(f"group has {group.n_shard} replicas, sharding {self.sharding.n_shard}")
        self.replica_group = group
        self._shard_slot = shard_slot

    def _group(self) -> ReplicaGroup:
        if self.replica_group is None:
            self.replica_group = SingleProcessGroup(self.sharding.n_shard)
        return self.replica_group

    def _local_table(self, shard: int) -> torch.Tensor:
        emb = self.score_fn.entity_embedding
        slot = shard if self._shard_slot is None else self._shard_slot[shard]
        if emb.dim() != 3 or slot >= emb.shape[0]:
            raise RuntimeError(
                f"entity_embedding {tuple(emb.shape)} does not hold shard {shard};"
                " use besskge.runtime to place the shards"
            )
        return emb.data[slot]

    def _aux_stream(self, device: torch.device) -> torch.cuda.Stream:
        if not hasattr(self, "_aux_streams"):
            self._aux_streams: Dict[torch.device, torch.cuda.Stream] = {}
        if device not in self._aux_streams:
            self._aux_streams[device] = torch.cuda.Stream(device=device)
        return self._aux_streams[device]

    def _unit_weight(self, device: torch.device) -> torch.Tensor:
        """[1.0] on the device, made once (the default triple weight of every step)."""
        cache = self.__dict__.setdefault("_unit_weights", {})
        if device not in cache:
            cache[device] = torch.ones(1, dtype=torch.float32, device=device)
            torch.cuda.current_stream(device).synchronize()
        return cache[device]

    def _triple_weight(self, batch: _Batch, device: torch.device) -> torch.Tensor:
        """Weights of the micro-batch's triples in the loss ([1] or [S], fp32), times the step's
        gradient scale (1 unless gradients are averaged over accumulated micro-batches / replicas)."""
        scale = self.__dict__.get("_grad_scale", 1.0)
        w = batch.get("triple_weight")
        if w is None:
            if scale == 1.0:
                return self._unit_weight(device)
            cache = self.__dict__.setdefault("_unit_weights", {})
            if (device, scale) not in cache:
                cache[(device, scale)] = torch.full((1,), scale, dtype=torch.float32, device=device)
                torch.cuda.current_stream(device).synchronize()
            return cache[(device, scale)]
        w = w.reshape(-1).to(device=device, dtype=torch.float32).contiguous()
        return w if scale == 1.0 else w * scale

    def _static_map(self, key: Any, build: Any, device: torch.device) -> torch.Tensor:
        k = (key, device)
        if k not in self._map_cache:
            self._map_cache[k] = build().to(device=device, dtype=torch.int32).contiguous()
            # built once, then read from any stream: make sure it has landed
            torch.cuda.current_stream(device).synchronize()
        return self._map_cache[k]

    # ---------------------------------------------------------------- forward
    def forward(
        self,
        head: torch.Tensor,
        relation: torch.Tensor,
        tail: torch.Tensor,
        negative: torch.Tensor,
        triple_mask: Optional[torch.Tensor] = None,
        triple_weight: Optional[torch.Tensor] = None,
        negative_mask: Optional[torch.Tensor] = None,
    ) -> Dict[str, Any]:
        """One micro-batch of one replica (this process must host exactly one).

        :param head: (1, n_shard, positive_per_partition) head rows.
        :param relation: (1, n_shard, positive_per_partition) relation ids.
        :param tail: (1, n_shard, positive_per_partition) tail rows.
        :param negative: (1, n_shard, B, padded_negative) negative rows,
            B = 1, 2 or n_shard * positive_per_partition.
        :param triple_mask: (1, n_shard, positive_per_partition) triples that
            count for the metrics.
        :param triple_weight: (1, n_shard * positive_per_partition) or (1,).
        :param negative_mask: (1, B, n_shard, padded_negative) real (non
            padding) negatives.
        :return: dict with `loss`, `positive_score`, `negative_score`,
            `ranks`, `metrics` as configured.
        """
        batch = dict(head=head, relation=relation, tail=tail, negative=negative)
        for k, v in (("triple_mask", triple_mask), ("triple_weight", triple_weight),
                     ("negative_mask", negative_mask)):
            if v is not None:
                batch[k] = v
        if len(self._group().local_shards) != 1:
            raise RuntimeError(
                "forward() steps a single replica; this process hosts"
                f" {len(self._group().local_shards)} - use forward_replicas()"
            )
        return self.forward_replicas([batch])[0]

    def forward_replicas(self, batches: List[_Batch]) -> List[Dict[str, Any]]:
        """Lock-step forward of all replicas hosted by this process."""
        steps = self._score_replicas(batches)
        return [self._finish(st, b, want_grad=False)[0] for st, b in zip(steps, batches)]

    # Two-phase form of forward_replicas for software pipelining over micro-batches:
    # `forward_begin` issues what the scoring has to wait for (row gathers and the collectives
    # that distribute queries), `forward_finish` does the rest.  Calling begin(i + 1) before
    # finish(i) puts the all-gathers of micro-batch i + 1 in front of the score all-to-all of
    # micro-batch i in the (in-order) collective queue, so they run under its scoring kernel.
    def forward_begin(self, batches: List[_Batch]) -> Dict[str, Any]:
        return dict(batches=batches)

    def forward_finish(self, ctx: Dict[str, Any]) -> List[Dict[str, Any]]:
        return self.forward_replicas(ctx["batches"])

    def _score_replicas(self, batches: List[_Batch]) -> List[_ReplicaStep]:
        group = self._group()
        if len(batches) != len(group.local_shards):
            raise ValueError(f"{len(batches)} batches for {len(group.local_shards)} local replicas")
        squeezed = []
        for b in batches:
            for k in ("head", "relation", "tail", "negative"):
                if b[k].shape[0] != 1:
                    raise ValueError(f"`{k}` must have a leading replica dim of 1, got {tuple(b[k].shape)}")
            sq: Dict[str, Any] = {k: _i32(b[k].squeeze(0)) for k in ("head", "relation", "tail", "negative")}
            # forward / training step: K7 may be applied by the scoring call itself (`score_batch` keeps
            # returning the unmasked scores, like the reference's)
            sq["_kill_from"] = b
            squeezed.append(sq)
        return self.score_batch_replicas(squeezed)

    def score_batch(
        self, head: torch.Tensor, relation: torch.Tensor, tail: torch.Tensor, negative: torch.Tensor
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Positive [S] and negative [S, n_negative] scores of one replica's
        micro-batch (inputs without the leading replica dim)."""
        if len(self._group().local_shards) != 1:
            raise RuntimeError("score_batch() steps a single replica; use score_batch_replicas()")
        st = self.score_batch_replicas(
            [dict(head=_i32(head), relation=_i32(relation), tail=_i32(tail), negative=_i32(negative))]
        )[0]
        return st.positive_score, st.negative_score

    @abstractmethod
    def score_batch_replicas(self, batches: List[_Batch]) -> List[_ReplicaStep]:
        """Scores of every local replica's micro-batch (lock-step)."""
        raise NotImplementedError

    # -------------------------------------------------- mask / loss / metrics
    def _finish(self, st: _ReplicaStep, batch: _Batch, want_grad: bool, want_norm: bool = False
                ) -> Tuple[Dict[str, Any], Optional[torch.Tensor], Optional[torch.Tensor]]:
        pos, neg = st.positive_score, st.negative_score
        dev = pos.device
        ns = self.negative_sampler
        n, ppp = st.n, st.ppp
        flat_ht = ns.flat_negative_format and ns.corruption_scheme == "ht"

        if not st.kill_applied:
            kill = self._kill_spec(batch, n, ppp, int(pos.shape[0]), dev)
            if kill is not None:
                nat.mask_scores(neg, kill[0], kill[1], kill[2], kill[3])

        out: Dict[str, Any] = dict()
        ret_neg = neg
        d_pos = d_neg = None
        if self.loss_fn:
            w = self._triple_weight(batch, dev)
            ld = self.loss_fn.kernel_desc(int(neg.shape[1]))
            if st.loss_pre is not None and not want_norm:  # the scoring call has done K8 already
                loss, d_pos, d_neg = st.loss_pre
            elif want_norm:  # ScoreMoving's fused training forward: the softmax normalisation goes back to the shards
                loss, d_pos, d_neg, st.loss_norm = nat.loss_fwd_bwd(ld, pos, neg, w, want_grad, want_norm=True)
            else:
                loss, d_pos, d_neg = nat.loss_fwd_bwd(ld, pos, neg, w, want_grad)
            scale = self.__dict__.get("_grad_scale", 1.0)
            # (gradients averaged over accumulated micro-batches / replicas travel as a scaled triple weight;
            # the value handed back stays the micro-batch's own loss)
            out["loss"] = loss if scale == 1.0 else loss / scale
            if isinstance(self.loss_fn, SampledSoftmaxCrossEntropyLoss) and self.return_scores \
                    and self.score_fn.relation_embedding.dtype == torch.float32:
                # the reference shifts fp32 negative scores in place before the
                # cross entropy (loss.py:233-237) and returns that same tensor
                ret_neg = neg + ld.ssce_shift
        if self.return_scores:
            dt = self.score_fn.relation_embedding.dtype
            out.update(positive_score=pos.to(dt), negative_score=ret_neg.to(dt))
        if self.evaluation:
            tm = batch.get("triple_mask")
            if tm is not None:
                tm = tm.flatten().to(dev)
            with torch.no_grad():
                ranks = self.evaluation.ranks_from_scores(pos, neg)
                if self.evaluation.return_ranks:
                    out["ranks"] = ranks
                out["metrics"] = self.evaluation.stacked_metrics_from_ranks(ranks, tm)
        return out, d_pos, d_neg

    def _kill_spec(self, batch: _Batch, n: int, ppp: int, S: int, dev: torch.device
                   ) -> Optional[Tuple[int, bool, int, Optional[torch.Tensor]]]:
        """K7 of this micro-batch as (diag_step, ht, ppp, mask2d) - the arguments of
        `bess_mask_scores` - or None (reference bess.py:182-245)."""
        ns = self.negative_sampler
        flat_ht = ns.flat_negative_format and ns.corruption_scheme == "ht"
        negative_mask = batch.get("negative_mask")
        mask2d = None
        if negative_mask is not None:
            # (1, B', n_shard, L) -> [B', n_shard * L]   (bess.py:182-199)
            mask2d = negative_mask.squeeze(0).flatten(start_dim=-2).to(device=dev, dtype=torch.bool).contiguous()
            if mask2d.shape[0] not in (1, 2, S):
                raise ValueError(f"negative_mask has {mask2d.shape[0]} rows")
            if mask2d.shape[0] == 2 and not flat_ht:
                raise ValueError("a 2-row negative_mask needs flat 'ht' negatives")
        if self.augment_negative:
            # true head/tail sits at column step * (position of the triple among the queries)
            step = 1 if ns.flat_negative_format else 1 + n * int(batch["negative"].shape[-1])
            return step, ns.corruption_scheme == "ht", ppp, mask2d
        if mask2d is not None:
            return 0, False, ppp if flat_ht else 0, mask2d
        return None

    # ------------------------------------------------------- optimiser step
    def _opt_state(self, table: torch.Tensor, n_state: int, state_rows: Optional[int] = None) -> Dict[str, Any]:
        """Lazily allocated per-row optimiser state of one table.  `state_rows` (the optimiser's
        `state_rows` option; only for tables with more rows than that): paged state - pools of that
        many rows plus a row -> state-row map, rows get a state row the first time they are stepped
        (a 128 GB shard cannot carry Adam's two fp32 tables of its own size)."""
        if not hasattr(self, "_optimizer_state"):
            self._optimizer_state: Dict[int, Dict[str, Any]] = {}
        st = self._optimizer_state.setdefault(table.data_ptr(), dict(step=0, s=[]))
        paged = state_rows is not None and int(state_rows) < table.shape[0]
        if paged and "slot_map" not in st and n_state > 0:
            st["capacity"] = int(state_rows)
            st["slot_map"] = torch.full((table.shape[0],), -1, dtype=torch.int32, device=table.device)
            st["slot_counter"] = torch.zeros((1,), dtype=torch.int32, device=table.device)
        shape = (st["capacity"], table.shape[1]) if "slot_map" in st else tuple(table.shape)
        while len(st["s"]) < n_state:
            st["s"].append(torch.zeros(shape, dtype=torch.float32, device=table.device))
        return st

    def optimizer_state_rows_used(self) -> Dict[int, Tuple[int, int]]:
        """{table pointer: (state rows asked for so far, capacity)} of the tables with paged optimiser
        state; asked > capacity means rows are being stepped without state (raise `state_rows`)."""
        out = {}
        for key, st in getattr(self, "_optimizer_state", {}).items():
            if "slot_map" in st:
                out[key] = (int(st["slot_counter"].item()), st["capacity"])
        return out

    def _assign_state_rows(self, table: torch.Tensor, seg: Any, keep: Optional[torch.Tensor] = None) -> None:
        st = getattr(self, "_optimizer_state", {}).get(table.data_ptr())
        if st is not None and "slot_map" in st:
            nat.assign_state_rows(seg, st["slot_map"], st["slot_counter"], st["capacity"], keep)

    def _opt_desc(self, opt: Any, table: torch.Tensor) -> Tuple[Any, Optional[torch.Tensor], Optional[torch.Tensor]]:
        """(descriptor, state1, state2) of one optimiser step on `table` (advances its step count)."""
        if opt.kind == nat.OPT_SGD:
            n_state = 1 if opt.momentum != 0.0 else 0
        else:
            n_state = 1 if opt.kind == nat.OPT_ADAGRAD else 2
        prev = getattr(self, "_optimizer_state", {}).get(table.data_ptr())
        if prev is not None and prev.get("kind", opt.kind) != opt.kind:
            # the table was stepped (or its state loaded from a checkpoint) under another optimiser: state tensors
            # mean something else there (a momentum buffer is not Adam's first moment)
            if prev["s"]:
                raise RuntimeError(
                    f"besskge: this table's optimiser state belongs to optimiser kind {prev['kind']} (0 SGD, 1 Adagrad, "
                    f"2 Adam; e.g. loaded from a checkpoint), the step is taken with kind {opt.kind}")
            prev["step"] = 0  # stateless so far (plain SGD): the new optimiser counts its own steps
            prev.pop("step_dev", None)
        state = self._opt_state(table, n_state, getattr(opt, "state_rows", None))
        state["kind"] = opt.kind
        state["step"] += 1
        o = nat.OptDesc()
        o.kind, o.step, o.lr = opt.kind, state["step"], float(opt.lr)
        if "slot_map" in state:
            o.slot_map = state["slot_map"].data_ptr()
        if getattr(self, "_device_step", False) and opt.kind == nat.OPT_ADAM:
            # hipGraph replay (runtime.Options.use_graphs): the launch is recorded once, so the step count of
            # Adam's bias correction lives on the device and the increment is part of the recorded step
            if "step_dev" not in state:
                state["step_dev"] = torch.zeros((1,), dtype=torch.int32, device=table.device)
            # (a call of the library, not a torch operator: the increment is then part of a recorded plan too)
            nat.step_prologue([(state["step_dev"], state["step_dev"], 1)])
            o.step_ptr = state["step_dev"].data_ptr()
        o.momentum = float(getattr(opt, "momentum", 0.0))
        o.beta1, o.beta2 = float(getattr(opt, "beta1", 0.9)), float(getattr(opt, "beta2", 0.999))
        o.eps = float(getattr(opt, "eps", 0.0))
        o.weight_decay = float(getattr(opt, "weight_decay", 0.0))
        s = state["s"]
        return o, (s[0] if n_state > 0 else None), (s[1] if n_state > 1 else None)

    def _apply_optimizer(self, opt: Any, table: torch.Tensor, contributions: List[Tuple[torch.Tensor, torch.Tensor]],
                         ahead: Optional[Tuple[List[torch.Tensor], Any]] = None,
                         axpy: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None) -> None:
        """General K9 + K10: all (row, gradient row) lists of a table coalesced per unique row and one
        optimiser update per row (`bess_coalesced_update`: the sums are formed straight from the lists, in
        a fixed order; every touched row is written once).  `contributions` all index `table`.
        `ahead`: (row-id lists, their SegmentIndex) built earlier on the side stream (`_small_index_ahead`);
        used when the lists turned out to be exactly those."""
        ids = [i.reshape(-1) for i, _ in contributions]
        grads = [g.contiguous() for _, g in contributions]
        seg = None
        if ahead is not None:
            seg, grads = self._match_ahead(ahead, ids, grads)
        if seg is None:
            seg = nat.SegmentIndex(torch.cat(ids).contiguous(), table.shape[0],
                                   scratch=self.__dict__.setdefault("_seg_scratch", {}))
        o, s1, s2 = self._opt_desc(opt, table)
        self._assign_state_rows(table, seg)
        if len(grads) > nat.MAX_ROW_LISTS:
            grads = [torch.cat(grads, dim=0)]
        nat.coalesced_update(o, table, seg, grads, s1, s2, axpy=axpy)

    @staticmethod
    def _match_ahead(ahead: Tuple[List[torch.Tensor], Any], ids: List[torch.Tensor], grads: List[torch.Tensor]
                     ) -> Tuple[Any, List[torch.Tensor]]:
        """Is the index built ahead the index of exactly these lists?  A list of the step may be the concatenation
        of several planned lists (the candidate list of an augmented step, named by its parts): its gradient
        rows are then handed over in the same pieces.  (index | None, gradient arrays per planned list)."""
        plan, seg = ahead
        out: List[torch.Tensor] = []
        k = 0
        for i, g in zip(ids, grads):
            if k < len(plan) and plan[k].data_ptr() == i.data_ptr() and plan[k].numel() == i.numel():
                out.append(g)
                k += 1
                continue
            # the concatenation of the next planned lists?  Only the tensor the plan itself named as one (by
            # address: `_SmallPlan.concats`) - an unplanned list that merely has the length of the next few
            # planned ones must not reuse their index (its gradient rows would land on other rows' ids)
            n, start = int(i.numel()), k
            named = getattr(plan, "concats", {}).get(start)
            if named is None or named[0] != i.data_ptr() or start + named[1] > len(plan):
                return None, grads
            k = start + named[1]
            if sum(int(p_.numel()) for p_ in plan[start:k]) != n:
                return None, grads
            at = 0
            for part in plan[start:k]:
                out.append(g[at: at + part.numel()])
                at += part.numel()
        return (seg, out) if k == len(plan) else (None, grads)

    def _small_plan(self, st: _ReplicaStep, optimizer: Any) -> Optional[List[torch.Tensor]]:
        """Row-id lists of a shard's small update lists (heads, tails, shared negatives, rows returned by C8), in
        the order the backward will hand them over (`_apply_optimizer` checks that) - or None when the update
        will not coalesce them through one index (plain SGD on an fp32 shard: atomics; per-triple groups
        reduced by the segmented K9 in a way that leaves the lists unknown until the backward has run).  A
        candidate list that the step's prologue concatenates is named by its parts: the index is built in the
        same launch, from the lists as the sampler handed them over."""
        plain = not hasattr(optimizer, "kind") or optimizer.is_plain_sgd
        if plain and self.score_fn.entity_embedding.dtype == torch.float32:
            return None
        fn = self.score_fn
        plan = _SmallPlan()

        def add(src: RowSource, parts: Optional[List[torch.Tensor]] = None) -> None:
            if src.base is st.table and src.idx is not None:
                if parts is not None and len(parts) > 1:
                    plan.concats[len(plan)] = (src.idx.data_ptr(), len(parts))  # `src.idx` IS these parts, joined
                plan.extend(parts if parts is not None else [src.idx.reshape(-1)])

        head = RowSource(st.table, st.head_idx)
        if not st.fused_qt:
            add(head)
            add(st.tail)
        for g in st.groups:
            if not g.shared and g.neg.base is st.table and fn.supports_fused_segments:
                # reduced by the segmented K9.  With ONE such group and a stateful optimiser the small lists
                # ride along there (`_apply_optimizer_fused`) through an index of their own - the one planned
                # here; several groups go through `_apply_optimizer` with lists that do not exist yet
                if not (len(st.groups) == 1 and not plain and st.n == 1):
                    return None
            else:
                add(g.neg, g.neg_parts)
            if st.fused_qt:
                add(head)
                add(st.tail)
            else:
                add(g.ent)
        if not plan:
            return None
        if st.n > 1:
            if st.ext_src is not None:
                add(st.ext_src)
            plan.append(st.send_idx.reshape(-1))
        return plan

    def _small_index_ahead(self, steps: List[_ReplicaStep], optimizer: Any) -> Dict[int, Any]:
        """The index of a shard's small lists only needs their row ids, which are inputs of the step: long lists
        (more than the prologue's one workgroup indexes) are indexed here - on the side stream, right behind the
        forward kernels - instead of on the critical path after the backward."""
        out: Dict[int, Any] = {}
        for st in steps:
            plan = None if st.direct is not None else self._small_plan(st, optimizer)
            if plan is None:
                continue
            if sum(int(x.numel()) for x in plan) < 4096:
                # a short list is indexed in ~10 us: not worth a fork / join of the streams (which costs
                # about as much inside a replayed hipGraph); `_apply_optimizer` builds it where it is needed
                continue
            dev = st.table.device
            side = self._aux_stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                out[id(st)] = (plan, nat.SegmentIndex(torch.cat(plan).contiguous(), st.table.shape[0],
                                                      scratch=self.__dict__.setdefault("_seg_scratch", {})))
        return out

    #: row ids the step prologue's single workgroup indexes; longer lists go to the side stream
    prologue_index_max = 4096
    #: plain SGD on an fp32 shard: update lists with at least this many row references in total are coalesced per
    #: unique row (`bess_coalesced_update`) instead of added with fp32 atomics
    coalesce_sgd_from = 65536

    #: shards whose fp32 image (+ 4 bytes per row) is at most this many bytes get a direct-addressed accumulator
    #: (`bess_direct_update`): their small update lists need no index.  0 switches it off.
    direct_update_max_bytes = 2 << 30
    #: per-triple negatives through the fused forward: loss, score gradients and the positive / query backward in one
    #: launch behind it (`bess_pertriple_tail`); False keeps the four separate launches (tests compare the two)
    pertriple_tail = True

    def _direct_scratch(self, table: torch.Tensor) -> Any:
        held = self.__dict__.setdefault("_direct_acc", {})
        key = (table.data_ptr(), tuple(table.shape))
        if key not in held:
            held[key] = nat.DirectAccumulator(table)
        return held[key]

    def _direct_ok(self, st: _ReplicaStep, optimizer: Any, desc: Any) -> bool:
        """Does this step's shard update go through `bess_direct_update`?  One shard in the process, one group of
        shared candidates of the own shard, query + positive score fused (so are their backwards), an update that
        coalesces per row (f16 shard or a stateful optimiser - plain SGD on fp32 rows adds its atomics straight to
        the table), full-size optimiser state, a backward that can add its candidate rows by row id, and a shard
        whose fp32 image fits the scratch budget."""
        if st.n != 1 or len(st.groups) != 1 or not st.fused_qt:
            return False
        g = st.groups[0]
        if not g.shared or g.neg.base is not st.table or g.neg.idx is None or st.tail.base is not st.table:
            return False
        plain = not hasattr(optimizer, "kind") or optimizer.is_plain_sgd
        if plain and st.table.dtype == torch.float32:
            return False
        if getattr(optimizer, "dense", False):
            return False  # (every row is stepped: `_apply_dense`)
        rows = getattr(optimizer, "state_rows", None)
        if rows is not None and int(rows) < st.table.shape[0]:
            return False
        if nat.DirectAccumulator.bytes_for(st.table) > self.direct_update_max_bytes:
            return False
        return nat.shared_bwd_parts_plan(desc, len(g.ent), len(g.neg))[0] > 0

    def _launch_prologue(self, st: _ReplicaStep, optimizer: Any, d_rel: Optional[torch.Tensor]) -> Optional[Any]:
        """ONE launch in front of a training step's kernels (`bess_step_prologue`): the concatenated candidate
        list of an augmented step, the cleared relation gradient and backward targets, and - when the update
        coalesces them and they are few enough for one workgroup - the index of the step's small update lists.
        Returns what `_small_index_ahead` would have: (lists, their SegmentIndex), or None."""
        jobs = st.jobs or []
        desc = self.score_fn.kernel_desc()
        if len(jobs) + 2 <= nat.MAX_WORD_JOBS and self._direct_ok(st, optimizer, desc):
            # (the backward's sums leave as partial results with plain stores: nothing of it to clear)
            st.direct = self._direct_scratch(st.table)
            jobs.append(st.direct.increment_job())  # this step's generation number
        for g in st.groups:
            if st.direct is not None:
                break
            if g.shared and len(jobs) < nat.MAX_WORD_JOBS - 1 and self.score_fn.supports_fused_query_triple:
                g.bwd_buf = nat.shared_bwd_buffer(desc, len(g.ent), len(g.neg), st.table.device)
                if g.bwd_buf is not None:
                    jobs.append((g.bwd_buf, None, 0))
        if d_rel is not None:
            jobs.append((d_rel, None, 0))
        if st.n > 1 and st.recv is not None and len(jobs) < nat.MAX_WORD_JOBS:
            # the gradients of the rows that came through the all-to-all are summed into this (C8 sends it back)
            st.d_recv = torch.empty((st.recv.shape[0], st.recv.shape[1]), dtype=torch.float32, device=st.table.device)
            if self._recv_negatives_in_place(st):
                # the backward STORES the rows of the received negatives (every one exactly once): only the tails'
                # rows - the first ppp of each block - are sums and start from zero (1 GB less to clear per step at
                # C2's per-triple shape)
                st.d_recv.view(st.n, -1, st.d_recv.shape[1])[:, : st.ppp].zero_()
            else:
                jobs.append((st.d_recv, None, 0))
        plan = None if st.direct is not None else self._small_plan(st, optimizer)
        if plan is not None and sum(int(x.numel()) for x in plan) > self.prologue_index_max:
            plan = None
        if plan is None and st.fused_qt and st.n == 1:
            # nothing to index in a launch of its own: the jobs ride in the query / positive-score launch
            # (`fn.query_triple_fwd(jobs=...)`), none of whose inputs they write
            st.jobs = jobs
            st.jobs_ride = True
            return None
        # one workgroup indexes up to 4096 ids in ~12-15 us (notebook-size steps: cheaper than any fork / join);
        # longer lists are indexed by the device-wide pipeline on the side stream, under the forward kernels
        # (`_small_index_ahead`) - 12.5 k ids in the prologue's one workgroup were 58 us on the critical path
        if plan is not None and sum(int(x.numel()) for x in plan) > self.prologue_index_max:
            plan = None
        if plan is not None and len(plan) > nat.MAX_ROW_LISTS:
            plan = None
        seg = nat.step_prologue(jobs, plan or (), st.table.shape[0], st.table.shape[1],
                                self.__dict__.setdefault("_seg_scratch", {}))
        st.jobs = None
        return (plan, seg) if plan is not None else None

    def _recv_negatives_in_place(self, st: _ReplicaStep) -> bool:
        """Do the backward kernels of this step store the gradient rows of the received negatives straight into
        the receive-buffer gradient (`BESS_FLAG_DNEG_BY_ROW`)?  Per-triple negatives that all came through the
        all-to-all - the static index map names each of them once, nothing else adds to their rows - of a native
        scorer, no augmentation (which appends rows and makes positives candidates)."""
        fn = self.score_fn
        return bool(st.n > 1 and st.recv is not None and st.groups and not self.augment_negative and st.ext_src is None
                    and 0 <= fn._scorer_id <= nat.COMPLEX
                    and all((not g.shared) and g.neg.base is st.recv for g in st.groups)
                    and st.recv.shape[0] == st.n * (st.recv.shape[0] // st.n)
                    and sum(len(g.neg) for g in st.groups) == st.recv.shape[0] - st.n * st.ppp)

    def _apply_dense(self, steps: List[_ReplicaStep], local_updates: List[List[Tuple[torch.Tensor, torch.Tensor]]],
                     deferred: List[Tuple[torch.Tensor, _NegGroup, torch.Tensor]], seg_index: Dict[int, Any],
                     optimizer: Any, desc: nat.ModelDesc) -> None:
        """`optimizer.dense`: the step every dense optimiser of torch / PopTorch takes - EVERY row of the shard, with
        the step's gradient rows summed into a dense [M, W] fp32 matrix first (the direct-update accumulator: zero
        between steps), zero for the rows the micro-batch did not touch.  What the notebooks' `poptorch.optim.AdamW`
        does (`notebooks/1_biokg_training_inference.ipynb:525-531`): reproducible step for step, at the price of a
        pass over the whole shard and its state per update - for shards that fit, not for BASELINE configs[4]."""
        if getattr(optimizer, "state_rows", None) is not None:
            raise ValueError("a dense optimiser steps every row: it cannot have paged state (state_rows)")
        for st, upd in zip(steps, local_updates):
            acc = self._direct_scratch(st.table).acc
            lists = [(idx.reshape(-1).contiguous(), g.contiguous()) for idx, g in upd]
            for table, g, go in deferred:
                if table is st.table:  # per-triple negatives of the own shard: summed per unique row on chip first
                    seg = seg_index[id(g)]
                    gseg = nat.neg_pertriple_grad_segments(desc, g.query, table, g.n_per_query, go, seg)
                    lists.append(nat.pad_segments(seg, gseg))
            for i in range(0, len(lists), nat.MAX_ROW_LISTS):
                nat.sparse_sgd_lists(acc, lists[i: i + nat.MAX_ROW_LISTS], -1.0)  # acc += rows (fp32 atomics)
            if hasattr(optimizer, "kind") and not optimizer.is_plain_sgd:
                self._apply_optimizer_dense(optimizer, st.table, acc)
            else:
                nat.dense_sgd(st.table, acc, float(optimizer.lr) if hasattr(optimizer, "lr") else float(optimizer))
            nat.step_prologue([(acc, None, 0)])  # back to zero for the next step

    def _apply_optimizer_dense(self, opt: Any, table: torch.Tensor, grad: torch.Tensor) -> None:
        """Optimiser step on every row of a small replicated table (relation table, dense parameters):
        the rows are their own segments - nothing to sort or sum."""
        cache = self.__dict__.setdefault("_dense_segments", {})
        key = (table.data_ptr(), int(table.shape[0]))
        if key not in cache:
            cache[key] = nat.identity_segments(int(table.shape[0]), table.device)
        o, s1, s2 = self._opt_desc(opt, table)
        nat.apply_segments_opt(o, table, cache[key], grad.contiguous(), s1, s2)

    def _apply_optimizer_fused(self, opt: Any, desc: nat.ModelDesc, table: torch.Tensor, g: _NegGroup,
                               go: torch.Tensor, seg: Any, extras: List[Tuple[torch.Tensor, torch.Tensor]],
                               ahead: Optional[Tuple[List[torch.Tensor], Any]] = None) -> None:
        """K9 + K10 of a shard whose per-triple negatives form one group, for a stateful optimiser:
        the big per-row reduction applies the optimiser itself (no [unique rows, W] gradient, no host
        sync); the small lists (heads, tails, ...) are summed per unique row first and ride along, so
        every touched row still gets one update with its total gradient."""
        o, s1, s2 = self._opt_desc(opt, table)
        xmap = xsum = xseg = keep = None
        if extras:
            ids = [i.reshape(-1) for i, _ in extras]
            grads = [x.contiguous() for _, x in extras]
            xseg = None
            if ahead is not None:  # built ahead: by the step's prologue, or on the side stream behind the forward
                xseg, grads = self._match_ahead(ahead, ids, grads)
            if xseg is None:
                xseg = nat.SegmentIndex(torch.cat(ids).contiguous(), table.shape[0])
            if len(grads) > nat.MAX_ROW_LISTS:
                grads = [torch.cat(grads, dim=0)]
            xsum = nat.coalesced_update(None, table, xseg, grads, sum_only=True)
            xmap, keep = nat.map_extra_rows(seg, xseg)
        self._assign_state_rows(table, seg)
        nat.neg_pertriple_step_segments(desc, g.query, table, g.n_per_query, go, seg, o, s1, s2, xmap, xsum)
        if extras:  # rows of the small lists that no negative points at
            self._assign_state_rows(table, xseg, keep)
            nat.apply_segments_opt(o, table, xseg, xsum, s1, s2, keep=keep)

    def _wants_segments(self, g: _NegGroup, st: _ReplicaStep) -> bool:
        return not g.shared and g.neg.base is st.table and self.score_fn.supports_fused_segments

    def _segment_index_on_side(self, g: _NegGroup, st: _ReplicaStep) -> Any:
        dev = st.table.device
        side = self._aux_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))  # the row ids are ready; nothing later is waited for
        with torch.cuda.stream(side):
            return nat.SegmentIndex(g.neg.idx, st.table.shape[0], width=st.table.shape[1],
                                    scratch=self.__dict__.setdefault("_seg_scratch", {}))

    def _prefetch_segment_indices(self, steps: List[_ReplicaStep]) -> Dict[int, Any]:
        """The inverted indices of per-triple negatives only depend on the sampled indices: they are
        built on a side stream while the forward kernels run (`_run_groups_one` starts them before the
        scoring kernel of their group is queued; groups scored elsewhere are started here)."""
        seg_index: Dict[int, Any] = dict(getattr(self, "_seg_ahead", None) or {})
        for st in steps:
            for g in st.groups:
                if self._wants_segments(g, st) and id(g) not in seg_index:
                    seg_index[id(g)] = self._segment_index_on_side(g, st)
        return seg_index

    def apply_accumulated(self, pending: List[_PendingUpdate], optimizer: Any) -> None:
        """ONE optimiser step with the summed gradients of the micro-batches in `pending`
        (`train_step_replicas(..., pending=...)`; PopTorch's `Training.gradientAccumulation`, reference
        `notebooks/1_biokg_training_inference.ipynb:408-417,470-477`).  All of them were computed from the
        tables as they are now.  Row lists are concatenated per shard; the per-triple groups of the
        micro-batches become one group (queries, score gradients and references stacked), so the
        segmented reduction still sums every row's references on chip and every touched row gets one
        update - the same kernels as a single micro-batch, one index over all references."""
        if not pending:
            return
        first = pending[0]
        steps = first.steps
        local_updates: List[List[Tuple[torch.Tensor, torch.Tensor]]] = [[] for _ in steps]
        deferred: List[Tuple[torch.Tensor, Any, torch.Tensor]] = []
        seg_index: Dict[int, Any] = {}
        for p in pending:
            for mine, upd in zip(local_updates, p.local_updates):
                mine.extend(upd)
        # per-triple groups: merge the k-th group of every micro-batch (same shard, side and list length)
        by_slot: Dict[Tuple[int, int, int], List[Tuple[torch.Tensor, _NegGroup, torch.Tensor]]] = {}
        for p in pending:
            seen: Dict[Tuple[int, int], int] = {}
            for table, g, go in p.deferred:
                k = seen.get((table.data_ptr(), g.side), 0)
                seen[(table.data_ptr(), g.side)] = k + 1
                by_slot.setdefault((table.data_ptr(), g.side, k), []).append((table, g, go))
        scratch = self.__dict__.setdefault("_seg_scratch", {})
        for items in by_slot.values():
            table, g0, go0 = items[0]
            if len(items) == 1:
                merged, go, idx = g0, go0, g0.neg.idx
            else:
                if any(g.n_per_query != g0.n_per_query for _, g, _ in items):
                    raise RuntimeError("accumulated micro-batches differ in negatives per triple")
                merged = _MergedGroup(torch.cat([g.query for _, g, _ in items], dim=0), g0.n_per_query)
                go = torch.cat([x for _, _, x in items], dim=0)
                idx = torch.cat([g.neg.idx.reshape(-1) for _, g, _ in items])
            seg_index[id(merged)] = nat.SegmentIndex(idx, table.shape[0], width=table.shape[1], scratch=scratch)
            deferred.append((table, merged, go))
        self.__dict__["_small_ahead"] = None
        self._apply_updates(steps, local_updates, deferred, seg_index, optimizer, self.score_fn.kernel_desc(),
                            first.d_rel)
        pending.clear()

    def _apply_updates(self, steps: List[_ReplicaStep], local_updates: List[List[Tuple[torch.Tensor, torch.Tensor]]],
                       deferred: List[Tuple[torch.Tensor, _NegGroup, torch.Tensor]], seg_index: Dict[int, Any],
                       optimizer: Any, desc: nat.ModelDesc, d_rel: torch.Tensor) -> None:
        """K9 + K10 on every hosted shard and C9 + update of the relation table."""
        group = self._group()
        rel_table = self.score_fn.relation_embedding.data
        plain = not hasattr(optimizer, "kind") or optimizer.is_plain_sgd
        lr = float(optimizer.lr) if hasattr(optimizer, "lr") else float(optimizer)
        # f16 shards: a packed-f16 atomic add rounds the row once per contribution.  Plain SGD on them
        # therefore takes the coalescing path of the stateful optimisers: contributions summed per unique
        # row in fp32, one read-modify-write (one rounding) per touched row and step.
        plain_rows = plain and self.score_fn.entity_embedding.dtype == torch.float32
        if plain and not plain_rows and not hasattr(optimizer, "kind"):
            optimizer = _PlainSGD(lr)
        main = torch.cuda.current_stream(rel_table.device)
        side = self._aux_stream(rel_table.device)
        # K9 + K10.  Every gradient has been computed from the pre-update tables by now.
        main.wait_stream(side)
        # C9: replicated relation table
        # (single process: d_rel already holds the sum over the local replicas)
        (d_rel,) = group.all_reduce_sum([d_rel]) if len(group.local_shards) == 1 else (d_rel,)
        mean_over_replicas = getattr(optimizer, "replica_reduction", "sum") == "mean" and group.n_shard > 1
        if mean_over_replicas:
            d_rel = d_rel / group.n_shard
        # plain SGD on the relation table rides along in the launch that updates the (one) shard hosted here
        rel_axpy, rel_done = None, False
        if plain and not plain_rows and rel_table.dtype == steps[-1].table.dtype \
                and not any(item[0] is steps[-1].table for item in deferred):
            rel_axpy = (rel_table, d_rel, -lr)  # (with the last shard hosted here)
        if plain_rows and rel_table.dtype == steps[-1].table.dtype and local_updates and local_updates[-1] \
                and sum(int(i_.numel()) for i_, _ in local_updates[-1]) < self.coalesce_sgd_from \
                and len(local_updates[-1]) <= nat.MAX_ROW_LISTS:
            rel_axpy = (rel_table, d_rel, -lr)  # (in the launch of the last shard's atomic row updates)
        if getattr(optimizer, "dense", False):
            self._apply_dense(steps, local_updates, deferred, seg_index, optimizer, desc)
            if plain:
                nat.dense_sgd(rel_table, d_rel, lr)
            else:
                self._apply_optimizer_dense(optimizer, rel_table, d_rel)
            rel_done = True
        elif plain_rows:
            # per-triple negatives of the own shard: segmented reduction.  A shard with a
            # single such group gets the SGD step fused into the reduction; with two
            # ("ht") all row gradients are formed before the first row is changed.
            per_table: Dict[int, List[Tuple[torch.Tensor, _NegGroup, torch.Tensor]]] = {}
            for item in deferred:
                per_table.setdefault(item[0].data_ptr(), []).append(item)
            for items in per_table.values():
                if len(items) == 1:
                    table, g, go = items[0]
                    nat.neg_pertriple_grad_segments(desc, g.query, table, g.n_per_query, go, seg_index[id(g)],
                                                    fused_sgd_lr=lr)
                else:
                    grads = [nat.neg_pertriple_grad_segments(desc, g.query, table, g.n_per_query, go,
                                                             seg_index[id(g)]) for table, g, go in items]
                    for (table, g, _), gseg in zip(items, grads):
                        nat.apply_segments_sgd(table, seg_index[id(g)], gseg, lr)
            # everything else: sparse atomic SGD on the shard (duplicates accumulate), one launch per shard -
            # unless the lists are long (the gradients of per-triple negatives that C8 returned: half a million
            # 2 KB rows at C2's shape): fp32 atomics on them run at ~1.8 TB/s of read-modify-write (1.17 ms); the
            # coalescing path of the stateful optimisers - sort the ids once (0.09 ms), sum per unique row straight
            # from the lists, one write per touched row (0.3 ms) - is 2.5x faster there
            for st, upd in zip(steps, local_updates):
                if sum(int(idx.numel()) for idx, _ in upd) >= self.coalesce_sgd_from:
                    self._apply_optimizer(_PlainSGD(lr), st.table, list(upd))
                    continue
                lists = [(idx.contiguous(), g.contiguous()) for idx, g in upd]
                ride = rel_axpy if (st is steps[-1] and len(lists) <= nat.MAX_ROW_LISTS) else None
                for i in range(0, len(lists), nat.MAX_ROW_LISTS):
                    nat.sparse_sgd_lists(st.table, lists[i: i + nat.MAX_ROW_LISTS], lr, axpy=ride)
                rel_done = rel_done or ride is not None
        else:
            # non-linear optimisers need the *summed* gradient of every row first
            native = desc.scorer <= nat.COMPLEX
            for st, upd in zip(steps, local_updates):
                mine = [item for item in deferred if item[0] is st.table]
                if st.direct is not None:
                    assert not upd and not mine, "a direct-update step hands over no row gradients"
                    o, s1, s2 = self._opt_desc(optimizer, st.table)
                    last = st is steps[-1]
                    nat.direct_update(o, st.table, [x.contiguous() for x in st.direct_lists], st.direct, s1, s2,
                                      axpy=rel_axpy if last else None)
                    rel_done = rel_done or (last and rel_axpy is not None)
                    continue
                if native and len(mine) == 1:
                    table, g, go = mine[0]
                    self._apply_optimizer_fused(optimizer, desc, table, g, go, seg_index[id(g)], list(upd),
                                                (getattr(self, "_small_ahead", None) or {}).get(id(st)))
                    continue
                contrib = list(upd)
                for table, g, go in deferred:
                    if table is st.table:
                        seg = seg_index[id(g)]
                        gseg = nat.neg_pertriple_grad_segments(desc, g.query, table, g.n_per_query, go, seg)
                        # the group's unique rows with their summed gradients become one more row list; the unused
                        # tail of the segment arrays is neutralised on the device (no read-back of the row
                        # count: no host sync, and the step can be recorded into a hipGraph)
                        contrib.append(nat.pad_segments(seg, gseg))
                last = st is steps[-1]
                self._apply_optimizer(optimizer, st.table, contrib,
                                      (getattr(self, "_small_ahead", None) or {}).get(id(st)),
                                      axpy=rel_axpy if last else None)
                rel_done = rel_done or (last and rel_axpy is not None)
        if rel_done:
            pass  # updated in the shard's launch (or by the dense branch above)
        elif plain:
            nat.dense_sgd(rel_table, d_rel, lr)
        else:
            self._apply_optimizer_dense(optimizer, rel_table, d_rel)
        # dense parameters of the scorer (ConvE's query network): replicated like the relation table
        dense_grads = self.score_fn.__dict__.pop("dense_grads", {})
        for p in self.score_fn.dense_parameters():
            g = dense_grads.get(p)
            if g is None:
                continue
            (g,) = group.all_reduce_sum([g.contiguous()]) if len(group.local_shards) == 1 else (g,)
            as_table = p.data.view(p.shape[0], -1) if p.dim() > 1 else p.data.view(1, -1)
            g = g.reshape(as_table.shape).contiguous()
            if mean_over_replicas:
                g = g / group.n_shard
            if plain:
                nat.dense_sgd(as_table, g, lr)
            else:
                self._apply_optimizer_dense(optimizer, as_table, g)


    # ------------------------------------------------------ group execution
    def _run_groups_one(self, g: _NegGroup, desc: nat.ModelDesc, st: Optional[_ReplicaStep] = None,
                        fuse: Optional[Dict[str, Any]] = None, partials_loss: Any = None) -> torch.Tensor:
        ahead = getattr(self, "_seg_ahead", None)
        if ahead is not None and st is not None and self._wants_segments(g, st):
            # training: the inverted index of the negatives only needs their row ids - start it on the
            # side stream *before* the scoring kernel is queued, so that it runs under it
            ahead[id(g)] = self._segment_index_on_side(g, st)
        if g.query is None:
            g.query, g.query_ctx = self.score_fn.query_fwd(g.side, g.ent, g.rel_idx)
        with_loss = getattr(self, "_train_loss_in_scoring", None)
        if g.shared and with_loss is not None and st is not None and len(st.groups) == 1 and g.sel is None \
                and (st.kill_applied or with_loss.get("no_kill")):
            # training: K4 + K7 + K8 behind one call (one launch for the packed L1 kernel at notebook sizes)
            b = with_loss["batch"]
            w = self._triple_weight(b, g.query.device)
            g.out, loss, d_pos, d_neg = nat.neg_score_shared_fwd_loss(
                desc, self.loss_fn.kernel_desc(len(g.neg)), g.query, g.neg, st.positive_score, w, kill=g.kill)
            st.loss_pre = (loss, d_pos, d_neg)
        elif g.shared:
            g.out = nat.neg_score_shared_fwd(desc, g.query, g.neg, kill=g.kill)
        elif fuse is not None and st is not None and (
                g.neg.base is st.table or (g.neg.base is st.recv and self.score_fn.bilinear_candidates)):
            # training, nothing masked afterwards: scores and d loss / d query in one pass (the
            # backward then never re-reads the negative rows: shard rows through the segmented reduction K9;
            # rows that arrived through the all-to-all - EmbeddingMoving, n > 1 - for the bilinear scorers, whose
            # d_neg rows are coefficient x query)
            pos, w = st.positive_score, fuse["weight"]
            if g.sel is not None:
                pos = pos[g.sel].contiguous()
                w = w if w.numel() == 1 else w[g.sel].contiguous()
            # one group over all triples of the own shard, query + positive score out of one launch: everything
            # per triple that follows - d loss / d query from the pass's partials, K8, K3' + K6' - is ONE more launch
            d_rel = self.__dict__.get("_step_d_rel")
            defer = (self.pertriple_tail and st.fused_qt and g.sel is None and g.neg.base is st.table and d_rel is not None
                     and not self.evaluation and nat.pertriple_tail_supported(desc, g.n_per_query))
            ldesc = fuse["loss"](g.n_per_query)
            g.out, g.dq = nat.neg_score_pertriple_fwd_dq(desc, ldesc, g.query, g.neg, g.n_per_query, pos, w,
                                                         mask=fuse.get("mask"), defer=defer)
            if defer:
                loss, d_pos, d_neg, dh, dt = nat.pertriple_tail(
                    desc, ldesc, g.side, RowSource(st.table, st.head_idx), st.tail, self.score_fn.relation_embedding.data,
                    st.rel_idx, g.dq, pos, g.out, w, d_rel)
                st.loss_pre = (loss, d_pos, d_neg)
                st.tail_pre = (dh, dt)
        elif partials_loss is not None:
            # ScoreMoving training: this shard holds a part of each query's negatives - scores plus the partials
            # from which its share of d loss / d query is formed once the owner has normalised over all shards
            g.out, g.partials = nat.neg_score_pertriple_fwd_partials(desc, partials_loss, g.query, g.neg, g.n_per_query)
        else:
            g.out = nat.neg_score_pertriple_fwd(desc, g.query, g.neg, g.n_per_query)
        return g.out

    def _run_groups(self, st: _ReplicaStep, desc: nat.ModelDesc, fuse: Optional[Dict[str, Any]] = None
                    ) -> List[torch.Tensor]:
        return [self._run_groups_one(g, desc, st, fuse) for g in st.groups]

    def _fusable(self, batch: _Batch) -> Optional[Dict[str, Any]]:
        """Can the training forward of this micro-batch also produce d loss / d query?  Needs a
        loss taken over exactly the scores of one per-triple group, with nothing masked."""
        if self.loss_fn is None or not self.score_fn.supports_fused_forward or self.augment_negative:
            return None
        if not hasattr(self.loss_fn, "kernel_desc") or not nat.row_fits_registers(self.score_fn.kernel_desc()):
            return None  # (rows wider than a group's registers are scored in column windows: two-pass path)
        dev = self.score_fn.relation_embedding.device
        w = self._triple_weight(batch, dev)
        # a negative_mask (the padding of triple-specific negatives) is applied inside the fused pass when it
        # has one row or one per triple (`_run_groups_one`); other layouts take the two-pass path
        return dict(weight=w, loss=self.loss_fn.kernel_desc, masked=batch.get("negative_mask") is not None)


class EmbeddingMovingBessKGE(BessKGE):
    """Negatives are scored where the positive triple is (head shard): the
    embeddings of tails and negatives move, with a single all-to-all.

    Each triple is scored against `n_negative * n_shard` entities without
    negative sample sharing, `n_negative * n_shard * B` with it ("h"/"t"), or
    `n_negative * n_shard * (B > 2 ? B // 2 : 1)` for "ht"
    (reference `bess.py:308-468`).
    """

    def score_batch_replicas(self, batches: List[_Batch]) -> List[_ReplicaStep]:
        group = self._group()
        n = group.n_shard
        ns = self.negative_sampler
        fn = self.score_fn
        W = self.entity_embedding_size
        steps: List[_ReplicaStep] = []
        sends: List[torch.Tensor] = []
        exchange_negatives = n > 1 and not ns.local_sampling
        # NativeGroup: gather into the send buffer + all-to-all behind one entry point (bess_pack_exchange)
        packed_exchange = (hasattr(group, "pack_exchange")
                           and (W * self.score_fn.entity_embedding.element_size()) % 16 == 0)
        prologue = self.__dict__.get("_small_early") is not None  # a training step that applies its own update
        for shard, b in zip(group.local_shards, batches):
            st = _ReplicaStep()
            if prologue:
                st.jobs = []
            st.table = self._local_table(shard)
            dev = st.table.device
            head, rel, tail, neg = (b[k].to(dev) for k in ("head", "relation", "tail", "negative"))
            if head.shape[0] != n or neg.shape[0] != n:
                raise ValueError(f"batch is laid out for {head.shape[0]} shards, group has {n}")
            st.n, st.ppp = n, int(head.shape[1])
            st.head_idx = head.reshape(-1)
            st.rel_idx = rel.reshape(-1)
            st.neg_shape = tuple(neg.shape)
            if n > 1:
                # K1: pack what leaves the shard, block j -> replica j
                parts = [tail] + ([neg.flatten(start_dim=1)] if exchange_negatives else [])
                st.send_idx = torch.cat(parts, dim=1).contiguous()  # [n, L]
                if packed_exchange:
                    sends.append(group.pack_exchange(st.table, st.send_idx))  # K1 + C1, one native call
                else:
                    sends.append(nat.gather_rows(st.table, st.send_idx.reshape(-1)).reshape(n, -1, W))
            st.local_neg = neg  # type: ignore
            st.local_tail = tail  # type: ignore
            steps.append(st)
        if n > 1:
            recvs = sends if packed_exchange else group.all_to_all(sends)  # C1
            for st, y in zip(steps, recvs):
                st.recv = y.reshape(-1, W)
                st.recv_rows = st.recv.shape[0]

        desc = fn.kernel_desc()
        rel_table = fn.relation_embedding.data
        done: List[_ReplicaStep] = []
        fuse = getattr(self, "_train_fuse", None)
        for st, b in zip(steps, batches):
            self._build_groups(st, exchange_negatives)
            src = b.get("_kill_from")
            if src is not None and len(st.groups) == 1 and st.groups[0].shared:
                st.groups[0].kill = self._kill_spec(src, st.n, st.ppp, st.n * st.ppp, st.table.device)
                st.kill_applied = st.groups[0].kill is not None
            g0 = st.groups[0]
            st.fused_qt = (len(st.groups) == 1 and g0.sel is None and fn.supports_fused_query_triple
                           and st.tail.base.dtype == st.table.dtype)
            early = getattr(self, "_small_early", None)
            if early is not None:
                # training: what the step's kernels need prepared - the concatenated candidate list, cleared
                # gradient targets, the index of the small update lists (it needs only row ids) - in ONE launch
                d_rel = None
                if not done:  # the relation gradient is shared by the replicas hosted here
                    d_rel = self.__dict__["_step_d_rel"] = torch.empty(rel_table.shape, dtype=torch.float32,
                                                                       device=rel_table.device)
                ahead = self._launch_prologue(st, self._ahead_optimizer, d_rel)
                if ahead is not None:
                    early[id(st)] = ahead
            if st.fused_qt:
                # K2 + K3 + K6: the query of the one negative-scoring problem and the positive scores, one launch
                g0.query, st.positive_score = fn.query_triple_fwd(g0.side, RowSource(st.table, st.head_idx), st.tail,
                                                                  st.rel_idx, jobs=st.jobs if st.jobs_ride else None)
                if st.jobs_ride:
                    st.jobs = None
            else:
                st.positive_score, st.triple_ctx = fn.triple_fwd(
                    RowSource(st.table, st.head_idx), st.tail, st.rel_idx)
            if early is not None and id(st) not in early:
                # longer lists: indexed on the side stream, started before the scoring kernels are queued
                early.update(self._small_index_ahead([st], self._ahead_optimizer))
            fz = fuse[len(done)] if fuse else None
            if fz is not None and fz.get("masked"):
                # K7 inside the fused pass: one per-triple group over all triples, a mask of 1 or S rows, no 'ht' halves
                kill = self._kill_spec(b["_kill_from"], st.n, st.ppp, st.n * st.ppp, st.table.device) \
                    if b.get("_kill_from") is not None else None
                g0 = st.groups[0]
                ok = (kill is not None and kill[0] == 0 and not kill[1] and kill[3] is not None
                      and kill[3].shape[0] in (1, st.n * st.ppp) and len(st.groups) == 1 and g0.sel is None
                      and not g0.shared and g0.neg.base is st.table and kill[3].shape[1] <= g0.n_per_query)
                if ok:
                    fz = dict(fz, mask=kill[3])
                    st.kill_applied = True
                else:
                    fz = None  # two-pass path: scores, K7, loss, backward
            in_scoring = self.__dict__.get("_train_loss_in_scoring_on")
            if in_scoring and self.loss_fn is not None and hasattr(self.loss_fn, "kernel_desc") and not self.evaluation:
                src_b = b.get("_kill_from")
                if src_b is not None:
                    no_kill = (not st.kill_applied
                               and self._kill_spec(src_b, st.n, st.ppp, st.n * st.ppp, st.table.device) is None)
                    self.__dict__["_train_loss_in_scoring"] = dict(batch=src_b, no_kill=no_kill)
            try:
                outs = self._run_groups(st, desc, fz)
            finally:
                self.__dict__["_train_loss_in_scoring"] = None
            done.append(st)
            if len(outs) == 1:
                st.negative_score = outs[0]
            else:
                # "ht": first half of every block corrupts heads, second half tails
                n_, cut = st.n, st.ppp // 2
                st.negative_score = torch.cat(
                    [outs[0].reshape(n_, cut, -1), outs[1].reshape(n_, st.ppp - cut, -1)], dim=1
                ).flatten(end_dim=1).contiguous()
        return steps

    # ------------------------------------------------------------------
    def _build_groups(self, st: _ReplicaStep, exchange_negatives: bool) -> None:
        """Translate the reference's embedding reshapes (bess.py:356-466) into
        index lists over (shard | receive buffer)."""
        ns = self.negative_sampler
        fn = self.score_fn
        n, ppp = st.n, st.ppp
        S = n * ppp
        _, B, K = st.neg_shape  # type: ignore
        dev = st.table.device
        sharing = bool(fn.negative_sample_sharing)

        # --- where do tails and negatives live?
        if n == 1:
            tail_src = RowSource(st.table, st.local_tail.reshape(-1))  # type: ignore
            neg_base = st.table
            neg_idx2d = st.local_neg[0]  # type: ignore  # [B, K]
        else:
            L = ppp + (B * K if exchange_negatives else 0)
            tail_map = self._static_map(
                ("tail", n, ppp, L),
                lambda: (torch.arange(n)[:, None] * L + torch.arange(ppp)[None, :]).reshape(-1), dev)
            tail_src = RowSource(st.recv, tail_map)
            if exchange_negatives:
                neg_base = st.recv
                # recv block j = [tails(ppp) | negatives(B, K)]; wanted order [B, n*K]
                neg_idx2d = self._static_map(
                    ("neg", n, ppp, B, K),
                    lambda: (torch.arange(n)[None, :, None] * L + ppp
                             + torch.arange(B)[:, None, None] * K + torch.arange(K)[None, None, :]
                             ).reshape(B, n * K), dev)
            else:
                neg_base = st.table
                neg_idx2d = st.local_neg.transpose(0, 1).reshape(B, n * K).contiguous()  # type: ignore
        st.tail = tail_src
        head_src = RowSource(st.table, st.head_idx)
        nK = int(neg_idx2d.shape[1])
        scheme = ns.corruption_scheme

        # Augmentation concatenates the positives of the corrupted side with the
        # negatives (bess.py:369-393, 430-448): both must be addressable in one
        # row space.  Tails live in the receive buffer, heads in the shard; the
        # negatives in the receive buffer (exchanged) or in the shard (local
        # sampling, bess.py:340-347).  Where a group's positives and negatives sit in
        # different spaces, a copy of the shard-side rows is appended to the receive
        # buffer: the S head rows when heads are corrupted against exchanged
        # negatives, the local negatives when tails are corrupted against them.
        pos_of = {nat.CORRUPT_HEAD: head_src, nat.CORRUPT_TAIL: tail_src}
        base_of = {nat.CORRUPT_HEAD: neg_base, nat.CORRUPT_TAIL: neg_base}
        rows_of = {nat.CORRUPT_HEAD: neg_idx2d, nat.CORRUPT_TAIL: neg_idx2d}
        if self.augment_negative and n > 1:
            if neg_base is st.recv and scheme in ("h", "ht"):
                st.recv = torch.cat([st.recv, nat.gather_rows(st.table, st.head_idx)], dim=0)
                st.ext_src = head_src
                tail_src = st.tail = RowSource(st.recv, tail_src.idx)
                ext_idx = st.recv_rows + torch.arange(S, dtype=torch.int32, device=dev)
                pos_of[nat.CORRUPT_HEAD] = RowSource(st.recv, ext_idx)
                pos_of[nat.CORRUPT_TAIL] = tail_src
                base_of = {nat.CORRUPT_HEAD: st.recv, nat.CORRUPT_TAIL: st.recv}
            elif neg_base is st.table and scheme in ("t", "ht"):
                local_rows = neg_idx2d.reshape(-1).contiguous()
                st.recv = torch.cat([st.recv, nat.gather_rows(st.table, local_rows)], dim=0)
                st.ext_src = RowSource(st.table, local_rows)
                tail_src = st.tail = RowSource(st.recv, tail_src.idx)
                pos_of[nat.CORRUPT_TAIL] = tail_src
                base_of[nat.CORRUPT_TAIL] = st.recv
                rows_of[nat.CORRUPT_TAIL] = (st.recv_rows + torch.arange(local_rows.numel(), dtype=torch.int32, device=dev)
                                             ).reshape(neg_idx2d.shape)

        def make(side: int, sel: Optional[torch.Tensor], rows2d: torch.Tensor) -> _NegGroup:
            """rows2d: [Bg, nK] candidate rows for this group's queries."""
            ent_src = tail_src if side == nat.CORRUPT_HEAD else head_src
            pos_src = pos_of[side]  # the corrupted side
            rel = st.rel_idx
            if sel is not None:
                ent_src = RowSource(ent_src.base, ent_src.idx[sel])
                pos_src = RowSource(pos_src.base, pos_src.idx[sel])
                rel = rel[sel].contiguous()
            Q = len(ent_src)
            Bg = int(rows2d.shape[0])
            base = base_of[side]
            if self.augment_negative:
                # positives of the group become extra candidates (bess.py:369-393, 430-448)
                assert pos_src.base is base
                if st.jobs is not None and Bg == 1 and len(st.jobs) + 4 <= nat.MAX_WORD_JOBS:
                    # training step: the concatenation is one of the jobs of the step's prologue launch
                    parts = [pos_src.idx.reshape(-1).contiguous(), rows2d.reshape(-1).contiguous()]
                    lst = torch.empty((parts[0].numel() + parts[1].numel(),), dtype=torch.int32, device=dev)
                    st.jobs.append((lst[: parts[0].numel()], parts[0], 0))
                    st.jobs.append((lst[parts[0].numel():], parts[1], 0))
                    g = _NegGroup(side, sel, ent_src, rel, RowSource(base, lst), True, int(lst.numel()))
                    g.neg_parts = parts
                    return g
                rows2d = torch.cat([pos_src.idx.reshape(Bg, -1), rows2d], dim=1)
            if sharing or Bg == 1:
                lst = rows2d.reshape(-1).contiguous()
                return _NegGroup(side, sel, ent_src, rel, RowSource(base, lst), True, int(lst.numel()))
            if Bg != Q:
                raise ValueError(f"per-triple negatives: {Bg} candidate lists for {Q} queries")
            return _NegGroup(side, sel, ent_src, rel, RowSource(base, rows2d.reshape(-1).contiguous()),
                             False, int(rows2d.shape[1]))

        rows_hd, rows_tl = rows_of[nat.CORRUPT_HEAD], rows_of[nat.CORRUPT_TAIL]
        if scheme == "h":
            st.groups = [make(nat.CORRUPT_HEAD, None, rows_hd)]
        elif scheme == "t":
            st.groups = [make(nat.CORRUPT_TAIL, None, rows_tl)]
        elif scheme == "ht":
            cut = ppp // 2
            slot = torch.arange(S, device=dev).reshape(n, ppp)
            sel_h = slot[:, :cut].reshape(-1)
            sel_t = slot[:, cut:].reshape(-1)
            if ns.flat_negative_format:
                rows_h, rows_t = rows_hd[0:1], rows_tl[1:2]
            else:
                rows_h = rows_hd.reshape(n, ppp, nK)[:, :cut].reshape(-1, nK)
                rows_t = rows_tl.reshape(n, ppp, nK)[:, cut:].reshape(-1, nK)
            st.groups = [make(nat.CORRUPT_HEAD, sel_h, rows_h), make(nat.CORRUPT_TAIL, sel_t, rows_t)]
        else:
            raise ValueError(f"corruption scheme {scheme!r} not supported")

    # ---------------------------------------------------------------- training
    def train_step_replicas(self, batches: List[_Batch], optimizer: Any,
                            pending: Optional[List[_PendingUpdate]] = None) -> List[Dict[str, Any]]:
        """Forward + backward + sparse optimiser update of every local replica.

        `optimizer`: a learning rate (plain SGD) or one of
        `besskge.runtime.{SGD, Adagrad, Adam}`.  `pending` (gradient accumulation): a list the
        micro-batch's gradients are appended to instead of being applied; `apply_accumulated`
        applies the sum of what it holds in one optimiser step.

        Backward of the reference's autograd graph (`bess.py:322-468`) written
        out: K8' -> K4'/K5' -> K6' -> K3' give the gradient of every gathered
        row; rows received through the all-to-all are returned to their owner
        by a second all-to-all (C8); K9+K10 apply them to the shard sparsely.
        """
        if self.loss_fn is None:
            raise RuntimeError("train_step needs a loss function")
        plain = not hasattr(optimizer, "kind") or optimizer.is_plain_sgd
        lr = float(optimizer.lr) if hasattr(optimizer, "lr") else float(optimizer)
        group = self._group()
        fn = self.score_fn
        n = group.n_shard
        W = self.entity_embedding_size
        # (per-step scratch goes straight into the instance dict: nn.Module.__setattr__ costs ~2.5 us a piece)
        accumulating = pending is not None
        self.__dict__["_train_fuse"] = [self._fusable(b) for b in batches]
        # (accumulating: the references of all micro-batches are indexed together when they are applied)
        self.__dict__["_seg_ahead"] = None if accumulating else {}
        self.__dict__["_small_early"] = None if accumulating else {}
        self.__dict__["_ahead_optimizer"] = optimizer
        self.__dict__["_train_loss_in_scoring_on"] = True
        try:
            steps = self._score_replicas(batches)
            seg_index = {} if accumulating else self._prefetch_segment_indices(steps)
            self.__dict__["_small_ahead"] = dict(self._small_early or {})
        finally:
            self.__dict__["_train_fuse"] = None
            self.__dict__["_seg_ahead"] = None
            self.__dict__["_small_early"] = None
            self.__dict__["_ahead_optimizer"] = None
            self.__dict__["_train_loss_in_scoring_on"] = False
        if not accumulating:
            self._small_ahead.update(self._small_index_ahead([st for st in steps if id(st) not in self._small_ahead],
                                                             optimizer))
        desc = fn.kernel_desc()
        rel_table = fn.relation_embedding.data
        results = []
        # (the relation gradient of accumulated micro-batches is summed in the first one's buffer)
        d_rel = self.__dict__.pop("_step_d_rel", None)  # allocated and cleared by the step's prologue launch
        if pending:
            d_rel = pending[0].d_rel
        elif d_rel is None:
            d_rel = torch.zeros(rel_table.shape, dtype=torch.float32, device=rel_table.device)
        back: List[torch.Tensor] = []
        deferred: List[Tuple[torch.Tensor, _NegGroup, torch.Tensor]] = []
        local_updates: List[List[Tuple[torch.Tensor, torch.Tensor]]] = []
        for st, b in zip(steps, batches):
            out, d_pos, d_neg = self._finish(st, b, want_grad=True)
            results.append(out)
            dev = st.table.device
            upd: List[Tuple[torch.Tensor, torch.Tensor]] = []  # (rows of my shard, gradient rows)
            d_recv = None
            if n > 1:
                d_recv = st.d_recv  # cleared by the step's prologue launch
                if d_recv is None and self._recv_negatives_in_place(st):
                    d_recv = torch.empty((st.recv.shape[0], W), dtype=torch.float32, device=dev)
                    d_recv.view(n, -1, W)[:, : st.ppp].zero_()  # (the negatives' rows are stored, not summed)
                elif d_recv is None:
                    d_recv = torch.zeros((st.recv.shape[0], W), dtype=torch.float32, device=dev)

            into_recv: List[Tuple[torch.Tensor, torch.Tensor]] = []  # gradients of rows that came from other shards

            def sink(src: RowSource, g: torch.Tensor) -> None:
                if src.base is st.table:
                    upd.append((src.idx, g))
                elif src.base is st.recv:
                    into_recv.append((src.idx, g))
                else:  # pragma: no cover
                    raise RuntimeError("gradient for an unknown row space")

            # K3': positive scores (with the one group's K6' when query and score came out of one launch)
            if not st.fused_qt:
                dh, dt = fn.triple_bwd(RowSource(st.table, st.head_idx), st.tail, st.rel_idx, st.triple_ctx, d_pos,
                                       d_rel)
                sink(RowSource(st.table, st.head_idx), dh)
                sink(st.tail, dt)
            # K4'/K5' + K6': negative scores
            if len(st.groups) == 1:
                d_outs = [d_neg]
            else:
                cut = st.ppp // 2
                dn3 = d_neg.reshape(st.n, st.ppp, -1)
                d_outs = [dn3[:, :cut].reshape(-1, dn3.shape[-1]).contiguous(),
                          dn3[:, cut:].reshape(-1, dn3.shape[-1]).contiguous()]
            for g, go in zip(st.groups, d_outs):
                if g.shared and st.direct is not None:
                    # both products as partial sums (plain stores, no atomics); the query / triple backward adds them
                    # up where it reads them and puts every gradient row into the shard's accumulator at its row id
                    dq_parts, dneg_parts = nat.neg_score_shared_bwd_parts(desc, g.query, g.neg, go)
                    acc = st.direct.acc
                    fn.query_triple_bwd_parts(g.side, RowSource(st.table, st.head_idx), st.tail, st.rel_idx, d_pos,
                                              dq_parts, dneg_parts, g.neg.idx.reshape(-1), (acc, acc, acc), d_rel)
                    st.direct_lists += [g.neg.idx.reshape(-1), st.head_idx.reshape(-1), st.tail.idx.reshape(-1)]
                    continue
                elif g.shared:
                    dq, dn = nat.neg_score_shared_bwd(desc, g.query, g.neg, g.out, go, prezeroed=g.bwd_buf)
                    sink(g.neg, dn)
                elif g.neg.base is st.table and fn.supports_fused_segments:
                    # per-triple negatives read straight from the shard: no [S*N, W]
                    # gradient, no atomics - references are grouped by destination row
                    # and reduced on chip (K9), unique rows updated afterwards (K10)
                    if g.dq is not None:  # came out of the fused forward
                        dq = g.dq
                    else:
                        dq, _ = nat.neg_score_pertriple_bwd(desc, g.query, g.neg, g.n_per_query, go,
                                                            want_d_neg=False)
                    deferred.append((st.table, g, go))
                else:
                    # Negatives that arrived through the all-to-all are named once each by the static index map
                    # (`_run_groups`) and nothing else contributes to their rows of the receive-buffer gradient:
                    # the native scorers' backward stores them there directly - no [S * N, W] copy, no scatter pass
                    in_place = d_recv is not None and self._recv_negatives_in_place(st)
                    fused_dq = g.dq is not None  # fused forward over received rows (bilinear scorers)
                    dq, dn = nat.neg_score_pertriple_bwd(desc, g.query, g.neg, g.n_per_query, go,
                                                         want_d_query=not fused_dq,
                                                         d_neg_rows=d_recv if in_place else None)
                    if fused_dq:
                        dq = g.dq
                    if not in_place:
                        sink(g.neg, dn)
                if st.fused_qt:
                    if st.tail_pre is not None:  # `pertriple_tail` has done K3' + K6' behind the forward
                        dh, dt = st.tail_pre
                    else:
                        dh, dt = fn.query_triple_bwd(g.side, RowSource(st.table, st.head_idx), st.tail, st.rel_idx,
                                                     d_pos, dq, d_rel)
                    sink(RowSource(st.table, st.head_idx), dh)
                    sink(st.tail, dt)
                else:
                    dx = fn.query_bwd(g.side, g.ent, g.rel_idx, g.query_ctx, dq, d_rel)
                    sink(g.ent, dx)
            if n > 1:
                # all of them summed into the receive layout by ONE launch (d_recv -= -1 * g, fp32 atomics)
                lists = [(x.contiguous(), g.contiguous()) for x, g in into_recv if x is not None]
                for i in range(0, len(lists), nat.MAX_ROW_LISTS):
                    nat.sparse_sgd_lists(d_recv, lists[i:i + nat.MAX_ROW_LISTS], -1.0)
                for x, g in into_recv:
                    if x is None:  # rows in place
                        nat.scatter_add_rows(d_recv, None, g)
                into_recv.clear()
                if st.ext_src is not None:  # rows appended for augmentation -> their origin
                    sink(st.ext_src, d_recv[st.recv_rows:].contiguous())
                back.append(d_recv[: st.recv_rows].reshape(n, -1, W))
            local_updates.append(upd)
        if n > 1:
            returned = group.all_to_all(back)  # C8
            for st, upd, g in zip(steps, local_updates, returned):
                upd.append((st.send_idx.reshape(-1), g.reshape(-1, W)))
        if accumulating:
            self.__dict__["_small_ahead"] = None
            pending.append(_PendingUpdate(steps, local_updates, deferred, d_rel))
            return results
        try:
            self._apply_updates(steps, local_updates, deferred, seg_index, optimizer, desc, d_rel)
        finally:
            self.__dict__["_small_ahead"] = None
        return results

    def train_step(self, optimizer: Any, **batch: torch.Tensor) -> Dict[str, Any]:
        """Single-replica convenience wrapper of :meth:`train_step_replicas`."""
        return self.train_step_replicas([batch], optimizer)[0]


class ScoreMovingBessKGE(BessKGE):
    """Negatives are scored on the shard that stores them: queries are
    all-gathered, scores (not embeddings) are sent back with an all-to-all.
    No local sampling, no negative augmentation (reference `bess.py:471-603`).
    With negative sample sharing every query sees `n_shard` times the
    negatives documented for :class:`EmbeddingMovingBessKGE`.
    """

    def score_batch_replicas(self, batches: List[_Batch]) -> List[_ReplicaStep]:
        return self._score_finish(self._score_begin(batches))

    def forward_begin(self, batches: List[_Batch]) -> Dict[str, Any]:
        group = self._group()
        if len(batches) != len(group.local_shards):
            raise ValueError(f"{len(batches)} batches for {len(group.local_shards)} local replicas")
        squeezed = [{k: _i32(b[k].squeeze(0)) for k in ("head", "relation", "tail", "negative")} for b in batches]
        ctx = self._score_begin(squeezed)
        ctx["batches"] = batches
        return ctx

    def forward_finish(self, ctx: Dict[str, Any]) -> List[Dict[str, Any]]:
        steps = self._score_finish(ctx)
        return [self._finish(st, b, want_grad=False)[0] for st, b in zip(steps, ctx["batches"])]

    def _score_begin(self, batches: List[_Batch]) -> Dict[str, Any]:
        """K1 gathers and the all-gathers C2 / C3 (reference bess.py:502-518)."""
        group = self._group()
        n = group.n_shard
        ns = self.negative_sampler
        W = self.entity_embedding_size
        scheme = ns.corruption_scheme
        fn = self.score_fn
        # Tail corruption, inference, fp32 tables, native scorers: the processing replica owns the
        # kept heads *and* their relation ids, so it can send finished query rows [S, W] instead of
        # head rows + relation ids - one all-gather less, and no shard re-derives n * S queries.
        # (Heads are corrupted with queries built from tails that live on other shards than their
        # relation ids: those still travel as the reference's embeddings + ids.)
        send_queries = (scheme == "t" and not getattr(self, "_training_pass", False) and fn.supports_fused_forward
                        and fn.entity_embedding.dtype == torch.float32)
        steps: List[_ReplicaStep] = []
        tails_out, heads_q, tails_q, rels = [], [], [], []
        for shard, b in zip(group.local_shards, batches):
            st = _ReplicaStep()
            st.table = self._local_table(shard)
            dev = st.table.device
            head, rel, tail, neg = (b[k].to(dev) for k in ("head", "relation", "tail", "negative"))
            st.n, st.ppp = n, int(head.shape[1])
            cut = st.ppp // 2
            st.head_idx, st.rel_idx = head.reshape(-1), rel.reshape(-1)
            if isinstance(ns, TripleBasedShardedNegativeSampler) and ns.flat_negative_format:
                neg = neg[0:1]  # replicated along dim 0; one copy is enough (bess.py:511-517)
            st.local_neg = neg  # type: ignore
            st.local_tail = tail
            st.sm_head = head  # type: ignore[attr-defined]
            # rows of my shard that other replicas need
            tail_rows = nat.gather_rows(st.table, tail.reshape(-1)).reshape(n, st.ppp, W)
            tails_out.append(tail_rows)
            if scheme == "h":
                tails_q.append(tail_rows)
            elif send_queries:
                q, _ = fn.query_fwd(nat.CORRUPT_TAIL, RowSource(st.table, st.head_idx), st.rel_idx)
                heads_q.append(q.reshape(n, st.ppp, W))
            elif scheme == "t":
                heads_q.append(nat.gather_rows(st.table, st.head_idx).reshape(n, st.ppp, W))
            else:
                tails_q.append(tail_rows[:, :cut].contiguous())
                heads_q.append(nat.gather_rows(st.table, head[:, cut:].reshape(-1)).reshape(n, st.ppp - cut, W))
            rels.append(rel)
            steps.append(st)
        rel_all = None if send_queries else group.all_gather(rels)  # C2  [n(j), n, ppp]
        tq_all = group.all_gather(tails_q) if tails_q else None  # C3  [n(t), n(j), ., W]
        hq_all = group.all_gather(heads_q) if heads_q else None  # C3  [n(j), n(t), ., W]
        return dict(steps=steps, tails_out=tails_out, rel_all=rel_all, tq_all=tq_all, hq_all=hq_all,
                    queries_sent=send_queries)

    def _score_finish(self, ctx: Dict[str, Any]) -> List[_ReplicaStep]:
        """Local scoring of the gathered queries, score all-to-all C4 + C5, positive scores."""
        group = self._group()
        n = group.n_shard
        ns = self.negative_sampler
        fn = self.score_fn
        W = self.entity_embedding_size
        desc = fn.kernel_desc()
        sharing = bool(fn.negative_sample_sharing)
        scheme = ns.corruption_scheme
        steps, tails_out = ctx["steps"], ctx["tails_out"]
        rel_all, tq_all, hq_all = ctx["rel_all"], ctx["tq_all"], ctx["hq_all"]
        sm_fuse = getattr(self, "_sm_fuse", None)  # training, fusable loss: loss.kernel_desc (see train_step_replicas)

        scores_out = []
        for r, st in enumerate(steps):
            dev = st.table.device
            ppp, cut = st.ppp, st.ppp // 2
            neg = st.local_neg  # [n | 1, B, K]
            nB, B, K = (int(x) for x in neg.shape)
            queries_sent = bool(ctx.get("queries_sent"))
            relr = None if queries_sent else rel_all[r]

            def problem(side: int, ent_all: torch.Tensor, transpose: bool, rel_sel: torch.Tensor,
                        rows2d: torch.Tensor) -> torch.Tensor:
                """Score the gathered queries [n * Sg] (processing replica, block, triple order)
                against rows of my shard; rows2d [Bg, K]."""
                Sg = int(ent_all.shape[2])
                rows = ent_all.reshape(-1, W)
                idx = None
                if transpose:  # gathered as [t, j, p] but wanted as [j, t, p]
                    idx = self._static_map(
                        ("smT", n, Sg),
                        lambda: torch.arange(n * n * Sg).reshape(n, n, Sg).transpose(0, 1).reshape(-1), dev)
                rel_q = None if rel_sel is None else rel_sel.reshape(-1).contiguous()
                lst = rows2d.reshape(-1).contiguous()
                if sharing or rows2d.shape[0] == 1:
                    g = _NegGroup(side, None, RowSource(rows, idx), rel_q, RowSource(st.table, lst), True, int(lst.numel()))
                else:
                    if rows2d.shape[0] != rows.shape[0]:
                        raise ValueError("per-triple negatives do not match the gathered queries")
                    g = _NegGroup(side, None, RowSource(rows, idx), rel_q, RowSource(st.table, lst), False,
                                  int(rows2d.shape[1]))
                st.groups.append(g)
                if queries_sent:  # what was gathered are the finished queries
                    g.query = rows if rows.is_contiguous() else rows.contiguous()
                pl = None
                if sm_fuse is not None and not g.shared:
                    pl = sm_fuse(n * g.n_per_query)  # the loss sees the negatives of all shards
                return self._run_groups_one(g, desc, st=st, partials_loss=pl)  # st: the index of the negatives starts now

            if scheme == "h":
                sc = problem(nat.CORRUPT_HEAD, tq_all[r], True, relr, neg.reshape(nB * B, K))
            elif scheme == "t":
                sc = problem(nat.CORRUPT_TAIL, hq_all[r], False, relr, neg.reshape(nB * B, K))
            else:
                if ns.flat_negative_format:
                    rows_h, rows_t = neg[:, 0], neg[:, 1]  # [n | 1, K]
                else:
                    per = neg.reshape(n, n, ppp, K)
                    rows_h = per[:, :, :cut].reshape(-1, K)
                    rows_t = per[:, :, cut:].reshape(-1, K)
                sh = problem(nat.CORRUPT_HEAD, tq_all[r], True, relr[:, :, :cut], rows_h)
                stl = problem(nat.CORRUPT_TAIL, hq_all[r], False, relr[:, :, cut:], rows_t)
                sc = torch.cat([sh.reshape(n, n, cut, -1), stl.reshape(n, n, ppp - cut, -1)], dim=2).reshape(
                    n * n * ppp, -1)
            scores_out.append(sc.reshape(n, n * ppp, -1).contiguous())
        # C4 (scores back to the triples' owner, [n(src), S, Nl]) and C5 (positive
        # tails, [n, ppp, W]) ride one all-to-all when they have the same dtype
        scores_back, tails_in = self._paired_all_to_all(scores_out, tails_out)
        for st, sb, tl in zip(steps, scores_back, tails_in):
            st.negative_score = sb.transpose(0, 1).flatten(start_dim=1).contiguous()
            st.recv = tl.reshape(-1, W)
            st.tail = RowSource(st.recv, None)
            st.positive_score, st.triple_ctx = fn.triple_fwd(
                RowSource(st.table, st.head_idx), st.tail, st.rel_idx)
        return steps

    def _paired_all_to_all(self, xs: List[torch.Tensor], ys: List[torch.Tensor]
                           ) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        """all_to_all of two per-replica tensors [n, ...]; one collective (fewer,
        larger messages over xGMI) when dtypes agree, two otherwise."""
        group = self._group()
        if not xs or xs[0].dtype != ys[0].dtype or group.n_shard == 1:
            return group.all_to_all(xs), group.all_to_all(ys)
        n = group.n_shard
        packed = [torch.cat([x.reshape(n, -1), y.reshape(n, -1)], dim=1) for x, y in zip(xs, ys)]
        out = group.all_to_all(packed)
        xo, yo = [], []
        for o, x, y in zip(out, xs, ys):
            cut = x[0].numel()
            xo.append(o[:, :cut].reshape(x.shape))
            yo.append(o[:, cut:].reshape(y.shape))
        return xo, yo

    # ---------------------------------------------------------------- training
    def train_step_replicas(self, batches: List[_Batch], optimizer: Any,
                            pending: Optional[List[_PendingUpdate]] = None) -> List[Dict[str, Any]]:
        """Forward + backward + sparse optimiser update (ScoreMoving; `pending`: see EmbeddingMoving).

        Backward of the reference graph `bess.py:490-603`: score gradients travel
        back to the shards that produced them (all-to-all, transpose of C4); each
        shard back-propagates into its own negative rows (local update, no
        communication) and into the gathered queries, whose gradients are summed
        over shards and returned to the owner (reduce-scatter = transpose of the
        all-gather C3); positive tails return through an all-to-all (C5').
        """
        if self.loss_fn is None:
            raise RuntimeError("train_step needs a loss function")
        group = self._group()
        n = group.n_shard
        fn = self.score_fn
        W = self.entity_embedding_size
        scheme = self.negative_sampler.corruption_scheme
        accumulating = pending is not None
        self.__dict__["_training_pass"] = True  # the backward needs the gathered embeddings, not finished queries
        self.__dict__["_seg_ahead"] = None if accumulating else {}
        # Fused training forward (per-triple negatives, nothing masked, a loss whose negative weights do not need
        # the positive score): every shard keeps the online-softmax partials of the queries it scored; the second
        # pass over the negative rows (`neg_score_pertriple_bwd`) is replaced by a rescaling of those partials
        fuse = [self._fusable(b) for b in batches]
        kind = getattr(self.loss_fn, "_kind", -1)
        fused = (all(f is not None and not f.get("masked") for f in fuse)
                 and kind in (nat.LOSS_LOGSIGMOID, nat.LOSS_SSCE) and fn.supports_fused_segments)
        self.__dict__["_sm_fuse"] = self.loss_fn.kernel_desc if fused else None
        try:
            steps = self._score_replicas(batches)
            seg_index = {} if accumulating else self._prefetch_segment_indices(steps)
        finally:
            self.__dict__["_training_pass"] = False
            self.__dict__["_seg_ahead"] = None
            self.__dict__["_sm_fuse"] = None
        desc = fn.kernel_desc()
        rel_table = fn.relation_embedding.data
        d_rel = pending[0].d_rel if pending else torch.zeros(rel_table.shape, dtype=torch.float32,
                                                             device=rel_table.device)
        results, d_scores, d_tails = [], [], []
        local_updates: List[List[Tuple[torch.Tensor, torch.Tensor]]] = []
        for st, b in zip(steps, batches):
            with_norm = fused and any(g.partials is not None for g in st.groups)
            out, d_pos, d_neg = self._finish(st, b, want_grad=True, want_norm=with_norm)
            results.append(out)
            S = d_neg.shape[0]
            d_sc = d_neg.reshape(S, n, -1).transpose(0, 1)  # [n(shard), S, Nl]
            if with_norm:
                # two more columns ride with the score gradients: (m, L / C) of each triple's softmax, written by
                # the loss kernel (bess_loss_fwd_bwd_norm) - what bess_combine_dq_partials needs on the shards
                d_sc = torch.cat([d_sc, st.loss_norm.unsqueeze(0).expand(n, S, 2)], dim=2)
            d_scores.append(d_sc.contiguous())
            dh, dt = fn.triple_bwd(RowSource(st.table, st.head_idx), st.tail, st.rel_idx, st.triple_ctx, d_pos,
                                   d_rel)
            local_updates.append([(st.head_idx, dh)])
            d_tails.append(dt.reshape(n, st.ppp, W))
        # gradients of the scores I computed ([n(j), S, Nl]) and of my tail rows (C5'), one all-to-all
        d_sc_all, d_tail_back = self._paired_all_to_all(d_scores, d_tails)
        deferred: List[Tuple[torch.Tensor, _NegGroup, torch.Tensor]] = []
        d_tq, d_hq = [], []
        for st, dsc, dtb, upd in zip(steps, d_sc_all, d_tail_back, local_updates):
            ppp, cut = st.ppp, st.ppp // 2
            upd.append((st.local_tail.reshape(-1), dtb.reshape(-1, W)))
            if len(st.groups) == 1:
                d_outs = [dsc.reshape(n * n * ppp, -1)]
            else:
                d4 = dsc.reshape(n, n, ppp, -1)
                d_outs = [d4[:, :, :cut].reshape(n * n * cut, -1).contiguous(),
                          d4[:, :, cut:].reshape(n * n * (ppp - cut), -1).contiguous()]
            for g, go in zip(st.groups, d_outs):
                norm = None
                if g.partials is not None:  # the last two columns are the softmax normalisation
                    norm = go[:, -2:].contiguous()
                    go = go[:, :-2]
                go = go.contiguous()
                if g.shared:
                    dq, dn = nat.neg_score_shared_bwd(desc, g.query, g.neg, g.out, go)
                    upd.append((g.neg.idx, dn))
                elif norm is not None:
                    dq = nat.combine_dq_partials(g.partials, norm)
                    deferred.append((st.table, g, go))
                elif fn.supports_fused_segments:
                    dq, _ = nat.neg_score_pertriple_bwd(desc, g.query, g.neg, g.n_per_query, go, want_d_neg=False)
                    deferred.append((st.table, g, go))
                else:
                    dq, dn = nat.neg_score_pertriple_bwd(desc, g.query, g.neg, g.n_per_query, go)
                    upd.append((g.neg.idx, dn))
                dx = fn.query_bwd(g.side, g.ent, g.rel_idx, g.query_ctx, dq, d_rel)
                if g.ent.idx is not None:  # undo the [t, j, p] -> [j, t, p] re-ordering
                    buf = torch.empty_like(dx)
                    buf[g.ent.idx.long()] = dx
                    dx = buf
                (d_tq if g.side == nat.CORRUPT_HEAD else d_hq).append(dx.reshape(n, -1, W))
        # reduce-scatter (sum over shards) of the query gradients = all-to-all + local sum
        def to_owner(parts: List[torch.Tensor]) -> List[torch.Tensor]:
            return [x.sum(dim=0) for x in group.all_to_all(parts)] if parts else []

        back_tq, back_hq = to_owner(d_tq), to_owner(d_hq)
        for i, (st, upd) in enumerate(zip(steps, local_updates)):
            cut = st.ppp // 2
            head2d, tail2d = st.sm_head, st.local_tail  # type: ignore[attr-defined]
            if scheme == "h":
                upd.append((tail2d.reshape(-1), back_tq[i].reshape(-1, W)))
            elif scheme == "t":
                upd.append((st.head_idx, back_hq[i].reshape(-1, W)))
            else:
                upd.append((tail2d[:, :cut].reshape(-1).contiguous(), back_tq[i].reshape(-1, W)))
                upd.append((head2d[:, cut:].reshape(-1).contiguous(), back_hq[i].reshape(-1, W)))
        if accumulating:
            pending.append(_PendingUpdate(steps, local_updates, deferred, d_rel))
            return results
        self._apply_updates(steps, local_updates, deferred, seg_index, optimizer, desc, d_rel)
        return results

    def train_step(self, optimizer: Any, **batch: torch.Tensor) -> Dict[str, Any]:
        """Single-replica convenience wrapper of :meth:`train_step_replicas`."""
        return self.train_step_replicas([batch], optimizer)[0]


from besskge.query import AllScoresBESS, TopKQueryBessKGE  # noqa: E402,F401
