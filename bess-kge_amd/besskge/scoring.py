"""Scoring functions TransE / RotatE / DistMult / ComplEx on HIP kernels.

Host-side mirror of the reference interface (`besskge/scoring.py:28-462,
746-946`): same class names, constructor signatures, attributes
(`negative_sample_sharing` stays a plain mutable attribute) and method
semantics

    score_triple(h [S,W], r_id [S], t [S,W])            -> [S]
    score_heads(H [B,N,W], r_id [S], t [S,W])           -> [S, B*N] | [S, N]
    score_tails(h [S,W], r_id [S], T [B,N,W])           -> [S, B*N] | [S, N]

but every score is computed by `libbesskge_hip.so`:

    score            K2+K3  bess_score_triple_fwd
    query transform  K2+K6  bess_query_fwd     (h+r | rot(h,r) | h*r | cmul(h,r) ...)
    shared negatives K4     bess_neg_score_shared_fwd
    per-triple       K5     bess_neg_score_pertriple_fwd

There is no torch/CPU implementation behind these methods: tensors must live on
a HIP device.  Results are fp32 (accumulation is fp32 also for fp16 tables).

PairRE, TripleRE, InterHT and TranS (reference
`scoring.py:465-743, 1418-1750`) share one kernel family: their score is
`-|| U * c1(e) + V * c2(e) + R ||_p` with per-query vectors U, V, R and the
(optionally normalised) parts c1, c2 of the candidate row (`csrc/affine.hip`).
The fused BESS step runs entirely in those kernels (query transform
`[U | V | R]`, negative scoring, positive score = tail-corruption score of the one
true tail, and all backwards); the public `score_*` methods express the S-row
query transform with torch ops so that they stay differentiable through autograd.

ConvE (`scoring.py:949-1146`) keeps its query network (batch-norm, conv,
linear: dense parameters) as torch modules on the device; what is scored
against the candidates is a plain dot product over `[embedding | tail bias]`
rows with the query `[net(h, r) | 1]`, i.e. the DistMult kernels.

BoxE (`scoring.py:1149-1415`): the relation-side preprocessing (width
normalisation, box size, tanh of the box) and the positive score are torch ops
over S rows; candidates are scored by the `csrc/boxe.hip` kernels.
"""

import struct
from abc import ABC, abstractmethod
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch

from besskge import _native as nat
from besskge import ops
from besskge.embedding import (
    init_KGE_normal,
    init_KGE_uniform,
    init_uniform_norm,
    init_uniform_rotation,
    init_xavier_norm,
    initialize_entity_embedding,
    initialize_relation_embedding,
    refactor_embedding_sharding,
)
from besskge.sharding import Sharding

_Init = Union[torch.Tensor, List[Callable[..., torch.Tensor]]]


def _placement(kw: Optional[Dict[str, Any]]) -> Dict[str, Any]:
    """`device=`, `shards=`, `dtype=` of the scorer constructors (extension of the reference
    signatures, keyword only): allocate just the shard slices this process hosts
    (`[len(shards), M, W]`), directly on `device`, in `dtype` - so that a rank of a multi-GPU job
    never materialises the whole `[n_shard, M, W]` table on the host (BASELINE config 5: 128 GB
    per shard).  Defaults reproduce the reference: whole table, CPU, float32."""
    kw = dict(kw or {})
    out = dict(device=kw.pop("device", None), shards=kw.pop("shards", None), dtype=kw.pop("dtype", None))
    if kw:
        raise TypeError(f"unexpected keyword arguments {sorted(kw)}")
    return out


class BaseScoreFunction(torch.nn.Module, ABC):
    """Base class of all scoring functions."""

    #: score every query against the negatives of the whole micro-batch
    negative_sample_sharing: bool
    #: entity sharding
    sharding: Sharding
    #: entity table [n_shard, max_entity_per_shard, W]
    entity_embedding: torch.nn.Parameter
    #: relation table [n_relation, Wr]
    relation_embedding: torch.nn.Parameter

    #: kernel id of the scorer (include/besskge_hip.h BESS_TRANSE ...)
    _scorer_id: int = -1

    def kernel_desc(self) -> nat.ModelDesc:
        """Descriptor handed to the kernels (dtype follows the tables).  Built once per (dtype, widths, norm,
        switches) and handed out again: callers that set flags take a copy first (`nat.copy_desc`) - a training
        step asks for it five times."""
        rel, ent = self.relation_embedding, self.entity_embedding
        key = (rel.dtype, int(ent.shape[-1]), int(rel.shape[-1]), int(getattr(self, "scoring_norm", 0)),
               bool(getattr(self, "fp32_math", False)))
        cached = self.__dict__.get("_kernel_desc_cache")
        if cached is not None and cached[0] == key:
            return cached[1]
        d = nat.ModelDesc()
        d.scorer = self._scorer_id
        d.norm_p = key[3]
        d.dtype = nat._dtype_code(rel)
        d.width = key[1]
        d.rel_width = key[2]
        if 0 <= self._scorer_id <= nat.COMPLEX and key[4]:
            d.reserved[0] = nat.FLAG_FP32_MATH
        self.__dict__["_kernel_desc_cache"] = (key, d)
        return d

    #: TransE / RotatE with p = 1 on fp16 tables score shared negatives with packed-fp16 kernels that
    #: round the query to fp16 first (what the reference's fp16 mode does); True keeps the query in
    #: fp32 and uses the fp32 kernels
    fp32_math: bool = False

    def _table_dtype(self, x: torch.Tensor) -> torch.Tensor:
        dt = self.relation_embedding.dtype
        return x if x.dtype == dt else x.to(dt)

    # ---- hooks used by the fused BESS step (besskge.bess) -----------------------
    #: per-triple negatives of the own shard can use the segmented K9 reduction
    #: (gradient recomputed per reference from the query) instead of a [S*N, W] gradient
    supports_fused_segments = True
    #: the training forward can accumulate d loss / d query next to the scores
    #: (`bess_neg_score_pertriple_fwd_dq`)
    supports_fused_forward = True
    #: the score is bilinear in (query, candidate row): d score / d candidate = query, whatever the candidate
    #: (DistMult, ComplEx) - a backward that already has d_query never reads the candidate rows again
    bilinear_candidates = False
    #: query_fwd + triple_fwd, and their backwards, also exist as ONE launch each
    #: (`bess_query_triple_fwd / _bwd`: TransE / RotatE / DistMult / ComplEx)
    supports_fused_query_triple = True

    def query_triple_fwd(self, side: int, head: nat.RowSource, tail: nat.RowSource, rel_idx: torch.Tensor,
                         jobs: Any = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(query matrix of the side's negative-scoring problem, positive scores) of the same triples.  `jobs`: copy /
        fill jobs of the step's prologue that ride in the same launch (`nat.query_triple_fwd`)."""
        return nat.query_triple_fwd(self.kernel_desc(), side, head, tail, self.relation_embedding.data, rel_idx,
                                    jobs=jobs)

    def query_triple_bwd(self, side: int, head: nat.RowSource, tail: nat.RowSource, rel_idx: torch.Tensor,
                         d_pos: torch.Tensor, dq: torch.Tensor, d_rel: torch.Tensor
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
        """(d_head rows, d_tail rows) of positive score + query together; relation gradient added to `d_rel`."""
        return nat.query_triple_bwd(self.kernel_desc(), side, head, tail, self.relation_embedding.data, rel_idx,
                                    d_pos, dq, d_rel)

    def query_triple_bwd_parts(self, side: int, head: nat.RowSource, tail: nat.RowSource, rel_idx: torch.Tensor,
                               d_pos: torch.Tensor, dq_parts: torch.Tensor, dneg_parts: torch.Tensor, neg_idx: torch.Tensor,
                               rows_acc: Tuple[torch.Tensor, torch.Tensor, torch.Tensor], d_rel: torch.Tensor) -> None:
        """`query_triple_bwd` with every gradient row ADDED into accumulators over the tables' row spaces (`rows_acc` =
        heads', tails', candidates'), fed by the partial sums of `nat.neg_score_shared_bwd_parts`."""
        nat.query_triple_bwd_parts(self.kernel_desc(), side, head, tail, self.relation_embedding.data, rel_idx, d_pos,
                                   dq_parts, dneg_parts, neg_idx, rows_acc, d_rel)

    def dense_parameters(self) -> List[torch.nn.Parameter]:
        """Parameters besides the two embedding tables (ConvE's network); replicated like the
        relation table.  Their gradients are collected by `query_bwd` / `triple_bwd` in
        `dense_grads` (parameter -> fp32 gradient) and consumed by the training step."""
        return []

    def _collect_dense(self, grads: Any) -> None:
        store = self.__dict__.setdefault("dense_grads", {})
        for p, g in zip(self.dense_parameters(), grads):
            if g is not None:
                store[p] = g.float() if p not in store else store[p] + g.float()

    def query_fwd(self, side: int, ent: nat.RowSource, rel_idx: torch.Tensor) -> Tuple[torch.Tensor, Any]:
        """Query matrix of a negative-scoring problem and the context its backward needs."""
        return nat.query_fwd(self.kernel_desc(), side, ent, self.relation_embedding.data, rel_idx), None

    def query_bwd(self, side: int, ent: nat.RowSource, rel_idx: torch.Tensor, ctx: Any, dq: torch.Tensor,
                  d_rel: torch.Tensor) -> torch.Tensor:
        """Gradient wrt the rows of `ent`; the relation-table gradient is added to `d_rel`."""
        return nat.query_bwd(self.kernel_desc(), side, ent, self.relation_embedding.data, rel_idx, dq, d_rel)

    def triple_fwd(self, head: nat.RowSource, tail: nat.RowSource, rel_idx: torch.Tensor) -> Tuple[torch.Tensor, Any]:
        """Positive scores [S] (+ backward context)."""
        return nat.score_triple_fwd(self.kernel_desc(), head, tail, self.relation_embedding.data, rel_idx), None

    def triple_bwd(self, head: nat.RowSource, tail: nat.RowSource, rel_idx: torch.Tensor, ctx: Any,
                   d_pos: torch.Tensor, d_rel: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(d_head rows, d_tail rows); the relation-table gradient is added to `d_rel`."""
        return nat.score_triple_bwd(self.kernel_desc(), head, tail, self.relation_embedding.data, rel_idx, d_pos,
                                    d_rel)

    def score_triple(
        self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor
    ) -> torch.Tensor:
        """Score (h, r, t) triples.

        :param head_emb: [batch, W] head embeddings.
        :param relation_id: [batch] relation IDs.
        :param tail_emb: [batch, W] tail embeddings.
        :return: [batch] scores.
        """
        return ops.ScoreTriple.apply(
            self.kernel_desc(),
            self._table_dtype(head_emb),
            self.relation_embedding,
            relation_id,
            self._table_dtype(tail_emb),
        )

    def score_heads(
        self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor
    ) -> torch.Tensor:
        """Score candidate heads against fixed (r, t) queries.

        :param head_emb: [B, n_heads, W], B = 1 or batch.
        :param relation_id: [batch].
        :param tail_emb: [batch, W].
        :return: [batch, B * n_heads] with negative sample sharing, else
            [batch, n_heads].
        """
        return ops.ScoreNegatives.apply(
            self.kernel_desc(),
            nat.CORRUPT_HEAD,
            bool(self.negative_sample_sharing),
            self._table_dtype(tail_emb),
            self.relation_embedding,
            relation_id,
            self._table_dtype(head_emb),
        )

    def score_tails(
        self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor
    ) -> torch.Tensor:
        """Score candidate tails against fixed (h, r) queries.

        :param head_emb: [batch, W].
        :param relation_id: [batch].
        :param tail_emb: [B, n_tails, W], B = 1 or batch.
        :return: [batch, B * n_tails] with negative sample sharing, else
            [batch, n_tails].
        """
        return ops.ScoreNegatives.apply(
            self.kernel_desc(),
            nat.CORRUPT_TAIL,
            bool(self.negative_sample_sharing),
            self._table_dtype(head_emb),
            self.relation_embedding,
            relation_id,
            self._table_dtype(tail_emb),
        )

    def forward(
        self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor
    ) -> torch.Tensor:
        """Same as :meth:`score_triple`."""
        return self.score_triple(head_emb, relation_id, tail_emb)

    def update_sharding(self, new_sharding: Sharding) -> None:
        """Re-shard the entity table for a different :class:`Sharding`."""
        self.entity_embedding = refactor_embedding_sharding(
            entity_embedding=self.entity_embedding,
            old_sharding=self.sharding,
            new_sharding=new_sharding,
        )
        self.sharding = new_sharding

    def _allocate(
        self,
        sharding: Sharding,
        n_relation_type: int,
        inverse_relations: bool,
        entity_initializer: _Init,
        relation_initializer: _Init,
        entity_width: int,
        relation_width: int,
        what: str,
        placement: Optional[Dict[str, Any]] = None,
    ) -> None:
        self.sharding = sharding
        pl = _placement(placement)
        self.entity_embedding = initialize_entity_embedding(
            sharding, entity_initializer, [entity_width], **pl
        )
        self.relation_embedding = initialize_relation_embedding(
            n_relation_type, inverse_relations, relation_initializer, [relation_width],
            device=pl["device"], dtype=pl["dtype"]
        )
        assert (
            self.entity_embedding.shape[-1] == entity_width
            and self.relation_embedding.shape[-1] == relation_width
        ), what


class DistanceBasedScoreFunction(BaseScoreFunction, ABC):
    """Scorers of the form -||query - entity||_p."""

    #: norms the scorer's kernels implement; None = any integer p >= 1 (TransE, RotatE: p = 1 and p = 2 have their
    #: own instruction sequences, every other p goes through powf - reference scoring.py:174 takes any p)
    supported_norms: Optional[Tuple[int, ...]] = None

    def __init__(self, negative_sample_sharing: bool, scoring_norm: int) -> None:
        """
        :param negative_sample_sharing: see :class:`BaseScoreFunction`.
        :param scoring_norm: p of the p-norm.
        """
        super().__init__()
        if int(scoring_norm) != scoring_norm or scoring_norm < 1:
            raise ValueError(f"scoring_norm must be an integer p >= 1, got {scoring_norm!r}")
        if self.supported_norms is not None and scoring_norm not in self.supported_norms:
            raise ValueError(f"the HIP kernels of {type(self).__name__} implement scoring_norm in {self.supported_norms}")
        self.negative_sample_sharing = negative_sample_sharing
        self.scoring_norm = int(scoring_norm)

    def reduce_embedding(self, v: torch.Tensor) -> torch.Tensor:
        """p-norm over the embedding dimension (reference scoring.py:163-174)."""
        return torch.norm(v, p=self.scoring_norm, dim=-1)

    def broadcasted_distance(self, v1: torch.Tensor, v2: torch.Tensor) -> torch.Tensor:
        """Distances of queries v1 [batch, W] to candidates v2 [B, n_neg, W]: [batch, B * n_neg]
        with negative sample sharing, [batch, n_neg] otherwise (reference scoring.py:176-200);
        the HIP negative-scoring kernels with an identity query transform."""
        return -ops.ReduceNegatives.apply(_plain_desc(nat.TRANSE, self.scoring_norm, v2), bool(self.negative_sample_sharing),
                                          v1.float().contiguous(), v2)


def _plain_desc(scorer: int, norm_p: int, rows: torch.Tensor) -> nat.ModelDesc:
    """Descriptor of a bare reduction (`-||q - e||_p` or `<q, e>`) over rows of this dtype / width."""
    d = nat.ModelDesc()
    d.scorer, d.norm_p = scorer, int(norm_p)
    d.dtype = nat._dtype_code(rows)
    d.width = d.rel_width = int(rows.shape[-1])
    return d


class MatrixDecompositionScoreFunction(BaseScoreFunction, ABC):
    """Scorers of the form <query, entity>."""

    def __init__(self, negative_sample_sharing: bool) -> None:
        super().__init__()
        self.negative_sample_sharing = negative_sample_sharing

    def reduce_embedding(self, v: torch.Tensor) -> torch.Tensor:
        """Sum over the embedding dimension (reference scoring.py:219-229)."""
        return torch.sum(v, dim=-1)

    def broadcasted_dot_product(self, v1: torch.Tensor, v2: torch.Tensor) -> torch.Tensor:
        """Dot products of queries v1 [batch, W] with candidates v2 [B, n_neg, W]: [batch, B * n_neg]
        with negative sample sharing, [batch, n_neg] otherwise (reference scoring.py:231-255)."""
        return ops.ReduceNegatives.apply(_plain_desc(nat.DISTMULT, 0, v2), bool(self.negative_sample_sharing),
                                         v1.float().contiguous(), v2)


class TransE(DistanceBasedScoreFunction):
    """TransE: -|| h + r - t ||_p  (reference scoring.py:258-354)."""

    _scorer_id = nat.TRANSE

    def __init__(
        self,
        negative_sample_sharing: bool,
        scoring_norm: int,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        entity_initializer: _Init = [init_KGE_uniform],
        relation_initializer: _Init = [init_KGE_uniform],
        inverse_relations: bool = False,
        **placement: Any,
    ) -> None:
        super().__init__(negative_sample_sharing, scoring_norm)
        self._allocate(
            sharding, n_relation_type, inverse_relations, entity_initializer,
            relation_initializer, embedding_size, embedding_size,
            "TransE requires `embedding_size` embedding parameters"
            " for each entity and relation",
            placement=placement,
        )
        self.embedding_size = embedding_size


class RotatE(DistanceBasedScoreFunction):
    """RotatE: -|| h * e^{i r} - t ||_p over the [re | im] components
    (reference scoring.py:357-462; relation rows are phases)."""

    _scorer_id = nat.ROTATE

    def __init__(
        self,
        negative_sample_sharing: bool,
        scoring_norm: int,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        entity_initializer: _Init = [init_KGE_uniform],
        relation_initializer: _Init = [init_uniform_rotation],
        inverse_relations: bool = False,
        **placement: Any,
    ) -> None:
        super().__init__(negative_sample_sharing, scoring_norm)
        self._allocate(
            sharding, n_relation_type, inverse_relations, entity_initializer,
            relation_initializer, 2 * embedding_size, embedding_size,
            "RotatE requires `2*embedding_size` embedding parameters for each entity"
            " and `embedding_size` embedding parameters for each relation",
            placement=placement,
        )
        self.embedding_size = embedding_size


class DistMult(MatrixDecompositionScoreFunction):
    """DistMult: sum(h * r * t)  (reference scoring.py:746-837)."""

    _scorer_id = nat.DISTMULT
    bilinear_candidates = True

    def __init__(
        self,
        negative_sample_sharing: bool,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        entity_initializer: _Init = [init_KGE_uniform],
        relation_initializer: _Init = [init_KGE_uniform],
        inverse_relations: bool = False,
        **placement: Any,
    ) -> None:
        super().__init__(negative_sample_sharing)
        self._allocate(
            sharding, n_relation_type, inverse_relations, entity_initializer,
            relation_initializer, embedding_size, embedding_size,
            "DistMult requires `embedding_size` embedding parameters"
            " for each entity and relation",
            placement=placement,
        )
        self.embedding_size = embedding_size


class ComplEx(MatrixDecompositionScoreFunction):
    """ComplEx: Re<h * r, conj t> on [re | im] rows (reference scoring.py:840-946)."""

    _scorer_id = nat.COMPLEX
    bilinear_candidates = True

    def __init__(
        self,
        negative_sample_sharing: bool,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        entity_initializer: _Init = [init_KGE_normal],
        relation_initializer: _Init = [init_KGE_normal],
        inverse_relations: bool = False,
        **placement: Any,
    ) -> None:
        super().__init__(negative_sample_sharing)
        self._allocate(
            sharding, n_relation_type, inverse_relations, entity_initializer,
            relation_initializer, 2 * embedding_size, 2 * embedding_size,
            "ComplEx requires `2*embedding_size` embedding parameters"
            " for each entity and relation",
            placement=placement,
        )
        self.embedding_size = embedding_size


# --------------------------------------------------------------------------- #
# PairRE / TripleRE / InterHT / TranS: affine in the candidate
def _rows_of(src: nat.RowSource) -> torch.Tensor:
    return src.base if src.idx is None else nat.gather_rows(src.base, src.idx)


class _TorchQueryHooks:
    """Hooks of the fused step for scorers whose S-row work (query transform,
    positive score) is written with torch ops: autograd over S rows, the
    gradient of the embedding rows / relation table / dense parameters handed
    back to the step."""

    supports_fused_segments = False
    supports_fused_forward = False
    supports_fused_query_triple = False

    def _query_torch(self, side: int, rows: torch.Tensor, rel: torch.Tensor, rel_idx: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def _triple_torch(self, h: torch.Tensor, rel: torch.Tensor, rel_idx: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def query_fwd(self, side: int, ent: nat.RowSource, rel_idx: torch.Tensor) -> Tuple[torch.Tensor, Any]:
        track = torch.is_grad_enabled()  # inference runs under no_grad: no graph is kept
        rows = _rows_of(ent).detach().float().requires_grad_(track)
        rel = self.relation_embedding.detach().float().requires_grad_(track)  # type: ignore[attr-defined]
        q = self._query_torch(side, rows, rel, rel_idx)
        return q.detach().contiguous(), (rows, rel, q)

    def query_bwd(self, side: int, ent: nat.RowSource, rel_idx: torch.Tensor, ctx: Any, dq: torch.Tensor,
                  d_rel: torch.Tensor) -> torch.Tensor:
        rows, rel, q = ctx
        dense = self.dense_parameters()  # type: ignore[attr-defined]
        grads = torch.autograd.grad(q, [rows, rel] + dense, dq, allow_unused=True)
        d_rel += grads[1]
        self._collect_dense(grads[2:])  # type: ignore[attr-defined]
        return grads[0].contiguous()

    def triple_fwd(self, head: nat.RowSource, tail: nat.RowSource, rel_idx: torch.Tensor) -> Tuple[torch.Tensor, Any]:
        track = torch.is_grad_enabled()
        h = _rows_of(head).detach().float().requires_grad_(track)
        t = _rows_of(tail).detach().float().requires_grad_(track)
        rel = self.relation_embedding.detach().float().requires_grad_(track)  # type: ignore[attr-defined]
        sc = self._triple_torch(h, rel, rel_idx, t)
        return sc.detach().contiguous(), (h, t, rel, sc)

    def triple_bwd(self, head: nat.RowSource, tail: nat.RowSource, rel_idx: torch.Tensor, ctx: Any,
                   d_pos: torch.Tensor, d_rel: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        h, t, rel, sc = ctx
        dense = self.dense_parameters()  # type: ignore[attr-defined]
        grads = torch.autograd.grad(sc, [h, t, rel] + dense, d_pos, allow_unused=True)
        d_rel += grads[2]
        self._collect_dense(grads[3:])  # type: ignore[attr-defined]
        return grads[0].contiguous(), grads[1].contiguous()


class _AffineScoreFunction(_TorchQueryHooks, DistanceBasedScoreFunction, ABC):
    """`-|| U * c1 + V * c2 + R ||_p` scorers; see the module docstring."""

    _scorer_id = nat.AFFINE
    supports_fused_segments = True  # csrc/affine.hip: k_aff_grad_segments
    #: d-wide parts of an entity row (1 | 2)
    _n_part: int = 1
    normalize: bool
    embedding_size: int

    #: member of the family as csrc/affine.hip numbers them (0 PairRE, 1 TripleRE, 2 InterHT, 3 TranS)
    _member: int = 0

    def _constant(self) -> float:
        """The model's additive constant (TripleRE: u, InterHT / TranS: offset)."""
        return 0.0

    def _host_constant(self, buf: torch.Tensor) -> float:
        """Value of a one-element constant buffer as a Python float, read from the device ONCE per version of the
        buffer: `float(device tensor)` is a host sync on every step - and an error inside a hipGraph capture
        (`tests/test_multi_gpu.py`, recorded TripleRE / InterHT / TranS steps)."""
        key = (buf.data_ptr(), buf._version, buf.device)
        cached = self.__dict__.get("_host_constant_cache")
        if cached is None or cached[0] != key:
            cached = (key, float(buf))
            self.__dict__["_host_constant_cache"] = cached
        return cached[1]

    def kernel_desc(self) -> nat.ModelDesc:
        d = super().kernel_desc()
        d.reserved[0] = self._n_part
        d.reserved[1] = int(bool(self.normalize)) | (self._member << 8)
        d.reserved[2] = struct.unpack("i", struct.pack("f", self._constant()))[0]
        return d

    # hooks of the fused step: HIP query transform (k_aff_query_fwd / bwd); the positive score is
    # the tail-corruption score of the one true tail (the same vector up to sign inside the norm)
    supports_fused_forward = False

    def query_fwd(self, side: int, ent: nat.RowSource, rel_idx: torch.Tensor) -> Tuple[torch.Tensor, Any]:
        return nat.query_fwd(self.kernel_desc(), side, ent, self.relation_embedding.data, rel_idx), None

    def query_bwd(self, side: int, ent: nat.RowSource, rel_idx: torch.Tensor, ctx: Any, dq: torch.Tensor,
                  d_rel: torch.Tensor) -> torch.Tensor:
        return nat.query_bwd(self.kernel_desc(), side, ent, self.relation_embedding.data, rel_idx, dq, d_rel)

    def triple_fwd(self, head: nat.RowSource, tail: nat.RowSource, rel_idx: torch.Tensor) -> Tuple[torch.Tensor, Any]:
        desc = self.kernel_desc()
        q = nat.query_fwd(desc, nat.CORRUPT_TAIL, head, self.relation_embedding.data, rel_idx)
        return nat.neg_score_pertriple_fwd(desc, q, tail, 1).reshape(-1), q

    def triple_bwd(self, head: nat.RowSource, tail: nat.RowSource, rel_idx: torch.Tensor, ctx: Any,
                   d_pos: torch.Tensor, d_rel: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        desc = self.kernel_desc()
        dq, dt = nat.neg_score_pertriple_bwd(desc, ctx, tail, 1, d_pos.reshape(-1, 1).contiguous())
        dh = nat.query_bwd(desc, nat.CORRUPT_TAIL, head, self.relation_embedding.data, rel_idx, dq, d_rel)
        return dh, dt

    def _parts(self, x: torch.Tensor) -> List[torch.Tensor]:
        parts = list(torch.split(x.float(), self.embedding_size, dim=-1))
        if self.normalize:
            parts = [torch.nn.functional.normalize(p, p=2, dim=-1) for p in parts]
        return parts

    def _rel(self, relation_id: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
        return torch.index_select(table, 0, relation_id.reshape(-1).long()).float()

    @abstractmethod
    def _delta(self, h: List[torch.Tensor], rel: torch.Tensor, t: List[torch.Tensor]) -> torch.Tensor:
        """The vector whose p-norm is the (negated) score of a triple."""

    @abstractmethod
    def _uvr(self, side: int, kept: List[torch.Tensor], rel: torch.Tensor) -> List[torch.Tensor]:
        """[U, (V,) R] of the queries whose head (side = CORRUPT_HEAD) or tail is replaced."""

    def _score_norm(self, delta: torch.Tensor) -> torch.Tensor:
        return -torch.norm(delta, p=self.scoring_norm, dim=-1)

    # public API (differentiable through autograd + ops.ReduceNegatives)
    def score_triple(self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
        return self._score_norm(self._delta(self._parts(head_emb), self._rel(relation_id, self.relation_embedding),
                                            self._parts(tail_emb)))

    def _score_candidates(self, side: int, kept: torch.Tensor, relation_id: torch.Tensor,
                          cand: torch.Tensor) -> torch.Tensor:
        q = torch.cat(self._uvr(side, self._parts(kept), self._rel(relation_id, self.relation_embedding)), dim=-1)
        return ops.ReduceNegatives.apply(self.kernel_desc(), bool(self.negative_sample_sharing), q.contiguous(),
                                         self._table_dtype(cand))

    def score_heads(self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
        return self._score_candidates(nat.CORRUPT_HEAD, tail_emb, relation_id, head_emb)

    def score_tails(self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
        return self._score_candidates(nat.CORRUPT_TAIL, head_emb, relation_id, tail_emb)

    def _query_torch(self, side: int, rows: torch.Tensor, rel: torch.Tensor, rel_idx: torch.Tensor) -> torch.Tensor:
        return torch.cat(self._uvr(side, self._parts(rows), self._rel(rel_idx, rel)), dim=-1)

    def _triple_torch(self, h: torch.Tensor, rel: torch.Tensor, rel_idx: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        return self._score_norm(self._delta(self._parts(h), self._rel(rel_idx, rel), self._parts(t)))


class PairRE(_AffineScoreFunction):
    """PairRE: -|| h^ * r_h - t^ * r_t ||_p  (reference scoring.py:465-593)."""

    _n_part = 1
    _member = 0

    def __init__(
        self,
        negative_sample_sharing: bool,
        scoring_norm: int,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        entity_initializer: _Init = [init_KGE_uniform],
        relation_initializer: _Init = [init_KGE_uniform],
        normalize_entities: bool = True,
        inverse_relations: bool = False,
        **placement: Any,
    ) -> None:
        super().__init__(negative_sample_sharing, scoring_norm)
        self.sharding = sharding
        self.normalize = normalize_entities
        if isinstance(relation_initializer, list):
            relation_initializer = 2 * relation_initializer
        self.entity_embedding = initialize_entity_embedding(sharding, entity_initializer, [embedding_size], **_placement(placement))
        # [r_h | r_t]: projections of heads and tails
        self.relation_embedding = initialize_relation_embedding(
            n_relation_type, inverse_relations, relation_initializer, [embedding_size, embedding_size],
            device=_placement(placement)["device"], dtype=_placement(placement)["dtype"])
        assert 2 * self.entity_embedding.shape[-1] == self.relation_embedding.shape[-1] == 2 * embedding_size, (
            "PairRE requires `embedding_size` embedding parameters for each entity"
            " and `2*embedding_size` embedding parameters for each relation")
        self.embedding_size = embedding_size

    def _delta(self, h: List[torch.Tensor], rel: torch.Tensor, t: List[torch.Tensor]) -> torch.Tensor:
        r_h, r_t = torch.split(rel, self.embedding_size, dim=-1)
        return h[0] * r_h - t[0] * r_t

    def _uvr(self, side: int, kept: List[torch.Tensor], rel: torch.Tensor) -> List[torch.Tensor]:
        r_h, r_t = torch.split(rel, self.embedding_size, dim=-1)
        if side == nat.CORRUPT_TAIL:  # t^ r_t - h^ r_h
            return [r_t, -(kept[0] * r_h)]
        return [r_h, -(kept[0] * r_t)]  # h^ r_h - t^ r_t


class TripleRE(_AffineScoreFunction):
    """TripleRE: -|| h^ * (r_h + u) - t^ * (r_t + u) + r_m ||_p  (reference scoring.py:596-743)."""

    _n_part = 1
    _member = 1

    def _constant(self) -> float:
        return self._host_constant(self.rel_u) if self.use_v2 else 0.0

    def __init__(
        self,
        negative_sample_sharing: bool,
        scoring_norm: int,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        entity_initializer: _Init = [init_KGE_uniform],
        relation_initializer: _Init = [init_KGE_uniform],
        normalize_entities: bool = True,
        u: float = 0.0,
        inverse_relations: bool = False,
        **placement: Any,
    ) -> None:
        super().__init__(negative_sample_sharing, scoring_norm)
        self.sharding = sharding
        self.normalize = normalize_entities
        if isinstance(relation_initializer, list):
            relation_initializer = 3 * relation_initializer
        self.entity_embedding = initialize_entity_embedding(sharding, entity_initializer, [embedding_size], **_placement(placement))
        # [r_h | r_m | r_t]: head projection, translation, tail projection
        self.relation_embedding = initialize_relation_embedding(
            n_relation_type, inverse_relations, relation_initializer, [embedding_size] * 3,
            device=_placement(placement)["device"], dtype=_placement(placement)["dtype"])
        assert 3 * self.entity_embedding.shape[-1] == self.relation_embedding.shape[-1] == 3 * embedding_size, (
            "TripleRE requires `embedding_size` embedding parameters for each entity"
            " and `3*embedding_size` embedding parameters for each relation")
        self.embedding_size = embedding_size
        self.use_v2 = u > 0.0
        self.register_buffer("rel_u", torch.tensor([u], dtype=torch.float32))

    def _split(self, rel: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        r_h, r_m, r_t = torch.split(rel, self.embedding_size, dim=-1)
        if self.use_v2:
            u = self.rel_u.to(rel.device)
            r_h, r_t = r_h + u, r_t + u
        return r_h, r_m, r_t

    def _delta(self, h: List[torch.Tensor], rel: torch.Tensor, t: List[torch.Tensor]) -> torch.Tensor:
        r_h, r_m, r_t = self._split(rel)
        return h[0] * r_h - t[0] * r_t + r_m

    def _uvr(self, side: int, kept: List[torch.Tensor], rel: torch.Tensor) -> List[torch.Tensor]:
        r_h, r_m, r_t = self._split(rel)
        if side == nat.CORRUPT_TAIL:  # t^ r_t - (h^ r_h + r_m)
            return [r_t, -(kept[0] * r_h + r_m)]
        return [r_h, -(kept[0] * r_t - r_m)]  # h^ r_h - (t^ r_t - r_m)


class InterHT(_AffineScoreFunction):
    """InterHT: -|| h^ * (t^_aux + o) + r - t^ * (h^_aux + o) ||_p  (reference scoring.py:1418-1572)."""

    _n_part = 2
    _member = 2

    def _constant(self) -> float:
        return self._host_constant(self.offset)

    def __init__(
        self,
        negative_sample_sharing: bool,
        scoring_norm: int,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        entity_initializer: _Init = [init_KGE_uniform],
        relation_initializer: _Init = [init_KGE_uniform],
        normalize_entities: bool = True,
        offset: float = 1.0,
        inverse_relations: bool = False,
        **placement: Any,
    ) -> None:
        super().__init__(negative_sample_sharing, scoring_norm)
        self.sharding = sharding
        self.normalize = normalize_entities
        if isinstance(entity_initializer, list):
            entity_initializer = 2 * entity_initializer
        # [main | auxiliary]
        self.entity_embedding = initialize_entity_embedding(sharding, entity_initializer, [embedding_size] * 2, **_placement(placement))
        self.relation_embedding = initialize_relation_embedding(
            n_relation_type, inverse_relations, relation_initializer, [embedding_size],
            device=_placement(placement)["device"], dtype=_placement(placement)["dtype"])
        assert self.entity_embedding.shape[-1] == 2 * self.relation_embedding.shape[-1] == 2 * embedding_size, (
            "InterHT requires `2*embedding_size` embedding parameters for each entity"
            " and `embedding_size` embedding parameters for each relation")
        self.embedding_size = embedding_size
        self.register_buffer("offset", torch.tensor([offset], dtype=torch.float32))

    def _delta(self, h: List[torch.Tensor], rel: torch.Tensor, t: List[torch.Tensor]) -> torch.Tensor:
        o = self.offset.to(rel.device)
        return h[0] * (t[1] + o) + rel - t[0] * (h[1] + o)

    def _uvr(self, side: int, kept: List[torch.Tensor], rel: torch.Tensor) -> List[torch.Tensor]:
        o = self.offset.to(rel.device)
        main, aux = kept
        if side == nat.CORRUPT_TAIL:  # candidates t = (c1 main, c2 aux):  h_m (c2 + o) + r - c1 (h_a + o)
            return [-(aux + o), main, rel + main * o]
        return [aux + o, -main, rel - main * o]  # candidates h:  c1 (t_a + o) + r - t_m (c2 + o)


class TranS(_AffineScoreFunction):
    """TranS: -|| h^ * (t~ + o + r_bar) - t^ * (h~ + o - r_hat) + r ||_p  (reference scoring.py:1575-1750)."""

    _n_part = 2
    _member = 3

    def _constant(self) -> float:
        return self._host_constant(self.offset)

    def __init__(
        self,
        negative_sample_sharing: bool,
        scoring_norm: int,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        entity_initializer: _Init = [init_KGE_uniform],
        relation_initializer: _Init = [init_KGE_uniform],
        normalize_entities: bool = True,
        offset: float = 1.0,
        inverse_relations: bool = False,
        **placement: Any,
    ) -> None:
        super().__init__(negative_sample_sharing, scoring_norm)
        self.sharding = sharding
        self.normalize = normalize_entities
        if isinstance(entity_initializer, list):
            entity_initializer = 2 * entity_initializer
        # [main | tilde]
        self.entity_embedding = initialize_entity_embedding(sharding, entity_initializer, [embedding_size] * 2, **_placement(placement))
        if isinstance(relation_initializer, list):
            relation_initializer = 3 * relation_initializer
        # [r | r_bar | r_hat]
        self.relation_embedding = initialize_relation_embedding(
            n_relation_type, inverse_relations, relation_initializer, [embedding_size] * 3,
            device=_placement(placement)["device"], dtype=_placement(placement)["dtype"])
        assert self.entity_embedding.shape[-1] / 2 == self.relation_embedding.shape[-1] / 3 == embedding_size, (
            "TranS requires `2*embedding_size` embedding parameters for each entity"
            " and `3*embedding_size` embedding parameters for each relation")
        self.embedding_size = embedding_size
        self.register_buffer("offset", torch.tensor([offset], dtype=torch.float32))

    def _delta(self, h: List[torch.Tensor], rel: torch.Tensor, t: List[torch.Tensor]) -> torch.Tensor:
        o = self.offset.to(rel.device)
        r, r_bar, r_hat = torch.split(rel, self.embedding_size, dim=-1)
        return h[0] * (t[1] + o + r_bar) - t[0] * (h[1] + o - r_hat) + r

    def _uvr(self, side: int, kept: List[torch.Tensor], rel: torch.Tensor) -> List[torch.Tensor]:
        o = self.offset.to(rel.device)
        r, r_bar, r_hat = torch.split(rel, self.embedding_size, dim=-1)
        main, tilde = kept
        if side == nat.CORRUPT_TAIL:  # h_m (c2 + o + r_bar) - c1 (h~ + o - r_hat) + r
            return [-(tilde + o - r_hat), main, r + main * (o + r_bar)]
        return [tilde + o + r_bar, -main, r - main * (o - r_hat)]  # c1 (t~ + o + r_bar) - t_m (c2 + o - r_hat) + r


class ConvE(_TorchQueryHooks, MatrixDecompositionScoreFunction):
    """ConvE: <net(h, r), t> + b_t, scores not passed through a sigmoid; only
    tails can be corrupted (reference scoring.py:949-1146)."""

    _scorer_id = nat.DISTMULT  # candidates see a dot product over [embedding | bias] rows
    supports_fused_segments = True  # ... so their K9 is the dot-product segmented reduction

    def __init__(
        self,
        negative_sample_sharing: bool,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        embedding_height: int,
        embedding_width: int,
        entity_initializer: _Init = [init_xavier_norm, torch.nn.init.zeros_],
        relation_initializer: _Init = [init_xavier_norm],
        inverse_relations: bool = True,
        input_channels: int = 1,
        output_channels: int = 32,
        kernel_height: int = 3,
        kernel_width: int = 3,
        input_dropout: float = 0.2,
        feature_map_dropout: float = 0.2,
        hidden_dropout: float = 0.3,
        batch_normalization: bool = True,
        **placement: Any,
    ) -> None:
        super().__init__(negative_sample_sharing)
        self.sharding = sharding
        if input_channels * embedding_width * embedding_height != embedding_size:
            raise ValueError("`embedding_size` needs to be equal to"
                             " `input_channels * embedding_width * embedding_height`")
        # [embedding | tail bias]
        self.entity_embedding = initialize_entity_embedding(sharding, entity_initializer, [embedding_size, 1], **_placement(placement))
        self.relation_embedding = initialize_relation_embedding(
            n_relation_type, inverse_relations, relation_initializer, [embedding_size],
            device=_placement(placement)["device"], dtype=_placement(placement)["dtype"])
        assert self.entity_embedding.shape[-1] - 1 == self.relation_embedding.shape[-1] == embedding_size, (
            "ConvE requires `embedding_size + 1` embedding parameters for each entity"
            " and `embedding_size` embedding parameters for each relation")
        self.embedding_size = embedding_size
        self.inp_channels = input_channels
        self.emb_h = embedding_height
        self.emb_w = embedding_width
        conv_layers = [
            torch.nn.Dropout(input_dropout),
            torch.nn.Conv2d(in_channels=input_channels, out_channels=output_channels,
                            kernel_size=(kernel_height, kernel_width)),
            torch.nn.ReLU(),
            torch.nn.Dropout2d(feature_map_dropout),
        ]
        fc_layers = [
            torch.nn.Linear(output_channels * (2 * embedding_height - kernel_height + 1)
                            * (embedding_width - kernel_width + 1), embedding_size),
            torch.nn.Dropout(hidden_dropout),
            torch.nn.ReLU(),
        ]
        if batch_normalization:
            conv_layers.insert(0, torch.nn.BatchNorm2d(input_channels))
            conv_layers.insert(3, torch.nn.BatchNorm2d(output_channels))
            fc_layers.insert(2, torch.nn.BatchNorm1d(embedding_size))
        self.conv_layers = torch.nn.Sequential(*conv_layers)
        self.fc_layers = torch.nn.Sequential(*fc_layers)

    def kernel_desc(self) -> nat.ModelDesc:
        d = super().kernel_desc()
        d.rel_width = d.width  # the dot-product kernels only see [embedding | bias] rows
        return d

    def dense_parameters(self) -> List[torch.nn.Parameter]:
        return list(self.conv_layers.parameters()) + list(self.fc_layers.parameters())

    def _net(self, head: torch.Tensor, rel: torch.Tensor) -> torch.Tensor:
        shape = (-1, self.inp_channels, self.emb_h, self.emb_w)
        hr = torch.cat([head[..., :-1].reshape(shape), rel.reshape(shape)], dim=-2)
        return self.fc_layers(self.conv_layers(hr).flatten(start_dim=1))

    def _query_torch(self, side: int, rows: torch.Tensor, rel: torch.Tensor, rel_idx: torch.Tensor) -> torch.Tensor:
        if side != nat.CORRUPT_TAIL:
            raise NotImplementedError("ConvE should not be used with head corruption")
        q = self._net(rows.float(), torch.index_select(rel, 0, rel_idx.reshape(-1).long()).float())
        return torch.cat([q, torch.ones_like(q[:, :1])], dim=-1)  # x 1 picks up the tail bias

    def _triple_torch(self, h: torch.Tensor, rel: torch.Tensor, rel_idx: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        q = self._net(h.float(), torch.index_select(rel, 0, rel_idx.reshape(-1).long()).float())
        t = t.float()
        return torch.sum(q * t[..., :-1], dim=-1) + t[..., -1]

    def score_triple(self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
        return self._triple_torch(head_emb, self.relation_embedding, relation_id, tail_emb)

    def score_heads(self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("ConvE should not be used with head corruption")

    def score_tails(self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
        q = self._query_torch(nat.CORRUPT_TAIL, head_emb, self.relation_embedding, relation_id)
        return ops.ReduceNegatives.apply(self.kernel_desc(), bool(self.negative_sample_sharing), q.contiguous(),
                                         self._table_dtype(tail_emb))


class BoxE(_TorchQueryHooks, DistanceBasedScoreFunction):
    """BoxE (reference scoring.py:1149-1415): entities are [base | bump]; the head is
    bumped by the tail's bump and vice versa, and each bumped point is scored by a
    piecewise distance to its relation box."""


    _scorer_id = nat.BOXE

    supports_fused_segments = True  # csrc/boxe.hip: k_box_grad_segments

    def __init__(
        self,
        negative_sample_sharing: bool,
        scoring_norm: int,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        entity_initializer: _Init = [torch.nn.init.uniform_],
        relation_initializer: _Init = [torch.nn.init.uniform_, init_uniform_norm],
        apply_tanh: bool = True,
        dist_func_per_dim: bool = True,
        eps: float = 1e-6,
        inverse_relations: bool = False,
        **placement: Any,
    ) -> None:
        super().__init__(negative_sample_sharing, scoring_norm)
        self.apply_tanh = apply_tanh
        self.dist_func_per_dim = dist_func_per_dim
        self.eps = eps
        self.sharding = sharding
        if isinstance(entity_initializer, list):
            entity_initializer = 2 * entity_initializer
        if isinstance(relation_initializer, list):
            relation_initializer = 4 * [relation_initializer[0]] + 2 * [relation_initializer[1]]
        # [base | bump]
        self.entity_embedding = initialize_entity_embedding(sharding, entity_initializer, [embedding_size] * 2, **_placement(placement))
        # [head centre | tail centre | head width | tail width | head size, tail size]
        self.relation_embedding = initialize_relation_embedding(
            n_relation_type, inverse_relations, relation_initializer, [embedding_size] * 4 + [1, 1],
            device=_placement(placement)["device"], dtype=_placement(placement)["dtype"])
        assert 2 * self.entity_embedding.shape[-1] == self.relation_embedding.shape[-1] - 2 == 4 * embedding_size, (
            "BoxE requires `2*embedding_size` embedding parameters for each entity"
            " and `4*embedding_size + 2` embedding parameters for each relation")
        self.embedding_size = embedding_size

    def kernel_desc(self) -> nat.ModelDesc:
        d = super().kernel_desc()
        d.reserved[0] = int(bool(self.apply_tanh)) | (int(bool(self.dist_func_per_dim)) << 1)
        return d

    def _boxes(self, rel: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(centre, half width) [S, 2 (head box, tail box), d] after the reference's normalisation."""
        d = self.embedding_size
        center = rel[:, : 2 * d].reshape(-1, 2, d)
        width = rel[:, 2 * d: 4 * d].reshape(-1, 2, d).abs()
        size = rel[:, 4 * d:].reshape(-1, 2, 1)
        geo_mean = torch.exp(torch.mean(torch.log(torch.clamp(width, min=self.eps)), dim=-1, keepdim=True))
        width = width / torch.clamp(geo_mean, min=self.eps) * (1.0 + torch.nn.functional.elu(size))
        if self.apply_tanh:
            low = torch.tanh(center - 0.5 * width)
            up = torch.tanh(low + width)
            center, width = 0.5 * (low + up), up - low
        return center, 0.5 * width

    def _box_distance(self, point: torch.Tensor, center: torch.Tensor, half: torch.Tensor) -> torch.Tensor:
        """-(||f_head|| + ||f_tail||) for bumped points [..., 2, d]."""
        if self.apply_tanh:
            point = torch.tanh(point)
        dist = (point - center).abs()
        grow = 1.0 + 2.0 * half
        inside = dist <= half
        if not self.dist_func_per_dim:
            inside = inside.all(dim=-1, keepdim=True)
        f = torch.where(inside, dist / grow, dist * grow - half * (grow - 1.0 / grow))
        return -torch.norm(f, p=self.scoring_norm, dim=-1).sum(-1)

    def boxe_score(self, bumped_ht: torch.Tensor, center_ht: torch.Tensor, width_ht: torch.Tensor,
                   box_size: torch.Tensor) -> torch.Tensor:
        """BoxE score of bumped points [..., 2, d] against boxes given by raw centres, widths
        [..., 2, d] and sizes [..., 2], with broadcasting (reference scoring.py:1250-1340)."""
        rel = torch.cat([center_ht.flatten(start_dim=-2), width_ht.flatten(start_dim=-2), box_size], dim=-1)
        lead = rel.shape[:-1]
        center, half = self._boxes(rel.reshape(-1, rel.shape[-1]).float())
        return self._box_distance(bumped_ht.float(), center.reshape(*lead, 2, -1), half.reshape(*lead, 2, -1))

    def _triple_torch(self, h: torch.Tensor, rel: torch.Tensor, rel_idx: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        d = self.embedding_size
        center, half = self._boxes(torch.index_select(rel, 0, rel_idx.reshape(-1).long()).float())
        # head box sees h_base + t_bump, tail box sees t_base + h_bump
        point = h.float().reshape(-1, 2, d) + t.float().reshape(-1, 2, d).flip(-2)
        return self._box_distance(point, center, half)

    def _query_torch(self, side: int, rows: torch.Tensor, rel: torch.Tensor, rel_idx: torch.Tensor) -> torch.Tensor:
        d = self.embedding_size
        center, half = self._boxes(torch.index_select(rel, 0, rel_idx.reshape(-1).long()).float())
        kept = rows.float().reshape(-1, 2, d)
        # candidate part 0 (base) is shifted by the kept bump, part 1 (bump) by the kept base; the
        # base of a candidate tail meets the tail box, the base of a candidate head the head box
        box = (1, 0) if side == nat.CORRUPT_TAIL else (0, 1)
        return torch.cat([kept[:, 1], center[:, box[0]], half[:, box[0]],
                          kept[:, 0], center[:, box[1]], half[:, box[1]]], dim=-1)

    def score_triple(self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
        return self._triple_torch(head_emb, self.relation_embedding, relation_id, tail_emb)

    def _score_candidates(self, side: int, kept: torch.Tensor, relation_id: torch.Tensor,
                          cand: torch.Tensor) -> torch.Tensor:
        q = self._query_torch(side, kept, self.relation_embedding, relation_id)
        return ops.ReduceNegatives.apply(self.kernel_desc(), bool(self.negative_sample_sharing), q.contiguous(),
                                         self._table_dtype(cand))

    def score_heads(self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
        return self._score_candidates(nat.CORRUPT_HEAD, tail_emb, relation_id, head_emb)

    def score_tails(self, head_emb: torch.Tensor, relation_id: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
        return self._score_candidates(nat.CORRUPT_TAIL, head_emb, relation_id, tail_emb)
