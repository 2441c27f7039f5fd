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

The six other scorers of the reference (PairRE, TripleRE, ConvE, BoxE, InterHT,
TranS) are outside the hot path this package accelerates.
"""

from abc import ABC, abstractmethod
from typing import Callable, List, Union

import torch

from besskge import _native as nat
from besskge import ops
from besskge.embedding import (
    init_KGE_normal,
    init_KGE_uniform,
    init_uniform_rotation,
    initialize_entity_embedding,
    initialize_relation_embedding,
    refactor_embedding_sharding,
)
from besskge.sharding import Sharding

_Init = Union[torch.Tensor, List[Callable[..., torch.Tensor]]]


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
        """Descriptor handed to the kernels (dtype follows the tables)."""
        d = nat.ModelDesc()
        d.scorer = self._scorer_id
        d.norm_p = int(getattr(self, "scoring_norm", 0))
        d.dtype = nat._dtype_code(self.relation_embedding)
        d.width = int(self.entity_embedding.shape[-1])
        d.rel_width = int(self.relation_embedding.shape[-1])
        return d

    def _table_dtype(self, x: torch.Tensor) -> torch.Tensor:
        dt = self.relation_embedding.dtype
        return x if x.dtype == dt else x.to(dt)

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
    ) -> None:
        self.sharding = sharding
        self.entity_embedding = initialize_entity_embedding(
            sharding, entity_initializer, [entity_width]
        )
        self.relation_embedding = initialize_relation_embedding(
            n_relation_type, inverse_relations, relation_initializer, [relation_width]
        )
        assert (
            self.entity_embedding.shape[-1] == entity_width
            and self.relation_embedding.shape[-1] == relation_width
        ), what


class DistanceBasedScoreFunction(BaseScoreFunction, ABC):
    """Scorers of the form -||query - entity||_p."""

    def __init__(self, negative_sample_sharing: bool, scoring_norm: int) -> None:
        """
        :param negative_sample_sharing: see :class:`BaseScoreFunction`.
        :param scoring_norm: p of the p-norm (1 or 2 on the HIP path).
        """
        super().__init__()
        if scoring_norm not in (1, 2):
            raise ValueError("the HIP kernels implement scoring_norm 1 and 2")
        self.negative_sample_sharing = negative_sample_sharing
        self.scoring_norm = scoring_norm


class MatrixDecompositionScoreFunction(BaseScoreFunction, ABC):
    """Scorers of the form <query, entity>."""

    def __init__(self, negative_sample_sharing: bool) -> None:
        super().__init__()
        self.negative_sample_sharing = negative_sample_sharing


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
    ) -> None:
        super().__init__(negative_sample_sharing, scoring_norm)
        self._allocate(
            sharding, n_relation_type, inverse_relations, entity_initializer,
            relation_initializer, embedding_size, embedding_size,
            "TransE requires `embedding_size` embedding parameters"
            " for each entity and relation",
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
    ) -> None:
        super().__init__(negative_sample_sharing, scoring_norm)
        self._allocate(
            sharding, n_relation_type, inverse_relations, entity_initializer,
            relation_initializer, 2 * embedding_size, embedding_size,
            "RotatE requires `2*embedding_size` embedding parameters for each entity"
            " and `embedding_size` embedding parameters for each relation",
        )
        self.embedding_size = embedding_size


class DistMult(MatrixDecompositionScoreFunction):
    """DistMult: sum(h * r * t)  (reference scoring.py:746-837)."""

    _scorer_id = nat.DISTMULT

    def __init__(
        self,
        negative_sample_sharing: bool,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        entity_initializer: _Init = [init_KGE_uniform],
        relation_initializer: _Init = [init_KGE_uniform],
        inverse_relations: bool = False,
    ) -> None:
        super().__init__(negative_sample_sharing)
        self._allocate(
            sharding, n_relation_type, inverse_relations, entity_initializer,
            relation_initializer, embedding_size, embedding_size,
            "DistMult requires `embedding_size` embedding parameters"
            " for each entity and relation",
        )
        self.embedding_size = embedding_size


class ComplEx(MatrixDecompositionScoreFunction):
    """ComplEx: Re<h * r, conj t> on [re | im] rows (reference scoring.py:840-946)."""

    _scorer_id = nat.COMPLEX

    def __init__(
        self,
        negative_sample_sharing: bool,
        sharding: Sharding,
        n_relation_type: int,
        embedding_size: int,
        entity_initializer: _Init = [init_KGE_normal],
        relation_initializer: _Init = [init_KGE_normal],
        inverse_relations: bool = False,
    ) -> None:
        super().__init__(negative_sample_sharing)
        self._allocate(
            sharding, n_relation_type, inverse_relations, entity_initializer,
            relation_initializer, 2 * embedding_size, 2 * embedding_size,
            "ComplEx requires `2*embedding_size` embedding parameters"
            " for each entity and relation",
        )
        self.embedding_size = embedding_size
