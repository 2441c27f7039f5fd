"""Sharded negative samplers (host side, numpy, bit-exact).

SURVEY.md section 8a row a4.  Each sampler returns, per call, the *local rows*
of the entities used to corrupt the micro-batches of one step, laid out as
`[step, source shard, destination shard, B, n_negative]`: block `[s, d]` is
gathered from shard `s`'s table and (unless `local_sampling`) delivered to the
GPU that scores shard `d`'s micro-batch by the balanced all-to-all.

Bit-identical to the reference for the same arguments and RNG state (pinned by
`tests/golden/negative_sampler.npz`):

  * `RandomShardedNegativeSampler`      <- reference `negative_sampler.py:57-132`
  * `TypeBasedShardedNegativeSampler`   <- `negative_sampler.py:135-230`
  * `TripleBasedShardedNegativeSampler` <- `negative_sampler.py:233-540`
  * `PlaceholderNegativeSampler`        <- `negative_sampler.py:543-574`
"""

from abc import ABC, abstractmethod
from typing import Dict, Optional, Tuple, Union

import numpy as np
from numpy.typing import NDArray

from besskge.sharding import Sharding

_SamplerOutput = Dict[str, Union[NDArray[np.int32], NDArray[np.bool_]]]


def _fold_triple_axes(x: NDArray, n_tail: int) -> NDArray:
    """[step, shard, *mid, triple, *tail] -> [step, shard, (mid triple), *tail]
    where `tail` are the last `n_tail` axes."""
    tail = x.shape[x.ndim - n_tail :] if n_tail else ()
    return x.reshape(x.shape[0], x.shape[1], -1, *tail)


class ShardedNegativeSampler(ABC):
    """Interface of all negative samplers."""

    #: negatives are drawn once per shard pair (shared by the whole
    #: micro-batch) rather than once per triple
    flat_negative_format: bool
    #: negatives come only from the shard that scores the triple
    local_sampling: bool
    #: "h" (corrupt heads), "t" (tails) or "ht" (heads in the first half of
    #: every triple block, tails in the second)
    corruption_scheme: str
    #: random stream
    rng: np.random.Generator

    @abstractmethod
    def __call__(self, sample_idx: NDArray[np.int64]) -> _SamplerOutput:
        """Negatives for one step.

        :param sample_idx: [step, n_shard, (n_shard,) positive_per_partition]
            indices (into the partitioned triple set) of the positive triples.
        :return: dict with "negative_entities"
            [step, n_shard, n_shard, B, n_negative] (B = 1 | 2 | shard_bs) and
            sampler specific extras (masks, sort indices).
        """
        raise NotImplementedError


class RandomShardedNegativeSampler(ShardedNegativeSampler):
    """Uniform random negatives from every shard."""

    def __init__(
        self,
        n_negative: int,
        sharding: Sharding,
        seed: int,
        corruption_scheme: str,
        local_sampling: bool,
        flat_negative_format: bool = False,
    ) -> None:
        """
        :param n_negative: negatives per shard pair (flat format) or per
            triple and source shard.
        :param sharding: entity sharding.
        :param seed: RNG seed.
        :param corruption_scheme: "h", "t" or "ht".
        :param local_sampling: only use negatives of the scoring shard.
        :param flat_negative_format: one negative set per shard pair, shared
            by all its triples (requires negative sample sharing).
        """
        self.n_negative = n_negative
        self.sharding = sharding
        self.shard_counts = sharding.shard_counts
        self.corruption_scheme = corruption_scheme
        self.local_sampling = local_sampling
        self.flat_negative_format = flat_negative_format
        self.seed = seed
        self.rng = np.random.default_rng(seed=seed)

    def _negative_batch_dim(self, sample_idx: NDArray[np.int64]) -> int:
        if self.flat_negative_format:
            return 2 if self.corruption_scheme == "ht" else 1
        n_shard, per_partition = sample_idx.shape[1], sample_idx.shape[-1]
        return per_partition if sample_idx.ndim == 3 else n_shard * per_partition

    def __call__(self, sample_idx: NDArray[np.int64]) -> _SamplerOutput:
        n_step, n_shard = sample_idx.shape[:2]
        shape = (
            n_step,
            n_shard,
            n_shard,
            self._negative_batch_dim(sample_idx),
            self.n_negative,
        )
        raw = self.rng.integers(1 << 31, size=shape).astype(np.int32)
        # rows of the *source* shard (axis 1)
        rows = raw % self.shard_counts.reshape(1, -1, 1, 1, 1)
        return dict(negative_entities=rows)


class TypeBasedShardedNegativeSampler(RandomShardedNegativeSampler):
    """Random negatives of the same entity type as the corrupted entity."""

    def __init__(
        self,
        triple_types: NDArray[np.int32],
        n_negative: int,
        sharding: Sharding,
        corruption_scheme: str,
        local_sampling: bool,
        seed: int,
    ) -> None:
        """
        :param triple_types: [n_triple, 2] type IDs of head and tail.
        (other parameters: see :class:`RandomShardedNegativeSampler`)
        """
        super().__init__(
            n_negative,
            sharding,
            seed,
            corruption_scheme,
            local_sampling,
            flat_negative_format=False,
        )
        if sharding.entity_type_counts is None or sharding.entity_type_offsets is None:
            raise ValueError("The provided entity sharding does not have entity types")
        self.triple_types = triple_types
        self.type_counts = sharding.entity_type_counts
        self.type_offsets = sharding.entity_type_offsets

    def _corrupted_type(self, sample_idx: NDArray[np.int64]) -> NDArray[np.int32]:
        types = self.triple_types[sample_idx]  # [..., triple, 2]
        if self.corruption_scheme == "h":
            return types[..., 0]
        if self.corruption_scheme == "t":
            return types[..., 1]
        if self.corruption_scheme == "ht":
            half = sample_idx.shape[-1] // 2
            return np.concatenate(
                [types[..., :half, 0], types[..., half:, 1]], axis=-1
            )
        raise ValueError(
            f"Corruption scheme {self.corruption_scheme}"
            f" not supported by {self.__class__}"
        )

    def __call__(self, sample_idx: NDArray[np.int64]) -> _SamplerOutput:
        n_shard = sample_idx.shape[1]
        # type wanted by every triple of every scoring shard: [step, shard, S]
        wanted = _fold_triple_axes(self._corrupted_type(sample_idx), 0)
        if self.local_sampling:
            # block [s, d] serves the triples scored on s
            wanted = np.repeat(wanted[:, :, None, :], n_shard, axis=2)
        else:
            # block [s, d] serves the triples scored on d
            wanted = np.repeat(wanted[:, None, :, :], n_shard, axis=1)
        raw = super().__call__(sample_idx)["negative_entities"]
        source = np.arange(n_shard).reshape(1, -1, 1, 1)
        count = self.type_counts[source, wanted][..., None]
        first = self.type_offsets[source, wanted][..., None]
        return dict(negative_entities=raw % count + first)


class TripleBasedShardedNegativeSampler(ShardedNegativeSampler):
    """Predetermined candidates (per triple, or one list for all triples)."""

    def __init__(
        self,
        negative_heads: Optional[NDArray[np.int32]],
        negative_tails: Optional[NDArray[np.int32]],
        sharding: Sharding,
        corruption_scheme: str,
        seed: int,
        mask_on_gather: bool = False,
        return_sort_idx: bool = False,
    ):
        """
        :param negative_heads: [N, n_negative] global IDs (N = n_triple or 1).
        :param negative_tails: [N, n_negative] global IDs (N = n_triple or 1).
        :param sharding: entity sharding.
        :param corruption_scheme: "h", "t" or "ht".
        :param seed: RNG seed (unused by the sampler itself).
        :param mask_on_gather: lay the padding mask out for the shard that
            *gathers* the negatives instead of the one that scores them.
        :param return_sort_idx: also return, per triple, the permutation that
            maps the caller's candidate order to the shard-grouped order.
        """
        if corruption_scheme not in ("h", "t", "ht"):
            raise ValueError(
                f"Corruption scheme {corruption_scheme}"
                f" not supported by {self.__class__}"
            )
        if negative_heads is None and negative_tails is None:
            raise ValueError(
                "At least one of negative_heads and negative_tails"
                " needs to be provided"
            )
        if negative_heads is not None and negative_tails is not None:
            assert (
                negative_heads.shape == negative_tails.shape
            ), "negative_heads and negative_tails need to have the same size"
        elif negative_tails is not None:
            assert corruption_scheme == "t", (
                f"Corruption scheme '{corruption_scheme}' requires"
                " providing negative_heads"
            )
        else:
            assert corruption_scheme == "h", (
                f"Corruption scheme '{corruption_scheme}' requires"
                " providing negative_tails"
            )
        if negative_heads is not None:
            negative_heads = negative_heads.reshape(-1, negative_heads.shape[-1])
        if negative_tails is not None:
            negative_tails = negative_tails.reshape(-1, negative_tails.shape[-1])
        some = negative_heads if negative_heads is not None else negative_tails
        self.N, self.n_negative = some.shape  # type: ignore

        self.sharding = sharding
        self.shard_counts = sharding.shard_counts
        self.corruption_scheme = corruption_scheme
        self.local_sampling = False
        self.flat_negative_format = self.N == 1
        self.return_sort_idx = return_sort_idx
        self.mask_on_gather = mask_on_gather
        self.rng = np.random.default_rng(seed=seed)

        def prepare(cands: NDArray[np.int32]) -> Tuple[NDArray, NDArray, NDArray]:
            counts, order = self.shard_negatives(cands)
            local = sharding.entity_to_idx[np.take_along_axis(cands, order, axis=-1)]
            return counts, order, local

        if corruption_scheme in ("h", "t"):
            cands = negative_heads if corruption_scheme == "h" else negative_tails
            counts, self.sort_neg_idx, local = prepare(cands)  # type: ignore
            self.padded_shard_length = counts.max()
            self.padded_negatives, self.mask = self.pad_negatives(
                local, counts, self.padded_shard_length
            )
        else:
            counts_h, self.sort_neg_h_idx, local_h = prepare(negative_heads)  # type: ignore
            counts_t, self.sort_neg_t_idx, local_t = prepare(negative_tails)  # type: ignore
            self.padded_shard_length = np.max([counts_h.max(), counts_t.max()])
            self.padded_negatives_h, self.mask_h = self.pad_negatives(
                local_h, counts_h, self.padded_shard_length
            )
            self.padded_negatives_t, self.mask_t = self.pad_negatives(
                local_t, counts_t, self.padded_shard_length
            )

    # -- layout helpers ----------------------------------------------------
    @staticmethod
    def _to_gather_layout(x: NDArray) -> NDArray:
        """[step, shard, *mid, triple, shard_neg, L] ->
        [step, shard_neg, shard, (mid triple), L]"""
        return np.moveaxis(_fold_triple_axes(x, 2), 3, 1)

    @staticmethod
    def _to_score_layout(x: NDArray) -> NDArray:
        """[step, shard, *mid, triple, shard_neg, L] ->
        [step, shard, (mid triple), shard_neg, L]"""
        return _fold_triple_axes(x, 2)

    def _mask_layout(self, x: NDArray) -> NDArray:
        return self._to_gather_layout(x) if self.mask_on_gather else self._to_score_layout(x)

    def __call__(self, sample_idx: NDArray[np.int64]) -> _SamplerOutput:
        n_step, n_shard = sample_idx.shape[:2]
        sort_idx = None
        if self.corruption_scheme in ("h", "t"):
            lookup = sample_idx
            if self.flat_negative_format:
                # the single candidate list serves every triple
                lookup = np.full(fill_value=0, shape=(n_step, n_shard, 1))
            entities = self._to_gather_layout(self.padded_negatives[lookup])
            mask = self._mask_layout(self.mask[lookup])
            if self.return_sort_idx:
                full = (
                    np.full(fill_value=0, shape=sample_idx.shape)
                    if self.flat_negative_format
                    else sample_idx
                )
                sort_idx = self.sort_neg_idx[full]
        else:
            half = sample_idx.shape[-1] // 2
            if self.flat_negative_format:
                both = np.concatenate(
                    [self.padded_negatives_h, self.padded_negatives_t], axis=0
                )  # [2, shard_neg, L]
                both_mask = np.concatenate([self.mask_h, self.mask_t], axis=0)
                gather_shape = (n_step, both.shape[1], n_shard, 2, both.shape[2])
                entities = np.broadcast_to(
                    np.moveaxis(both, 0, 1)[None, :, None, :, :], gather_shape
                ).copy()
                if self.mask_on_gather:
                    mask = np.broadcast_to(
                        np.moveaxis(both_mask, 0, 1)[None, :, None, :, :], gather_shape
                    ).copy()
                else:
                    mask = np.broadcast_to(
                        both_mask[None, None, :, :, :],
                        (n_step, n_shard, 2, both.shape[1], both.shape[2]),
                    ).copy()
                idx_h = np.full(fill_value=0, shape=(*sample_idx.shape[:-1], half))
                idx_t = np.full(
                    fill_value=0,
                    shape=(*sample_idx.shape[:-1], sample_idx.shape[-1] - half),
                )
            else:
                idx_h = sample_idx[..., :half]
                idx_t = sample_idx[..., half:]
                entities = self._to_gather_layout(
                    np.concatenate(
                        [self.padded_negatives_h[idx_h], self.padded_negatives_t[idx_t]],
                        axis=-3,
                    )
                )
                mask = self._mask_layout(
                    np.concatenate([self.mask_h[idx_h], self.mask_t[idx_t]], axis=-3)
                )
            if self.return_sort_idx:
                sort_idx = np.concatenate(
                    [self.sort_neg_h_idx[idx_h], self.sort_neg_t_idx[idx_t]], axis=-2
                )
        out: _SamplerOutput = dict(negative_entities=entities, negative_mask=mask)
        if self.return_sort_idx:
            out.update(negative_sort_idx=_fold_triple_axes(sort_idx, 1))
        return out

    def shard_negatives(
        self,
        negatives: NDArray[np.int32],
    ) -> Tuple[NDArray[np.int64], NDArray[np.int32]]:
        """Group every candidate list by owning shard.

        :param negatives: [N, n_negative] global IDs.
        :return: (candidates per shard [N, n_shard], permutation [N, n_negative]
            that orders each list by shard).
        """
        n_shard = self.sharding.n_shard
        owner = self.sharding.entity_to_shard[negatives]
        row_base = n_shard * np.arange(self.N)[:, None]
        per_shard = np.bincount(
            (owner + row_base).flatten(), minlength=n_shard * self.N
        ).reshape(self.N, n_shard)
        # default (unstable) argsort, as negative_sampler.py:499
        order = np.argsort(owner, axis=-1)
        return per_shard, order.astype(np.int32)

    def pad_negatives(
        self,
        negatives: NDArray[np.int32],
        shard_counts: NDArray[np.int64],
        padded_shard_length: int,
    ) -> Tuple[NDArray[np.int32], NDArray[np.bool_]]:
        """Cut shard-ordered lists into per-shard lists of equal length.

        Short lists are filled by cycling through their own entries; `mask`
        marks the real ones.

        :param negatives: [N, n_negative] local rows, grouped by shard.
        :param shard_counts: [N, n_shard] list lengths.
        :param padded_shard_length: common length L.
        :return: (padded [N, n_shard, L], mask [N, n_shard, L]).
        """
        slot = np.arange(padded_shard_length)[None, None, :]
        mask = slot < shard_counts[..., None]
        first = np.c_[[0] * self.N, np.cumsum(shard_counts, axis=-1)[:, :-1]]
        with np.errstate(divide="ignore"):
            # an empty list gives x % 0 == 0 (numpy integer semantics)
            cyc = slot % shard_counts[..., None]
        src = np.minimum(cyc + first[..., None], self.n_negative - 1)
        padded = negatives[np.arange(self.N)[:, None, None], src]
        return padded, mask


class PlaceholderNegativeSampler(ShardedNegativeSampler):
    """Returns no negatives; used when queries are scored against every
    entity of the graph."""

    def __init__(self, corruption_scheme: str, seed: int = 0) -> None:
        self.corruption_scheme = corruption_scheme
        self.local_sampling = False
        self.flat_negative_format = True
        self.seed = seed
        self.rng = np.random.default_rng(seed=seed)

    def __call__(self, sample_idx: NDArray[np.int64]) -> _SamplerOutput:
        return dict()
