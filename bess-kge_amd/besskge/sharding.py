"""Entity sharding and triple partitioning (host side, numpy, bit-exact).

Host-index layer of the BESS hot path.  The
entity table is cut into `n_shard` equal random shards, one per GPU; triples are
bucketed by the shard (pair) of their head / tail so that one balanced
all-to-all per step completes every micro-batch block.

Every array produced here is *bit-identical* to the reference for the same
arguments (it is pinned against `tests/golden/{sharding,partition}.npz`, which
were produced by the reference's own code):

  * `Sharding.create`                  <- reference `besskge/sharding.py:67-137`
  * `PartitionedTripleSet.partition_triples`   <- `sharding.py:226-265`
  * `PartitionedTripleSet.create_from_dataset` <- `sharding.py:267-376`
  * `PartitionedTripleSet.create_from_queries` <- `sharding.py:378-511`

The random stream (`np.random.default_rng(seed).permutation`) and the unstable
`np.argsort` of the bucket id are part of the contract, so they are called the
same way; everything around them is written for O(n) work where the reference
sorts (see `_inverse_permutation`, `_type_layout`).
"""

import dataclasses
import warnings
from pathlib import Path
from typing import Optional, Tuple

import numpy as np
from numpy.typing import NDArray

from besskge.dataset import KGDataset

_PARTITION_MODES = ("h_shard", "t_shard", "ht_shardpair")


def _inverse_permutation(perm: NDArray[np.int64]) -> NDArray[np.int64]:
    """inv[perm[i]] = i.  Equals `np.argsort(perm)` for a permutation (keys are
    unique, so stability does not matter) at O(n) instead of O(n log n)."""
    inv = np.empty(perm.shape[0], dtype=np.int64)
    inv[perm] = np.arange(perm.shape[0], dtype=np.int64)
    return inv


def _type_layout(
    sorted_ids: NDArray[np.int64],
    type_offsets: NDArray[np.int64],
    n_pad: NDArray[np.int64],
) -> Tuple[NDArray[np.int64], NDArray[np.int64]]:
    """Per-shard counts / local offsets of each entity type.

    Rows of `sorted_ids` are ascending, so the first local index of type k is a
    binary search for its first global ID.  Padding IDs (>= n_entity) fall in
    the last type and are removed from its count.
    """
    n_shard, per_shard = sorted_ids.shape
    starts = np.stack(
        [np.searchsorted(row, type_offsets, side="left") for row in sorted_ids]
    ).astype(np.int64)
    if np.any(starts[:, 0] != 0):
        raise ValueError("type_offsets[0] must be the smallest entity ID (0)")
    ends = np.concatenate(
        [starts[:, 1:], np.full((n_shard, 1), per_shard, dtype=np.int64)], axis=1
    )
    counts = ends - starts
    counts[:, -1] -= n_pad
    return counts, starts


@dataclasses.dataclass
class Sharding:
    """Entity <-> (shard, local row) maps."""

    #: number of shards (= GPUs holding a slice of the entity table)
    n_shard: int
    #: shard of each entity, int[n_entity]
    entity_to_shard: NDArray[np.int32]
    #: local row of each entity on its shard, int[n_entity]
    entity_to_idx: NDArray[np.int32]
    #: global ID at (shard, local row), int[n_shard, max_entity_per_shard];
    #: rows ascending, padding rows hold IDs >= n_entity
    shard_and_idx_to_entity: NDArray[np.int32]
    #: real (non padding) rows per shard, int64[n_shard]
    shard_counts: NDArray[np.int64]
    #: rows of each type per shard, int64[n_shard, n_types] (or None)
    entity_type_counts: Optional[NDArray[np.int64]]
    #: first local row of each type per shard, int64[n_shard, n_types] (or None)
    entity_type_offsets: Optional[NDArray[np.int64]]

    @property
    def n_entity(self) -> int:
        """Number of real entities."""
        return len(self.entity_to_shard)

    @property
    def max_entity_per_shard(self) -> int:
        """Rows in every shard (padding included)."""
        return self.shard_and_idx_to_entity.shape[1]

    @classmethod
    def create(
        cls,
        n_entity: int,
        n_shard: int,
        seed: int,
        type_offsets: Optional[NDArray[np.int64]] = None,
    ) -> "Sharding":
        """Balanced random sharding of `n_entity` entities over `n_shard` shards.

        :param n_entity: number of entities.
        :param n_shard: number of shards.
        :param seed: seed of the shuffling stream.
        :param type_offsets: first global ID of each entity type (IDs are
            clustered by type); None if entities are untyped.
        """
        per_shard = int(np.ceil(n_entity / n_shard))
        total = n_shard * per_shard
        shuffled = np.random.default_rng(seed).permutation(total)
        # ascending IDs inside a shard keep same-type entities contiguous
        ids = np.sort(shuffled.reshape(n_shard, per_shard), axis=1)
        where = _inverse_permutation(ids.reshape(-1))[:n_entity]
        entity_to_shard, entity_to_idx = np.divmod(where, per_shard)

        n_pad = np.sum(ids >= n_entity, axis=-1)
        shard_counts = per_shard - n_pad

        type_counts: Optional[NDArray[np.int64]] = None
        type_starts: Optional[NDArray[np.int64]] = None
        if type_offsets is not None:
            type_counts, type_starts = _type_layout(
                ids, np.asarray(type_offsets), n_pad
            )

        return cls(
            n_shard=n_shard,
            entity_to_shard=entity_to_shard,
            entity_to_idx=entity_to_idx,
            shard_and_idx_to_entity=ids,
            shard_counts=shard_counts,
            entity_type_counts=type_counts,
            entity_type_offsets=type_starts,
        )

    def save(self, out_file: Path) -> None:
        """Write all fields to an .npz file (fields that are None are left out: the file
        holds plain arrays only and loads without pickle)."""
        np.savez(out_file, **{k: v for k, v in dataclasses.asdict(self).items() if v is not None})

    @classmethod
    def load(cls, path: Path) -> "Sharding":
        """Read a sharding written by :meth:`save`."""
        fields = dict(np.load(path, allow_pickle=False))
        n_shard = int(fields.pop("n_shard"))
        for k in ("entity_type_counts", "entity_type_offsets"):
            fields.setdefault(k, None)
        return cls(n_shard=n_shard, **fields)


@dataclasses.dataclass
class PartitionedTripleSet:
    """Triples grouped by the shard of the head ("h_shard"), of the tail
    ("t_shard"), or by the (head shard, tail shard) pair ("ht_shardpair",
    pair (i, j) is bucket i * n_shard + j).  Entity IDs on the partitioning
    side(s) are rewritten to local shard rows."""

    sharding: Sharding
    #: the set holds (t, r + n_rel, h) for every (h, r, t)
    inverse_triples: bool
    #: "h_shard" | "t_shard" | "ht_shardpair"
    partition_mode: str
    #: which side of a query was filled with a dummy entity: "head", "tail",
    #: "none" (or None when a ground truth was supplied)
    dummy: Optional[str]
    #: int[n_triple, 3] bucket-ordered triples (local IDs as described above)
    triples: NDArray[np.int32]
    #: triples per bucket, int64[n_shard] or [n_shard, n_shard]
    triple_counts: NDArray[np.int64]
    #: first triple of each bucket, same shape
    triple_offsets: NDArray[np.int64]
    #: `original[triple_sort_idx]` is the bucket order, int64[n_triple]
    triple_sort_idx: NDArray[np.int64]
    #: int[n_triple, 2] head / tail type IDs (bucket order) or None
    types: Optional[NDArray[np.int32]]
    #: int[n_triple or 1, n_neg] global IDs of candidate heads / tails or None
    neg_heads: Optional[NDArray[np.int32]]
    neg_tails: Optional[NDArray[np.int32]]

    @classmethod
    def partition_triples(
        cls,
        triples: NDArray[np.int32],
        sharding: Sharding,
        partition_mode: str,
    ) -> Tuple[
        NDArray[np.int32], NDArray[np.int64], NDArray[np.int64], NDArray[np.int64]
    ]:
        """Bucket `triples` -> (ordered triples, counts, offsets, sort index)."""
        if partition_mode not in _PARTITION_MODES:
            raise ValueError(
                f"Partition mode {partition_mode} not supported"
                " for triple partitioning"
            )
        n = sharding.n_shard
        by_head = partition_mode in ("h_shard", "ht_shardpair")
        by_tail = partition_mode in ("t_shard", "ht_shardpair")
        if by_head and by_tail:
            shard_h, shard_t = sharding.entity_to_shard[triples[:, [0, 2]].T]
            bucket = shard_h * n + shard_t
            shape: Tuple[int, ...] = (n, n)
        else:
            bucket = sharding.entity_to_shard[triples[:, 0 if by_head else -1]]
            shape = (n,)
        n_bucket = int(np.prod(shape))
        flat_counts = np.bincount(bucket, minlength=n_bucket)
        flat_offsets = np.concatenate([np.array([0]), np.cumsum(flat_counts)[:-1]])

        # NOTE: default (unstable) argsort is the reference's choice
        # (sharding.py:257); the order inside a bucket is part of the contract.
        order = np.argsort(bucket)
        ordered = triples[order]
        if by_head:
            ordered[:, 0] = sharding.entity_to_idx[ordered[:, 0]]
        if by_tail:
            ordered[:, -1] = sharding.entity_to_idx[ordered[:, -1]]
        return (
            ordered,
            flat_counts.reshape(shape),
            flat_offsets.reshape(shape),
            order,
        )

    @classmethod
    def create_from_dataset(
        cls,
        dataset: KGDataset,
        part: str,
        sharding: Sharding,
        partition_mode: str = "ht_shardpair",
        add_inverse_triples: bool = False,
    ) -> "PartitionedTripleSet":
        """Partition one part of a :class:`KGDataset`."""
        base = dataset.triples[part]
        n_base = base.shape[0]
        triples = base
        if add_inverse_triples:
            flipped = np.copy(base[:, ::-1])
            flipped[:, 1] += dataset.n_relation_type
            triples = np.concatenate([base, flipped], axis=0)

        ordered, counts, offsets, order = cls.partition_triples(
            triples, sharding, partition_mode
        )

        types = None
        all_types = dataset.ht_types
        if all_types and part in all_types:
            types = all_types[part]
            if add_inverse_triples:
                types = np.concatenate([types, types[:, ::-1]], axis=0)
            types = types[order]

        has_h = bool(dataset.neg_heads) and part in dataset.neg_heads  # type: ignore
        has_t = bool(dataset.neg_tails) and part in dataset.neg_tails  # type: ignore
        neg_h = dataset.neg_heads[part] if has_h else None  # type: ignore
        neg_t = dataset.neg_tails[part] if has_t else None  # type: ignore
        if add_inverse_triples and (has_h != has_t):
            raise ValueError(
                "To use inverse triples, either both or"
                " neither of negative heads and tails need to"
                f" be defined for the {part} part of the dataset"
            )
        if add_inverse_triples and has_h:
            # candidates of an inverse triple are those of the opposite side
            width = neg_h.shape[-1]  # type: ignore
            bh = np.broadcast_to(neg_h, (n_base, width))
            bt = np.broadcast_to(neg_t, (n_base, width))
            neg_h = np.concatenate([bh, bt], axis=0)
            neg_t = np.concatenate([bt, bh], axis=0)

        def _ordered_candidates(neg: Optional[NDArray[np.int32]]) -> Optional[NDArray[np.int32]]:
            if neg is None:
                return None
            neg = neg.reshape(-1, neg.shape[-1])
            return neg if neg.shape[0] == 1 else neg[order]

        return cls(
            sharding=sharding,
            inverse_triples=add_inverse_triples,
            partition_mode=partition_mode,
            dummy="none",
            triples=ordered,
            triple_counts=counts,
            triple_offsets=offsets,
            triple_sort_idx=order,
            types=types,
            neg_heads=_ordered_candidates(neg_h),
            neg_tails=_ordered_candidates(neg_t),
        )

    @classmethod
    def create_from_queries(
        cls,
        dataset: KGDataset,
        sharding: Sharding,
        queries: NDArray[np.int32],
        query_mode: str,
        ground_truth: Optional[NDArray[np.int32]] = None,
        negative: Optional[NDArray[np.int32]] = None,
        negative_type: Optional[str] = None,
    ) -> "PartitionedTripleSet":
        """Partition (h, r, ?) ("hr") or (?, r, t) ("rt") queries.

        The missing side is filled with `ground_truth` if given, else with a
        dummy entity (ID 0, or the first ID of `negative_type`).  `negative`
        (shape (n_query | 1, k), global IDs) are the candidates to score; the
        default is every entity (of `negative_type`, if set).
        """
        if query_mode not in ("hr", "rt"):
            raise ValueError(f"Query mode {query_mode} not supported")
        n_query = queries.shape[0]

        lo = hi = 0
        if negative_type:
            offs = dataset.type_offsets
            if not offs or negative_type not in offs:
                raise ValueError(
                    f"{negative_type} is not the label of"
                    " a type of entity in the KGDataset"
                )
            labels = list(offs.keys())
            firsts = list(offs.values())
            k = labels.index(negative_type)
            lo = firsts[k]
            # the reference's upper bound is (next type's first ID) - 1, used
            # as an exclusive end (sharding.py:429-440,457); kept as is
            hi = (firsts[k + 1] if k + 1 < len(firsts) else dataset.n_entity) - 1
            if negative is not None and (
                np.any(negative < lo) or np.any(negative >= hi)
            ):
                warnings.warn(
                    "The negative entities provided are not all"
                    " of the specified negative_type"
                )

        if ground_truth is not None:
            fill = ground_truth.reshape(n_query, 1)
        else:
            fill = np.full(fill_value=lo if negative_type else 0, shape=(n_query, 1))

        if negative is not None:
            candidates = negative.reshape(-1, negative.shape[-1])
        elif negative_type:
            candidates = np.expand_dims(np.arange(lo, hi), axis=0)
        else:
            candidates = np.expand_dims(np.arange(sharding.n_entity), axis=0)

        if query_mode == "hr":
            triples = np.concatenate([queries, fill], axis=-1)
            mode, side = "h_shard", "tail"
        else:
            triples = np.concatenate([fill, queries], axis=-1)
            mode, side = "t_shard", "head"
        dummy = side if ground_truth is None else None

        ordered, counts, offsets, order = cls.partition_triples(
            triples, sharding, mode
        )

        types = None
        if negative_type:
            bounds = np.fromiter(dataset.type_offsets.values(), dtype=np.int32)  # type: ignore
            types = np.digitize(ordered[:, [0, 2]], bounds) - 1

        if candidates.shape[0] != 1:
            candidates = candidates[order]

        return cls(
            sharding=sharding,
            inverse_triples=False,
            partition_mode=mode,
            dummy=dummy,
            triples=ordered,
            triple_counts=counts,
            triple_offsets=offsets,
            triple_sort_idx=order,
            types=types,
            neg_heads=candidates if query_mode == "rt" else None,
            neg_tails=candidates if query_mode == "hr" else None,
        )
