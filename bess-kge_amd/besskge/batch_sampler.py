"""Sharded batch samplers (host side, numpy, bit-exact).

SURVEY.md section 8a row a3.  One call yields `batches_per_step` micro-batches
for *every* shard: index tensors `head / relation / tail` of shape
`[step, n_shard, n_shard, positive_per_partition]` (for "ht_shardpair"
partitioning) whose layout already encodes the routing of the balanced
all-to-all - `tail[:, j, i]` are rows of shard `j` that GPU `j` gathers and
sends to GPU `i` - plus the negatives drawn by the attached negative sampler.

Bit-identical to the reference for the same arguments and RNG state (pinned by
`tests/golden/batch_sampler.npz`):

  * `ShardedBatchSampler`         <- reference `batch_sampler.py:22-296`
  * `RigidShardedBatchSampler`    <- `batch_sampler.py:299-363`
  * `RandomShardedBatchSampler`   <- `batch_sampler.py:366-409`

The PopTorch async DataLoader is replaced by a plain
`torch.utils.data.DataLoader` (as the reference's own CI does,
`tests/test_bess.py:129-135`).
"""

import warnings
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Union, cast

import numpy as np
import torch
from numpy.typing import NDArray

from besskge.negative_sampler import ShardedNegativeSampler
from besskge.sharding import PartitionedTripleSet

_SampleDict = Dict[str, Union[NDArray[np.int64], NDArray[np.bool_]]]


def _steps_first(x: NDArray, n_step: int) -> NDArray:
    """[shard, ..., (step triple)] -> [step, shard, ..., triple]"""
    split = x.reshape(*x.shape[:-1], n_step, -1)
    return np.moveaxis(split, -2, 0)


class ShardedBatchSampler(torch.utils.data.Dataset, ABC):  # type: ignore
    """Base class: turns triple indices into per-shard index tensors."""

    def __init__(
        self,
        partitioned_triple_set: PartitionedTripleSet,
        negative_sampler: ShardedNegativeSampler,
        shard_bs: int,
        batches_per_step: int,
        seed: int,
        hrt_freq_weighting: bool = False,
        weight_smoothing: float = 0.0,
        duplicate_batch: bool = False,
        return_triple_idx: bool = False,
    ):
        """
        :param partitioned_triple_set: bucketed triples.
        :param negative_sampler: sampler of corrupting entities.
        :param shard_bs: positive triples per shard and micro-batch.
        :param batches_per_step: micro-batches produced per call.
        :param seed: RNG seed.
        :param hrt_freq_weighting: weight triples by inverse (h,r)/(r,t)
            frequency.
        :param weight_smoothing: additive smoothing of those frequencies.
        :param duplicate_batch: every block consists of two identical halves
            (inference with the "ht" corruption scheme).
        :param return_triple_idx: also return the indices of the sampled
            triples in `partitioned_triple_set.triples`.
        """
        pts = partitioned_triple_set
        self.n_shard = pts.sharding.n_shard
        self.triples = pts.triples
        self.dummy = pts.dummy
        self.triple_counts = pts.triple_counts
        self.triple_offsets = pts.triple_offsets
        self.triple_partition_mode = pts.partition_mode
        self.negative_sampler = negative_sampler
        self.shard_bs = shard_bs
        self.batches_per_step = batches_per_step
        self.duplicate_batch = duplicate_batch

        # GPU i scores blocks (i, 0..n-1) when triples are bucketed by shard pair
        per_part = shard_bs
        if self.triple_partition_mode == "ht_shardpair":
            per_part = int(np.ceil(shard_bs / self.n_shard))
        if duplicate_batch:
            per_part //= 2
        if negative_sampler.corruption_scheme == "ht":
            per_part -= per_part % 2  # two equal halves per block
        self.positive_per_partition = per_part
        #: triples drawn from every bucket per call
        self.partition_sample_size = batches_per_step * per_part

        self.hrt_freq_weighting = hrt_freq_weighting
        self.return_triple_idx = return_triple_idx
        self.seed = seed
        self.rng = np.random.default_rng(seed)

        if hrt_freq_weighting:
            if self.dummy != "none":
                warnings.warn(
                    "hrt frequency weights are being computed on dummy entities"
                )
            n_entity = pts.sharding.n_entity
            rel_key = n_entity * self.triples[..., 1]

            def pair_frequency(col: int) -> NDArray[np.int64]:
                """occurrences of each triple's (entity[col], relation) pair"""
                _, inverse, count = np.unique(
                    self.triples[..., col] + rel_key,
                    return_counts=True,
                    return_inverse=True,
                )
                return cast(NDArray[np.int64], count[inverse])

            freq = pair_frequency(0) + pair_frequency(2)
            self.hrt_weights = np.sqrt(1.0 / (freq + weight_smoothing))

    def __len__(self) -> int:
        """Sampler length: the largest bucket rounded up to whole calls."""
        n_call = int(np.ceil(self.triple_counts.max() / self.partition_sample_size))
        return n_call * self.partition_sample_size

    def __getitem__(self, idx: List[int]) -> Dict[str, torch.Tensor]:
        """Index tensors of one step (see module docstring)."""
        extras = self.sample_triples(idx)
        if self.duplicate_batch:
            extras = {k: np.concatenate([v, v], axis=-1) for k, v in extras.items()}
        sample_idx = cast(NDArray[np.int64], extras.pop("sample_idx"))

        hrt = self.triples[sample_idx]
        head, relation, tail = hrt[..., 0], hrt[..., 1], hrt[..., 2]
        if self.triple_partition_mode == "ht_shardpair":
            # [step, shard_h, shard_t, .] -> [step, shard_t, shard_h, .]: the
            # tail rows are gathered on shard_t and shipped to shard_h
            tail = np.swapaxes(tail, 1, 2)

        batch: Dict[str, Any] = dict(
            head=head.astype(np.int32),
            relation=relation.astype(np.int32),
            tail=tail.astype(np.int32),
        )
        batch.update(extras)
        drawn = dict(self.negative_sampler(sample_idx))
        if "negative_entities" in drawn:
            batch["negative"] = drawn.pop("negative_entities").astype(np.int32)
        batch.update(drawn)

        if self.dummy in ("head", "tail"):
            del batch[self.dummy]

        if self.hrt_freq_weighting:
            w = self.hrt_weights[sample_idx]
            w = w.reshape(w.shape[0], w.shape[1], -1)
            w /= np.sum(w, axis=-1, keepdims=True)
            w *= self.shard_bs
            batch["triple_weight"] = w.astype(np.float32)

        if self.return_triple_idx:
            batch["triple_idx"] = sample_idx

        return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}

    @abstractmethod
    def sample_triples(self, idx: List[int]) -> _SampleDict:
        """Pick the positive triples of one step.

        :return: dict with "sample_idx" [step, n_shard, (n_shard,) triple]
            and sampler specific extras.
        """
        raise NotImplementedError

    def get_dataloader_sampler(
        self, shuffle: bool
    ) -> torch.utils.data.Sampler:  # type: ignore
        """The `torch.utils.data.Sampler` that drives this dataset."""
        base = (
            torch.utils.data.RandomSampler(self)
            if shuffle
            else torch.utils.data.SequentialSampler(self)
        )
        return torch.utils.data.BatchSampler(
            base, batch_size=self.partition_sample_size, drop_last=False
        )

    def get_dataloader(
        self,
        options: Any = None,
        shuffle: bool = True,
        num_workers: int = 0,
        persistent_workers: bool = False,
        buffer_size: int = 16,
    ) -> torch.utils.data.DataLoader:  # type: ignore
        """DataLoader over the sampler.

        `options` is accepted for call compatibility with the reference
        (`batch_sampler.py:236-280`) and ignored; `buffer_size` maps to the
        prefetch depth of the worker processes.
        """
        kw: Dict[str, Any] = {}
        if num_workers > 0:
            kw.update(
                prefetch_factor=max(2, buffer_size // max(1, num_workers)),
                persistent_workers=persistent_workers,
            )
        return torch.utils.data.DataLoader(
            self,
            batch_size=None,
            sampler=self.get_dataloader_sampler(shuffle=shuffle),
            num_workers=num_workers,
            worker_init_fn=self.worker_init_fn,
            pin_memory=torch.cuda.is_available(),
            **kw,
        )

    @staticmethod
    def worker_init_fn(worker_id: int) -> None:
        """Give every DataLoader worker its own RNG streams."""
        info = torch.utils.data.get_worker_info()
        if info:
            ds = cast(ShardedBatchSampler, info.dataset)
            ds.rng = np.random.default_rng(ds.seed + worker_id)
            ds.negative_sampler.rng = np.random.default_rng(ds.seed + worker_id)


class RigidShardedBatchSampler(ShardedBatchSampler):
    """Sweeps all buckets with the same indices; shorter buckets wrap around
    and a mask flags the repeated (padding) triples."""

    def __init__(
        self,
        partitioned_triple_set: PartitionedTripleSet,
        negative_sampler: ShardedNegativeSampler,
        shard_bs: int,
        batches_per_step: int,
        seed: int,
        hrt_freq_weighting: bool = False,
        weight_smoothing: float = 0.0,
        duplicate_batch: bool = False,
        return_triple_idx: bool = False,
    ) -> None:
        super().__init__(
            partitioned_triple_set,
            negative_sampler,
            shard_bs,
            batches_per_step,
            seed,
            hrt_freq_weighting,
            weight_smoothing,
            duplicate_batch,
            return_triple_idx,
        )
        position = np.arange(len(self)).reshape(
            (1,) * self.triple_counts.ndim + (-1,)
        )
        counts = self.triple_counts[..., None]
        #: [n_shard, (n_shard,) padded_length]
        self.triple_mask = position < counts
        with np.errstate(divide="ignore"):
            wrapped = position % counts  # empty bucket: x % 0 == 0
        # clamp: an empty last bucket would point one past the end
        self.triple_padded_idx = np.minimum(
            wrapped + self.triple_offsets[..., None], self.triples.shape[0] - 1
        )

    def sample_triples(self, idx: List[int]) -> _SampleDict:
        n = self.batches_per_step
        return dict(
            sample_idx=_steps_first(self.triple_padded_idx[..., idx], n),
            triple_mask=_steps_first(self.triple_mask[..., idx], n),
        )


class RandomShardedBatchSampler(ShardedBatchSampler):
    """Uniform sampling with replacement inside every bucket."""

    def sample_triples(self, idx: List[int]) -> _SampleDict:
        shape = (
            self.batches_per_step,
            *self.triple_counts.shape,
            self.positive_per_partition,
        )
        draw = self.rng.integers(1 << 63, size=shape)
        counts = self.triple_counts[None, ..., None]
        offsets = self.triple_offsets[None, ..., None]
        return dict(sample_idx=offsets + draw % counts)

    def __len__(self) -> int:
        return int(np.ceil(self.triple_counts.max() / self.partition_sample_size))

    def get_dataloader_sampler(
        self, shuffle: bool = True
    ) -> torch.utils.data.Sampler:  # type: ignore
        return torch.utils.data.BatchSampler(
            torch.utils.data.SequentialSampler(self), batch_size=1, drop_last=False
        )
