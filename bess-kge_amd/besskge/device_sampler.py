"""Device-side batch / negative sampling.

The reference samples on the host (numpy in DataLoader workers,
`batch_sampler.py:138-196`, `negative_sampler.py:104-230`) and ships index
tensors to the device every step.  At MI355X step times (0.3 ms for the
BASELINE config-2 micro-batch, whose 4096 x 256 negatives take numpy ~4 ms to
draw) that path is the bottleneck.  `DeviceBatchSampler` wraps an existing
host sampler and produces **the same tensors, bit for bit, directly in HBM**:
the numpy `Generator(PCG64)` streams are continued on the device
(`csrc/sampler.hip`), the host only tracks the stream position with integer
arithmetic.  `sampler.sample(idx)` equals `host_sampler[idx]` for the same
generator state, and `sync_host()` writes the advanced state back into the
host sampler's generators, so the two can be interleaved.

Supported: `RandomShardedBatchSampler`, `RigidShardedBatchSampler`;
`RandomShardedNegativeSampler`, `TypeBasedShardedNegativeSampler`,
`PlaceholderNegativeSampler`, and `TripleBasedShardedNegativeSampler`: its
fixed candidate lists (sharded, sorted and padded once by the host class,
reference `negative_sampler.py:385-540`) are kept in HBM and every step's
look-up + layout for the exchange (`negative_sampler.py:422-477`) is one kernel
(`bess_gather_candidate_lists`) - the sampler the wikikg2 validation against
500 candidates per triple uses (`notebooks/3_wikikg2_fp16.ipynb:962-995`).

`shards=range(r, r + 1)` makes rank `r` of a one-process-per-GPU job produce
only its own slice `[:, r]` of every tensor (each rank jumps to its part of
the common stream: no broadcast of indices).
"""

import dataclasses
from typing import Any, Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from besskge import _native as nat
from besskge.batch_sampler import RandomShardedBatchSampler, RigidShardedBatchSampler, ShardedBatchSampler
from besskge.negative_sampler import (
    PlaceholderNegativeSampler,
    RandomShardedNegativeSampler,
    TripleBasedShardedNegativeSampler,
    TypeBasedShardedNegativeSampler,
)

_M128 = (1 << 128) - 1
_M64 = (1 << 64) - 1
#: PCG_DEFAULT_MULTIPLIER_128 (numpy/random/src/pcg64/pcg64.h)
PCG64_MULT = 0x2360ED051FC65DA44385DF649FCCF645


def _pcg64_output(state: int) -> int:
    """XSL-RR 128/64 output function."""
    x = ((state >> 64) ^ state) & _M64
    r = state >> 122
    return ((x >> r) | (x << ((64 - r) & 63))) & _M64


@dataclasses.dataclass
class Pcg64Stream:
    """Position in a numpy `Generator(PCG64)` stream, tracked with Python ints.

    Mirrors `rng.bit_generator.state`: the 128-bit LCG state and increment and
    the buffered upper half of the last 64-bit output that 32-bit draws leave
    behind (`pcg64_next32`)."""

    state: int
    inc: int
    has_uint32: int = 0
    uinteger: int = 0

    @classmethod
    def from_generator(cls, rng: np.random.Generator) -> "Pcg64Stream":
        st = rng.bit_generator.state
        if st["bit_generator"] != "PCG64":
            raise TypeError(f"device sampling continues PCG64 streams, got {st['bit_generator']}")
        return cls(int(st["state"]["state"]), int(st["state"]["inc"]), int(st["has_uint32"]), int(st["uinteger"]))

    def to_generator(self, rng: np.random.Generator) -> None:
        rng.bit_generator.state = {
            "bit_generator": "PCG64",
            "state": {"state": self.state, "inc": self.inc},
            "has_uint32": self.has_uint32,
            "uinteger": self.uinteger,
        }

    def native(self) -> nat.Pcg64State:
        g = nat.Pcg64State()
        g.state_hi, g.state_lo = self.state >> 64, self.state & _M64
        g.inc_hi, g.inc_lo = self.inc >> 64, self.inc & _M64
        g.has_uint32, g.uinteger = self.has_uint32, self.uinteger
        return g

    def jump_table(self) -> np.ndarray:
        """uint64 [64, 4] = (a_hi, a_lo, c_hi, c_lo): s -> a*s + c is 2^j steps."""
        a, c = PCG64_MULT, self.inc
        rows = []
        for _ in range(64):
            rows.append([a >> 64, a & _M64, c >> 64, c & _M64])
            a, c = (a * a) & _M128, (a * c + c) & _M128
        return np.array(rows, dtype=np.uint64)

    def _advance(self, steps: int) -> None:
        a, c = PCG64_MULT, self.inc
        s = self.state
        while steps:
            if steps & 1:
                s = (a * s + c) & _M128
            a, c = (a * a) & _M128, (a * c + c) & _M128
            steps >>= 1
        self.state = s

    def skip64(self, n: int) -> None:
        """n draws of `integers()` with a range above 32 bits."""
        self._advance(n)

    def skip32(self, n: int) -> None:
        """n draws of `integers()` with a range of at most 32 bits."""
        fresh = n - (1 if self.has_uint32 and n > 0 else 0)
        if n > 0:
            self.has_uint32 = 0
        if fresh <= 0:
            return
        self._advance((fresh + 1) // 2)
        self.uinteger = _pcg64_output(self.state) >> 32
        self.has_uint32 = fresh & 1


class DeviceBatchSampler:
    """Bit-exact device twin of a host `ShardedBatchSampler` (see module docstring)."""

    def __init__(self, batch_sampler: ShardedBatchSampler, device: torch.device,
                 shards: Optional[Sequence[int]] = None) -> None:
        """
        :param batch_sampler: configured host sampler (its triples, buckets,
            options and *current generator states* are taken over).
        :param device: HIP device the index tensors are produced on.
        :param shards: contiguous range of shards whose slices are produced
            (default: all).
        """
        bs = batch_sampler
        ns = bs.negative_sampler
        if type(bs) not in (RandomShardedBatchSampler, RigidShardedBatchSampler):
            raise TypeError(f"no device twin for {type(bs).__name__}")
        if type(ns) not in (RandomShardedNegativeSampler, TypeBasedShardedNegativeSampler, PlaceholderNegativeSampler,
                            TripleBasedShardedNegativeSampler):
            raise TypeError(f"no device twin for {type(ns).__name__}")
        self.host = bs
        self.device = torch.device(device)
        n = bs.n_shard
        shards = list(range(n)) if shards is None else [int(s) for s in shards]
        if not shards or shards != list(range(shards[0], shards[0] + len(shards))) or shards[-1] >= n:
            raise ValueError(f"`shards` must be a contiguous range inside [0, {n})")
        self.shards = shards
        self.n_shard = n
        dev = self.device

        def put(x: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
            return torch.from_numpy(np.ascontiguousarray(x)).to(device=dev, dtype=dtype).contiguous()

        self._triples = put(bs.triples, torch.int32)
        self._pair_buckets = bs.triple_partition_mode == "ht_shardpair"
        self._batch_stream = Pcg64Stream.from_generator(bs.rng)
        self._neg_stream = Pcg64Stream.from_generator(ns.rng)
        self._batch_table = put(self._batch_stream.jump_table().view(np.int64), torch.int64)
        self._neg_table = put(self._neg_stream.jump_table().view(np.int64), torch.int64)
        if isinstance(bs, RandomShardedBatchSampler):
            self._counts = put(bs.triple_counts.reshape(-1), torch.int64)
            self._offsets = put(bs.triple_offsets.reshape(-1), torch.int64)
        else:
            self._padded_idx = put(bs.triple_padded_idx, torch.int64)
            self._triple_mask = put(bs.triple_mask, torch.bool)
        if bs.hrt_freq_weighting:
            self._hrt_weights = put(bs.hrt_weights, torch.float64)
        if isinstance(ns, TripleBasedShardedNegativeSampler):
            # candidate lists, already grouped by owning shard and padded to a common length by the host class
            if ns.corruption_scheme in ("h", "t"):
                self._cand = [put(ns.padded_negatives, torch.int32)]
                self._cand_mask = [put(ns.mask, torch.bool)]
                self._cand_sort = [put(ns.sort_neg_idx, torch.int32)] if ns.return_sort_idx else None
            elif ns.flat_negative_format:  # one list per side: row 0 = heads, row 1 = tails
                self._cand = [put(np.concatenate([ns.padded_negatives_h, ns.padded_negatives_t], axis=0), torch.int32)]
                self._cand_mask = [put(np.concatenate([ns.mask_h, ns.mask_t], axis=0), torch.bool)]
                self._cand_sort = [put(ns.sort_neg_h_idx, torch.int32), put(ns.sort_neg_t_idx, torch.int32)] \
                    if ns.return_sort_idx else None
            else:
                self._cand = [put(ns.padded_negatives_h, torch.int32), put(ns.padded_negatives_t, torch.int32)]
                self._cand_mask = [put(ns.mask_h, torch.bool), put(ns.mask_t, torch.bool)]
                self._cand_sort = [put(ns.sort_neg_h_idx, torch.int32), put(ns.sort_neg_t_idx, torch.int32)] \
                    if ns.return_sort_idx else None
        elif not isinstance(ns, PlaceholderNegativeSampler):
            self._shard_counts = put(ns.shard_counts, torch.int32)
        if isinstance(ns, TypeBasedShardedNegativeSampler):
            self._triple_types = put(ns.triple_types, torch.int32)
            self._type_counts = put(ns.type_counts, torch.int32)
            self._type_offsets = put(ns.type_offsets, torch.int32)

    # ------------------------------------------------------------------ streams
    def sync_host(self) -> None:
        """Write the advanced stream positions back into the host generators."""
        self._batch_stream.to_generator(self.host.rng)
        self._neg_stream.to_generator(self.host.negative_sampler.rng)

    def __len__(self) -> int:
        return len(self.host.get_dataloader_sampler(shuffle=False))

    def epoch(self, shuffle: bool = True) -> Iterator[Dict[str, torch.Tensor]]:
        """One pass, driven by the host sampler's own index sampler (what
        `get_dataloader(shuffle=...)` iterates with `num_workers=0`)."""
        for idx in self.host.get_dataloader_sampler(shuffle=shuffle):
            yield self.sample(idx)

    # ----------------------------------------------------------------- sampling
    def _sample_triples(self, idx: Optional[List[int]]) -> Dict[str, torch.Tensor]:
        bs = self.host
        if isinstance(bs, RandomShardedBatchSampler):
            shape = (bs.batches_per_step, *bs.triple_counts.shape, bs.positive_per_partition)
            out = nat.sample_bucket_indices(self._batch_stream.native(), self._batch_table, shape, self._counts,
                                            self._offsets)
            self._batch_stream.skip64(out.numel())
            return dict(sample_idx=out)
        if idx is None:
            raise ValueError("the rigid sampler needs the positions to take (`idx`)")
        pos = torch.as_tensor(np.asarray(idx, dtype=np.int64), device=self.device)

        def steps_first(x: torch.Tensor) -> torch.Tensor:
            x = x[..., pos]
            x = x.reshape(*x.shape[:-1], bs.batches_per_step, -1)
            return torch.movedim(x, -2, 0).contiguous()

        return dict(sample_idx=steps_first(self._padded_idx), triple_mask=steps_first(self._triple_mask))

    def _sample_negatives(self, sample_idx: torch.Tensor) -> Dict[str, torch.Tensor]:
        ns = self.host.negative_sampler
        if isinstance(ns, PlaceholderNegativeSampler):
            return {}
        if isinstance(ns, TripleBasedShardedNegativeSampler):
            return self._candidate_lists(sample_idx)
        n = self.n_shard
        n_step = int(sample_idx.shape[0])
        if ns.flat_negative_format:
            B = 2 if ns.corruption_scheme == "ht" else 1
        else:
            per_part = int(sample_idx.shape[-1])
            B = per_part if sample_idx.dim() == 3 else n * per_part
        wanted = None
        if isinstance(ns, TypeBasedShardedNegativeSampler):
            types = self._triple_types[sample_idx]  # [..., triple, 2]
            if ns.corruption_scheme == "h":
                t = types[..., 0]
            elif ns.corruption_scheme == "t":
                t = types[..., 1]
            else:
                half = sample_idx.shape[-1] // 2
                t = torch.cat([types[..., :half, 0], types[..., half:, 1]], dim=-1)
            wanted = t.reshape(n_step, n, -1).contiguous()
        out = nat.sample_negatives(
            self._neg_stream.native(), self._neg_table, n_step, n, self.shards[0], len(self.shards), B,
            ns.n_negative, self._shard_counts, wanted,
            getattr(self, "_type_counts", None) if wanted is not None else None,
            getattr(self, "_type_offsets", None) if wanted is not None else None,
            ns.local_sampling,
        )
        self._neg_stream.skip32(n_step * n * n * B * ns.n_negative)
        return dict(negative_entities=out)

    def _candidate_lists(self, sample_idx: torch.Tensor) -> Dict[str, torch.Tensor]:
        """`TripleBasedShardedNegativeSampler.__call__` on the device (reference negative_sampler.py:422-477)."""
        ns = self.host.negative_sampler
        n_step, n = int(sample_idx.shape[0]), self.n_shard
        per_part = int(sample_idx.shape[-1])
        half = per_part // 2
        flat = ns.flat_negative_format
        dev = self.device
        out: Dict[str, torch.Tensor] = {}
        if ns.corruption_scheme in ("h", "t"):
            lookup = torch.zeros((n_step, n, 1), dtype=torch.int64, device=dev) if flat \
                else sample_idx.reshape(n_step, n, -1).contiguous()
            ent, mask = nat.gather_candidate_lists(self._cand[0], self._cand_mask[0], lookup,
                                                   mask_gather_layout=ns.mask_on_gather)
            if ns.return_sort_idx:
                full = torch.zeros_like(sample_idx) if flat else sample_idx
                sort = self._cand_sort[0][full]
        elif flat:
            lookup = torch.tensor([0, 1], dtype=torch.int64, device=dev).expand(n_step, n, 2).contiguous()
            ent, mask = nat.gather_candidate_lists(self._cand[0], self._cand_mask[0], lookup,
                                                   mask_gather_layout=ns.mask_on_gather)
            if ns.return_sort_idx:
                zh = torch.zeros((*sample_idx.shape[:-1], half), dtype=torch.int64, device=dev)
                zt = torch.zeros((*sample_idx.shape[:-1], per_part - half), dtype=torch.int64, device=dev)
                sort = torch.cat([self._cand_sort[0][zh], self._cand_sort[1][zt]], dim=-2)
        else:
            lookup = sample_idx.reshape(n_step, n, -1).contiguous()
            ent, mask = nat.gather_candidate_lists(self._cand[0], self._cand_mask[0], lookup, self._cand[1],
                                                   self._cand_mask[1], per_part=per_part, half=half,
                                                   mask_gather_layout=ns.mask_on_gather)
            if ns.return_sort_idx:
                sort = torch.cat([self._cand_sort[0][sample_idx[..., :half]], self._cand_sort[1][sample_idx[..., half:]]],
                                 dim=-2)
        out["negative_entities"] = ent
        out["negative_mask"] = mask
        if ns.return_sort_idx:
            out["negative_sort_idx"] = sort.reshape(n_step, n, -1, sort.shape[-1]).contiguous()
        return out

    def sample(self, idx: Optional[List[int]] = None) -> Dict[str, torch.Tensor]:
        """The tensors `host_sampler[idx]` would return, on the device
        (restricted to `shards` along axis 1).  `idx` is only used by the rigid
        sampler, exactly as on the host."""
        bs = self.host
        lo, hi = self.shards[0], self.shards[-1] + 1
        extras = self._sample_triples(idx)
        if bs.duplicate_batch:
            extras = {k: torch.cat([v, v], dim=-1) for k, v in extras.items()}
        sample_idx = extras.pop("sample_idx").contiguous()
        want = [k for k in ("head", "relation", "tail") if k != bs.dummy]
        batch: Dict[str, Any] = nat.lookup_triples(self._triples, sample_idx, self._pair_buckets, want)
        batch.update(extras)
        drawn = self._sample_negatives(sample_idx)
        if "negative_entities" in drawn:
            batch["negative"] = drawn.pop("negative_entities")
        batch.update(drawn)  # negative_mask / negative_sort_idx of the triple-based sampler
        if bs.hrt_freq_weighting:
            w = self._hrt_weights[sample_idx]
            w = w.reshape(w.shape[0], w.shape[1], -1)
            w = w / w.sum(dim=-1, keepdim=True) * bs.shard_bs
            batch["triple_weight"] = w.to(torch.float32)
        if bs.return_triple_idx:
            batch["triple_idx"] = sample_idx
        if (lo, hi) != (0, self.n_shard):
            # (the random samplers have drawn only the slice of "negative" already; the candidate lists of the
            # triple-based sampler come whole: axis 1 is the gathering shard of the entities, and the scoring -
            # or, with mask_on_gather, the gathering - shard of the mask)
            whole = isinstance(bs.negative_sampler, TripleBasedShardedNegativeSampler)
            batch = {k: (v if (k == "negative" and not whole) else v[:, lo:hi].contiguous()) for k, v in batch.items()}
        return batch
