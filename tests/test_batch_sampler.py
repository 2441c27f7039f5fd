"""Host index layer: sharded batch samplers.

Golden part: bit-exact against tests/golden/batch_sampler.npz (produced by the
reference's code).  Structural part follows the reference's
`tests/test_batch_sampler.py:56-235`: un-shard a batch through
`shard_and_idx_to_entity` (including the tail block transpose that stands for
the all-to-all) and compare with the dataset.
"""

import numpy as np
import pytest
import torch
from numpy.testing import assert_equal

from besskge.batch_sampler import RandomShardedBatchSampler, RigidShardedBatchSampler
from besskge.dataset import KGDataset
from besskge.negative_sampler import RandomShardedNegativeSampler
from besskge.sharding import PartitionedTripleSet, Sharding

from conftest import load_golden


def _dataset(n_entity, n_rel, triples):
    return KGDataset(
        n_entity=n_entity,
        n_relation_type=n_rel,
        triples={"train": triples},
        original_triple_ids={"train": np.arange(triples.shape[0])},
    )


# ----------------------------------------------------------------- golden ---
@pytest.mark.parametrize("mode", ["h_shard", "t_shard", "ht_shardpair"])
@pytest.mark.parametrize("kind", ["rigid", "random"])
@pytest.mark.parametrize("dup", [False, True])
@pytest.mark.parametrize("scheme, flat", [("h", False), ("ht", True)])
@pytest.mark.parametrize("hrt", [False, True])
def test_batch_sampler_golden(mode, kind, dup, scheme, flat, hrt):
    g = load_golden("batch_sampler")
    seed, n_entity, n_rel, n_shard, n_triple, bps, shard_bs, n_negative = (
        int(x) for x in g["args"]
    )
    ds = _dataset(n_entity, n_rel, g["triples"])
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding, partition_mode=mode)
    ns = RandomShardedNegativeSampler(
        n_negative=n_negative,
        sharding=sharding,
        seed=seed,
        corruption_scheme=scheme,
        local_sampling=False,
        flat_negative_format=flat,
    )
    cls = RigidShardedBatchSampler if kind == "rigid" else RandomShardedBatchSampler
    bs = cls(
        partitioned_triple_set=pts,
        negative_sampler=ns,
        shard_bs=shard_bs,
        batches_per_step=bps,
        seed=seed,
        hrt_freq_weighting=hrt,
        weight_smoothing=0.5 if hrt else 0.0,
        duplicate_batch=dup,
        return_triple_idx=True,
    )
    p = f"{mode}_{kind}_{int(dup)}_{scheme}_{int(hrt)}_"
    assert len(bs) == int(g[p + "len"])
    assert bs.positive_per_partition == int(g[p + "ppp"])
    it = iter(bs.get_dataloader_sampler(shuffle=False))
    idxs = [next(it), next(it)]
    if kind == "rigid":
        idxs.append(list(bs.get_dataloader_sampler(shuffle=False))[-1])
    for j, idx in enumerate(idxs):
        assert_equal(np.array(idx), g[p + f"b{j}_idx"])
        batch = bs[idx]
        want_keys = sorted(
            k[len(p + f"b{j}_") :] for k in g.files if k.startswith(p + f"b{j}_") and not k.endswith("_idx")
        ) + ["triple_idx"]
        assert sorted(batch.keys()) == sorted(set(want_keys))
        for k, v in batch.items():
            want = g[p + f"b{j}_{k}"]
            assert isinstance(v, torch.Tensor)
            got = v.numpy()
            assert got.dtype == want.dtype, (k, got.dtype, want.dtype)
            assert got.shape == want.shape, (k, got.shape, want.shape)
            assert np.array_equal(got, want), k


# ------------------------------------------------------------- structural ---
seed = 1234
n_entity = 500
n_relation_type = 10
n_shard = 4
n_triple = 2000
batches_per_step = 3
shard_bs = 120
n_negative = 250

_rng = np.random.default_rng(seed)
triples = np.stack(
    [
        _rng.integers(n_entity, size=n_triple),
        _rng.integers(n_relation_type, size=n_triple),
        _rng.integers(n_entity, size=n_triple),
    ],
    axis=1,
)
ds = _dataset(n_entity, n_relation_type, triples)
sharding = Sharding.create(n_entity, n_shard, seed=seed)
ns = RandomShardedNegativeSampler(
    n_negative=n_negative,
    sharding=sharding,
    seed=seed,
    corruption_scheme="h",
    local_sampling=False,
    flat_negative_format=False,
)


def reconstruct(batch, mode):
    """Global (h, r, t) seen by each processing shard: [shard, step*S, 3]."""
    out = []
    for dev in range(n_shard):
        rel = batch["relation"][:, dev].flatten()
        if mode == "t_shard":
            heads = batch["head"][:, dev].flatten()
        else:
            heads = sharding.shard_and_idx_to_entity[dev, batch["head"][:, dev]].flatten()
        if mode == "h_shard":
            tails = batch["tail"][:, dev].flatten()
        elif mode == "t_shard":
            tails = sharding.shard_and_idx_to_entity[dev, batch["tail"][:, dev]].flatten()
        else:
            # tails arrive from every shard through the all-to-all
            tails = sharding.shard_and_idx_to_entity[
                np.arange(n_shard)[None, :, None], batch["tail"][:, :, dev]
            ].flatten()
        out.append(np.stack([heads, rel, tails], axis=1))
    return np.stack(out)


@pytest.mark.parametrize("mode", ["h_shard", "t_shard", "ht_shardpair"])
@pytest.mark.parametrize("dup", [True, False])
def test_random_bs(mode, dup):
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding, partition_mode=mode)
    bs = RandomShardedBatchSampler(
        partitioned_triple_set=pts,
        negative_sampler=ns,
        shard_bs=shard_bs,
        batches_per_step=batches_per_step,
        seed=seed,
        duplicate_batch=dup,
        return_triple_idx=True,
    )
    b = {k: v.numpy() for k, v in bs[next(iter(bs.get_dataloader_sampler(shuffle=True)))].items()}
    rec = reconstruct(b, mode)  # [shard, step*S, 3]
    want = ds.triples["train"][pts.triple_sort_idx][b["triple_idx"]]
    want = np.moveaxis(want, 0, 1).reshape(n_shard, -1, 3)
    assert_equal(rec, want)
    assert b["head"].dtype == np.int32 and b["negative"].dtype == np.int32
    if dup:
        cut = b["head"].shape[-1] // 2
        for k in ("head", "relation", "tail"):
            assert_equal(b[k][..., :cut], b[k][..., cut:])


@pytest.mark.parametrize("mode", ["h_shard", "t_shard", "ht_shardpair"])
@pytest.mark.parametrize("dup", [True, False])
@pytest.mark.parametrize("shuffle", [True, False])
def test_rigid_bs_epoch(mode, dup, shuffle):
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding, partition_mode=mode)
    bs = RigidShardedBatchSampler(
        partitioned_triple_set=pts,
        negative_sampler=ns,
        shard_bs=shard_bs,
        batches_per_step=batches_per_step,
        seed=seed,
        duplicate_batch=dup,
    )
    seen = []
    for idx in bs.get_dataloader_sampler(shuffle=shuffle):
        b = {k: v.numpy() for k, v in bs[idx].items()}
        rec = reconstruct(b, mode)
        mask = np.moveaxis(b["triple_mask"], 0, 1).reshape(n_shard, -1)
        seen.append(rec[mask])
    everything = np.sort(np.vstack(seen), axis=0)
    if dup:
        assert_equal(everything[::2], everything[1::2])
        everything = everything[::2]
    # one epoch == the dataset, exactly once
    assert_equal(everything, np.sort(ds.triples["train"], axis=0))


def test_dataloader_plain_torch():
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding)
    bs = RigidShardedBatchSampler(pts, ns, shard_bs, batches_per_step, seed)
    dl = bs.get_dataloader(shuffle=False)
    assert isinstance(dl, torch.utils.data.DataLoader)
    first = next(iter(dl))
    direct = bs[next(iter(bs.get_dataloader_sampler(shuffle=False)))]
    assert_equal(first["head"].numpy(), direct["head"].numpy())
    assert first["head"].shape == (batches_per_step, n_shard, n_shard, bs.positive_per_partition)
