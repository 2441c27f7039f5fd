"""Host index layer: sharded negative samplers.

Golden part: bit-exact against tests/golden/negative_sampler.npz (produced by
the reference's code).  Structural part follows the reference's
`tests/test_negative_sampler.py:30-300` (the all-to-all is simulated by
indexing `[:, :, processing_shard]`).
"""

import numpy as np
import pytest
from numpy.testing import assert_equal

from besskge.negative_sampler import (
    PlaceholderNegativeSampler,
    RandomShardedNegativeSampler,
    TripleBasedShardedNegativeSampler,
    TypeBasedShardedNegativeSampler,
)
from besskge.sharding import Sharding

from conftest import load_golden


def exact(a, b, what=""):
    a = np.asarray(a)
    assert a.dtype == b.dtype, f"{what}: dtype {a.dtype} != {b.dtype}"
    assert a.shape == b.shape, f"{what}: shape {a.shape} != {b.shape}"
    assert np.array_equal(a, b), f"{what}: values differ"


def _setup():
    g = load_golden("negative_sampler")
    seed, n_entity, n_shard, n_triple, bps, ppp, n_negative = (int(x) for x in g["args"])
    sharding = Sharding.create(n_entity, n_shard, seed=seed, type_offsets=g["type_offsets"])
    return g, sharding, seed, n_negative


# ----------------------------------------------------------------- golden ---
@pytest.mark.parametrize("pm", ["shard", "shardpair"])
@pytest.mark.parametrize("flat", [True, False])
@pytest.mark.parametrize("scheme", ["h", "ht"])
def test_random_golden(pm, flat, scheme):
    g, sharding, seed, n_negative = _setup()
    ns = RandomShardedNegativeSampler(
        n_negative=n_negative,
        sharding=sharding,
        seed=seed,
        corruption_scheme=scheme,
        local_sampling=False,
        flat_negative_format=flat,
    )
    sample_idx = g[f"sample_idx_{pm}"]
    for draw in (0, 1):  # consecutive draws continue the same stream
        exact(
            ns(sample_idx)["negative_entities"],
            g[f"random_{pm}_{int(flat)}_{scheme}_{draw}"],
        )


@pytest.mark.parametrize("pm", ["shard", "shardpair"])
@pytest.mark.parametrize("local", [True, False])
@pytest.mark.parametrize("scheme", ["h", "t", "ht"])
def test_type_based_golden(pm, local, scheme):
    g, sharding, seed, n_negative = _setup()
    ns = TypeBasedShardedNegativeSampler(
        triple_types=g["triple_types"],
        n_negative=n_negative,
        sharding=sharding,
        corruption_scheme=scheme,
        local_sampling=local,
        seed=seed,
    )
    exact(
        ns(g[f"sample_idx_{pm}"])["negative_entities"],
        g[f"type_{pm}_{int(local)}_{scheme}"],
    )


@pytest.mark.parametrize("pm", ["shard", "shardpair"])
@pytest.mark.parametrize("flat", [True, False])
@pytest.mark.parametrize("scheme", ["h", "t", "ht"])
@pytest.mark.parametrize("mog", [False, True])
def test_triple_based_golden(pm, flat, scheme, mog):
    g, sharding, seed, _ = _setup()
    ns = TripleBasedShardedNegativeSampler(
        g[f"tb_neg_heads_{int(flat)}"],
        g[f"tb_neg_tails_{int(flat)}"],
        sharding,
        corruption_scheme=scheme,
        seed=seed,
        return_sort_idx=True,
        mask_on_gather=mog,
    )
    out = ns(g[f"sample_idx_{pm}"])
    p = f"tb_{pm}_{int(flat)}_{scheme}_{int(mog)}_"
    for k in ("negative_entities", "negative_mask", "negative_sort_idx"):
        exact(out[k], g[p + k], p + k)
    assert int(ns.padded_shard_length) == int(g[p + "padded_shard_length"])
    assert ns.flat_negative_format == flat and ns.local_sampling is False


# ------------------------------------------------------------- structural ---
seed = 1234
n_entity = 500
n_shard = 4
n_triple = 2000
batches_per_step = 5
positive_per_partition = 60
cutpoint = positive_per_partition // 2
n_negative = 250
sizes = {
    "shard": (batches_per_step, n_shard, positive_per_partition),
    "shardpair": (batches_per_step, n_shard, n_shard, positive_per_partition),
}


@pytest.mark.parametrize("pm", ["shard", "shardpair"])
@pytest.mark.parametrize("flat", [True, False])
def test_random_in_range(pm, flat):
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    ns = RandomShardedNegativeSampler(
        n_negative=n_negative,
        sharding=sharding,
        seed=seed,
        corruption_scheme="ht",
        local_sampling=False,
        flat_negative_format=flat,
    )
    neg = ns(np.ones(sizes[pm], dtype=np.int64))["negative_entities"]
    B = 2 if flat else (positive_per_partition if pm == "shard" else n_shard * positive_per_partition)
    assert neg.shape == (batches_per_step, n_shard, n_shard, B, n_negative)
    for src in range(n_shard):
        assert neg[:, src].min() >= 0
        assert neg[:, src].max() < sharding.shard_counts[src]


@pytest.mark.parametrize("pm", ["shard", "shardpair"])
@pytest.mark.parametrize("local", [True, False])
@pytest.mark.parametrize("scheme", ["h", "t", "ht"])
def test_type_based_types(pm, local, scheme):
    rng = np.random.default_rng(seed)
    entity_types = np.concatenate(
        [np.zeros(200), np.ones(60), 2 * np.ones(n_entity - 260)]
    ).astype(np.int32)
    sharding = Sharding.create(
        n_entity, n_shard, seed=seed, type_offsets=np.array([0, 200, 260])
    )
    triple_types = rng.integers(3, size=(n_triple, 2)).astype(np.int32)
    ns = TypeBasedShardedNegativeSampler(
        triple_types=triple_types,
        n_negative=8,
        sharding=sharding,
        corruption_scheme=scheme,
        local_sampling=local,
        seed=seed,
    )
    sample_idx = rng.integers(n_triple, size=sizes[pm])
    neg = ns(sample_idx)["negative_entities"]
    for dst in range(n_shard):
        assert neg[:, dst].max() < sharding.shard_counts[dst]
        if local:
            got = entity_types[sharding.shard_and_idx_to_entity[dst, neg[:, dst]]]
        else:
            got = entity_types[
                sharding.shard_and_idx_to_entity[
                    np.arange(n_shard)[None, :, None, None], neg[:, :, dst]
                ]
            ]
        # [step, shard_neg, S, K]: one type per (step, triple)
        t0 = got[:, :1, :, :1]
        assert np.all(got == t0)
        got_type = t0[:, 0, :, 0]
        types = triple_types[sample_idx[:, dst]]  # [step, (n,) ppp, 2]
        if scheme == "h":
            want = types[..., 0]
        elif scheme == "t":
            want = types[..., 1]
        else:
            want = np.concatenate(
                [types[..., :cutpoint, 0], types[..., cutpoint:, 1]], axis=-1
            )
        assert_equal(got_type, want.reshape(batches_per_step, -1))


@pytest.mark.parametrize("pm", ["shard", "shardpair"])
@pytest.mark.parametrize("scheme", ["h", "t", "ht"])
@pytest.mark.parametrize("flat", [True, False])
def test_triple_based_reconstruction(pm, scheme, flat):
    rng = np.random.default_rng(seed)
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    n_neg = 37
    N = 1 if flat else n_triple
    neg_heads = rng.integers(n_entity, size=(N, n_neg)).astype(np.int32)
    neg_tails = rng.integers(n_entity, size=(N, n_neg)).astype(np.int32)
    ns = TripleBasedShardedNegativeSampler(
        neg_heads,
        neg_tails,
        sharding,
        corruption_scheme=scheme,
        seed=seed,
        return_sort_idx=True,
        mask_on_gather=False,
    )
    sample_idx = rng.integers(n_triple, size=sizes[pm])
    out = ns(sample_idx)
    ent, mask, sort_idx = out["negative_entities"], out["negative_mask"], out["negative_sort_idx"]
    assert np.all(mask.sum(axis=(-2, -1)) == n_neg)
    S = int(np.prod(sample_idx.shape[2:]))
    if flat:
        sample_idx = np.zeros_like(sample_idx)
    for dst in range(n_shard):
        assert ent[:, dst].max() < sharding.shard_counts[dst]
        # entities delivered to dst by the all-to-all: [step, shard_neg, B, L]
        recv = sharding.shard_and_idx_to_entity[
            np.arange(n_shard)[None, :, None, None], ent[:, :, dst]
        ]
        m = mask[:, dst]  # [step, B, shard_neg, L]
        idx_dst = sample_idx[:, dst].reshape(batches_per_step, -1)  # [step, S]
        half = S // (2 * (n_shard if pm == "shardpair" else 1))
        for step in range(batches_per_step):
            for s in range(0, S, max(1, S // 7)):
                if scheme == "ht":
                    in_block = s % (S // (n_shard if pm == "shardpair" else 1))
                    is_head = in_block < half
                else:
                    is_head = scheme == "h"
                b = (0 if is_head else 1) if (flat and scheme == "ht") else (0 if flat else s)
                got = np.moveaxis(recv[step], 0, 1)[b][m[step, b]]
                want = (neg_heads if is_head else neg_tails)[idx_dst[step, s]]
                assert_equal(got, want[sort_idx[step, dst, s]])


def test_triple_based_argument_checks():
    sharding = Sharding.create(50, 2, seed=0)
    cands = np.zeros((1, 4), dtype=np.int32)
    with pytest.raises(ValueError):
        TripleBasedShardedNegativeSampler(None, None, sharding, "h", seed=0)
    with pytest.raises(AssertionError):
        TripleBasedShardedNegativeSampler(None, cands, sharding, "h", seed=0)
    with pytest.raises(AssertionError):
        TripleBasedShardedNegativeSampler(cands, None, sharding, "ht", seed=0)


def test_placeholder():
    ns = PlaceholderNegativeSampler("t")
    assert ns(np.zeros((1, 2, 3), dtype=np.int64)) == {}
    assert ns.flat_negative_format and not ns.local_sampling
