"""Host index layer: Sharding / PartitionedTripleSet.

Golden part: bit-exact against arrays produced by the reference's own code
(tests/golden/{sharding,partition}.npz, generator tests/golden/make_golden.py).
Structural part: the invariants the reference checks in
`tests/test_sharding.py:43-285`.
"""

import numpy as np
import pytest
from numpy.testing import assert_equal

from besskge.dataset import KGDataset
from besskge.sharding import PartitionedTripleSet, Sharding

from conftest import load_golden


def exact(a, b, what=""):
    a = np.asarray(a)
    assert a.dtype == b.dtype, f"{what}: dtype {a.dtype} != {b.dtype}"
    assert a.shape == b.shape, f"{what}: shape {a.shape} != {b.shape}"
    assert np.array_equal(a, b), f"{what}: values differ"


# ----------------------------------------------------------------- golden ---
def test_sharding_create_golden():
    g = load_golden("sharding")
    for i in range(int(g["n_cases"])):
        n_entity, n_shard, seed = (int(x) for x in g[f"c{i}_args"])
        to = g[f"c{i}_type_offsets"]
        s = Sharding.create(n_entity, n_shard, seed, to if to.size else None)
        for k in (
            "entity_to_shard",
            "entity_to_idx",
            "shard_and_idx_to_entity",
            "shard_counts",
        ):
            exact(getattr(s, k), g[f"c{i}_{k}"], f"case {i} {k}")
        if to.size:
            exact(s.entity_type_counts, g[f"c{i}_entity_type_counts"], "type counts")
            exact(s.entity_type_offsets, g[f"c{i}_entity_type_offsets"], "type offs")
        else:
            assert s.entity_type_counts is None and s.entity_type_offsets is None


def _partition_setup():
    g = load_golden("partition")
    n_entity, n_rel, n_triple, n_shard, seed = (int(x) for x in g["args"])
    type_offsets = dict(zip("abc", (int(x) for x in g["type_offsets"])))
    sharding = Sharding.create(
        n_entity, n_shard, seed=seed, type_offsets=g["type_offsets"]
    )
    ds = KGDataset(
        n_entity=n_entity,
        n_relation_type=n_rel,
        entity_dict=None,
        relation_dict=None,
        type_offsets=type_offsets,
        triples={"train": g["triples"]},
        original_triple_ids={"train": np.arange(n_triple)},
        neg_heads={"train": g["neg_heads"]},
        neg_tails={"train": g["neg_tails"]},
    )
    return g, ds, sharding


@pytest.mark.parametrize("mode", ["h_shard", "t_shard", "ht_shardpair"])
@pytest.mark.parametrize("inverse", [False, True])
def test_create_from_dataset_golden(mode, inverse):
    g, ds, sharding = _partition_setup()
    pts = PartitionedTripleSet.create_from_dataset(
        ds, "train", sharding, partition_mode=mode, add_inverse_triples=inverse
    )
    p = f"{mode}_{int(inverse)}_"
    for k in (
        "triples",
        "triple_counts",
        "triple_offsets",
        "triple_sort_idx",
        "types",
        "neg_heads",
        "neg_tails",
    ):
        exact(getattr(pts, k), g[p + k], p + k)
    assert pts.dummy == "none" and pts.inverse_triples == inverse


@pytest.mark.parametrize("case", ["q_hr_plain", "q_rt_gt", "q_hr_neg", "q_hr_type"])
def test_create_from_queries_golden(case):
    g, ds, sharding = _partition_setup()
    queries, gt, neg = g["q_queries"], g["q_ground_truth"], g["q_negative"]
    kw = {
        "q_hr_plain": dict(queries=queries, query_mode="hr"),
        "q_rt_gt": dict(queries=queries[:, ::-1], query_mode="rt", ground_truth=gt),
        "q_hr_neg": dict(
            queries=queries, query_mode="hr", negative=neg, ground_truth=gt
        ),
        "q_hr_type": dict(queries=queries, query_mode="hr", negative_type="b"),
    }[case]
    pts = PartitionedTripleSet.create_from_queries(ds, sharding, **kw)
    p = case + "_"
    for k in ("triples", "triple_counts", "triple_offsets", "triple_sort_idx"):
        exact(getattr(pts, k), g[p + k], p + k)
    assert str(pts.dummy) == str(g[p + "dummy"])
    assert pts.partition_mode == str(g[p + "partition_mode"])
    for k in ("types", "neg_heads", "neg_tails"):
        if p + k in g.files:
            exact(getattr(pts, k), g[p + k], p + k)
        else:
            assert getattr(pts, k) is None


# ------------------------------------------------------------- structural ---
seed = 1234
n_entity = 5000
n_relation_type = 50
n_shard = 7
n_triple = 20000
type_offsets = {"type-0": 0, "type-1": 2000, "type-2": 3000}


@pytest.mark.parametrize("typed", [False, True])
@pytest.mark.parametrize("ns", [1, 4, 7])
def test_sharding_invariants(typed, ns):
    to = np.array(list(type_offsets.values())) if typed else None
    s = Sharding.create(n_entity, ns, seed=seed, type_offsets=to)
    assert s.n_shard == ns and s.n_entity == n_entity
    assert s.max_entity_per_shard == int(np.ceil(n_entity / ns))
    assert s.shard_and_idx_to_entity.shape == (ns, s.max_entity_per_shard)
    # inverse maps round-trip
    assert_equal(
        s.shard_and_idx_to_entity[s.entity_to_shard, s.entity_to_idx],
        np.arange(n_entity),
    )
    assert s.shard_counts.sum() == n_entity
    # rows ascending, padding at the end
    assert np.all(np.diff(s.shard_and_idx_to_entity, axis=1) > 0)
    for k in range(ns):
        assert np.all(s.shard_and_idx_to_entity[k, : s.shard_counts[k]] < n_entity)
        assert np.all(s.shard_and_idx_to_entity[k, s.shard_counts[k] :] >= n_entity)
    if typed:
        assert_equal(s.entity_type_counts.sum(axis=1), s.shard_counts)
        bounds = np.array(list(type_offsets.values()) + [n_entity])
        assert_equal(s.entity_type_counts.sum(axis=0), np.diff(bounds))
        for k in range(ns):
            ids = s.shard_and_idx_to_entity[k]
            for t in range(len(to)):
                lo = s.entity_type_offsets[k, t]
                hi = lo + s.entity_type_counts[k, t]
                assert np.all((ids[lo:hi] >= bounds[t]) & (ids[lo:hi] < bounds[t + 1]))


def test_sharding_save_load(tmp_path):
    for to in (None, np.array([0, 2000, 3000])):
        s = Sharding.create(n_entity, n_shard, seed=seed, type_offsets=to)
        f = tmp_path / "s.npz"
        s.save(f)
        s2 = Sharding.load(f)
        assert s2.n_shard == s.n_shard
        assert_equal(s2.shard_and_idx_to_entity, s.shard_and_idx_to_entity)
        assert_equal(s2.entity_to_idx, s.entity_to_idx)
        if to is None:
            assert s2.entity_type_counts is None
        else:
            assert_equal(s2.entity_type_counts, s.entity_type_counts)


@pytest.mark.parametrize("mode", ["h_shard", "t_shard", "ht_shardpair"])
@pytest.mark.parametrize("inverse", [True, False])
def test_partition_invariants(mode, inverse):
    rng = np.random.default_rng(seed)
    triples = np.stack(
        [
            rng.integers(n_entity, size=n_triple),
            rng.integers(n_relation_type, size=n_triple),
            rng.integers(n_entity, size=n_triple),
        ],
        axis=1,
    )
    neg_h = rng.integers(n_entity, size=(n_triple, 3))
    neg_t = rng.integers(n_entity, size=(n_triple, 3))
    ds = KGDataset(
        n_entity=n_entity,
        n_relation_type=n_relation_type,
        type_offsets=type_offsets,
        triples={"train": triples},
        original_triple_ids={"train": np.arange(n_triple)},
        neg_heads={"train": neg_h},
        neg_tails={"train": neg_t},
    )
    s = Sharding.create(
        n_entity, n_shard, seed=seed, type_offsets=np.array(list(type_offsets.values()))
    )
    pts = PartitionedTripleSet.create_from_dataset(
        ds, "train", s, partition_mode=mode, add_inverse_triples=inverse
    )
    full = triples
    full_nh, full_nt = neg_h, neg_t
    if inverse:
        inv = triples[:, ::-1].copy()
        inv[:, 1] += n_relation_type
        full = np.concatenate([triples, inv])
        full_nh = np.concatenate([neg_h, neg_t])
        full_nt = np.concatenate([neg_t, neg_h])
    assert pts.triple_counts.sum() == full.shape[0]
    assert_equal(
        pts.triple_offsets.flatten(),
        np.concatenate([[0], np.cumsum(pts.triple_counts.flatten())[:-1]]),
    )
    ordered = full[pts.triple_sort_idx]
    # every bucket holds triples whose head/tail live on the bucket's shards
    flat_off = pts.triple_offsets.flatten()
    flat_cnt = pts.triple_counts.flatten()
    for b in range(flat_off.size):
        sl = slice(flat_off[b], flat_off[b] + flat_cnt[b])
        if mode == "ht_shardpair":
            sh, st = divmod(b, n_shard)
        elif mode == "h_shard":
            sh, st = b, None
        else:
            sh, st = None, b
        if sh is not None:
            assert np.all(s.entity_to_shard[ordered[sl, 0]] == sh)
            assert_equal(s.shard_and_idx_to_entity[sh, pts.triples[sl, 0]], ordered[sl, 0])
        else:
            assert_equal(pts.triples[sl, 0], ordered[sl, 0])
        if st is not None:
            assert np.all(s.entity_to_shard[ordered[sl, 2]] == st)
            assert_equal(s.shard_and_idx_to_entity[st, pts.triples[sl, 2]], ordered[sl, 2])
        else:
            assert_equal(pts.triples[sl, 2], ordered[sl, 2])
    assert_equal(pts.triples[:, 1], ordered[:, 1])
    assert_equal(pts.neg_heads, full_nh[pts.triple_sort_idx])
    assert_equal(pts.neg_tails, full_nt[pts.triple_sort_idx])
    bounds = np.array(list(type_offsets.values()))
    assert_equal(pts.types, np.digitize(ordered[:, [0, 2]], bounds) - 1)


def test_partition_bad_mode():
    s = Sharding.create(10, 2, seed=0)
    with pytest.raises(ValueError):
        PartitionedTripleSet.partition_triples(np.zeros((3, 3), dtype=np.int64), s, "x")
