"""Device-side samplers: the numpy PCG64 streams continued in
HBM must reproduce the host samplers - and therefore the reference's goldens in
tests/golden/batch_sampler.npz - bit for bit."""

import numpy as np
import pytest
import torch
from numpy.testing import assert_equal

from besskge.batch_sampler import RandomShardedBatchSampler, RigidShardedBatchSampler
from besskge.dataset import KGDataset
from besskge.negative_sampler import (
    PlaceholderNegativeSampler,
    RandomShardedNegativeSampler,
    TripleBasedShardedNegativeSampler,
    TypeBasedShardedNegativeSampler,
)
from besskge.sharding import PartitionedTripleSet, Sharding

from conftest import load_golden


# ------------------------------------------------------------ CPU: stream model
def test_pcg64_stream_tracks_numpy():
    from besskge.device_sampler import Pcg64Stream

    rng = np.random.default_rng(1234)
    st = Pcg64Stream.from_generator(rng)
    for kind, n in [(32, 9), (32, 1), (64, 5), (32, 2), (32, 7), (64, 1), (32, 1), (32, 1), (32, 1000), (64, 12345),
                    (32, 0), (32, 3)]:
        if kind == 32:
            rng.integers(1 << 31, size=n)
            st.skip32(n)
        else:
            rng.integers(1 << 63, size=n)
            st.skip64(n)
        ref = rng.bit_generator.state
        assert st.state == ref["state"]["state"] and st.inc == ref["state"]["inc"]
        assert st.has_uint32 == ref["has_uint32"]
        if st.has_uint32:
            assert st.uinteger == ref["uinteger"]
    # round trip into a fresh generator continues the same stream
    other = np.random.default_rng(0)
    st.to_generator(other)
    assert_equal(other.integers(1 << 31, size=11), rng.integers(1 << 31, size=11))


def test_pcg64_jump_table_and_draw_model():
    """The kernel's arithmetic, restated with Python ints: jump by set bits of the
    distance, step, XSL-RR output, halves low-then-high, value = half >> 1."""
    from besskge.device_sampler import PCG64_MULT, Pcg64Stream, _pcg64_output

    rng = np.random.default_rng(77)
    st = Pcg64Stream.from_generator(rng)
    table = st.jump_table()
    assert table.shape == (64, 4) and table.dtype == np.uint64
    want = rng.integers(1 << 31, size=4001)
    mask = (1 << 128) - 1

    def state_after(steps):
        s = st.state
        for j in range(64):
            if (steps >> j) & 1:
                a = (int(table[j, 0]) << 64) | int(table[j, 1])
                c = (int(table[j, 2]) << 64) | int(table[j, 3])
                s = (a * s + c) & mask
        return s

    for pos in [0, 1, 2, 3, 100, 101, 2047, 4000]:
        s = state_after(pos // 2)
        s = (s * PCG64_MULT + st.inc) & mask
        x = _pcg64_output(s)
        half = (x & 0xFFFFFFFF) if pos % 2 == 0 else (x >> 32)
        assert half >> 1 == want[pos]
    r2 = np.random.default_rng(78)
    st2 = Pcg64Stream.from_generator(r2)
    w64 = r2.integers(1 << 63, size=50)
    s = st2.state
    for i in range(50):
        s = (s * PCG64_MULT + st2.inc) & mask
        assert _pcg64_output(s) >> 1 == w64[i]


# ------------------------------------------------------------------- GPU part
def _dataset(n_entity, n_rel, triples):
    return KGDataset(n_entity=n_entity, n_relation_type=n_rel, triples={"train": triples},
                     original_triple_ids={"train": np.arange(triples.shape[0])})


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda", 0)


def _same(got, want, key):
    assert got.is_cuda
    want = want if isinstance(want, torch.Tensor) else torch.from_numpy(np.asarray(want))
    g = got.cpu()
    assert g.dtype == want.dtype, (key, g.dtype, want.dtype)
    assert tuple(g.shape) == tuple(want.shape), (key, g.shape, want.shape)
    if g.dtype.is_floating_point:
        torch.testing.assert_close(g, want, rtol=2e-7, atol=0)
    else:
        assert torch.equal(g, want), key


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["h_shard", "t_shard", "ht_shardpair"])
@pytest.mark.parametrize("kind", ["rigid", "random"])
@pytest.mark.parametrize("dup", [False, True])
@pytest.mark.parametrize("scheme, flat", [("h", False), ("ht", True)])
@pytest.mark.parametrize("hrt", [False, True])
def test_device_sampler_reproduces_reference_goldens(dev, mode, kind, dup, scheme, flat, hrt):
    """Same cases as tests/test_batch_sampler.py::test_batch_sampler_golden, but
    the tensors come out of the HIP kernels."""
    from besskge.device_sampler import DeviceBatchSampler

    g = load_golden("batch_sampler")
    seed, n_entity, n_rel, n_shard, n_triple, bps, shard_bs, n_negative = (int(x) for x in g["args"])
    ds = _dataset(n_entity, n_rel, g["triples"])
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding, partition_mode=mode)
    ns = RandomShardedNegativeSampler(n_negative=n_negative, sharding=sharding, seed=seed, corruption_scheme=scheme,
                                      local_sampling=False, flat_negative_format=flat)
    cls = RigidShardedBatchSampler if kind == "rigid" else RandomShardedBatchSampler
    bs = cls(partitioned_triple_set=pts, negative_sampler=ns, shard_bs=shard_bs, batches_per_step=bps, seed=seed,
             hrt_freq_weighting=hrt, weight_smoothing=0.5 if hrt else 0.0, duplicate_batch=dup,
             return_triple_idx=True)
    dbs = DeviceBatchSampler(bs, dev)
    p = f"{mode}_{kind}_{int(dup)}_{scheme}_{int(hrt)}_"
    it = iter(bs.get_dataloader_sampler(shuffle=False))
    idxs = [next(it), next(it)]
    if kind == "rigid":
        idxs.append(list(bs.get_dataloader_sampler(shuffle=False))[-1])
    for j, idx in enumerate(idxs):
        batch = dbs.sample(idx)
        want_keys = sorted(k[len(p + f"b{j}_"):] for k in g.files
                           if k.startswith(p + f"b{j}_") and not k.endswith("_idx")) + ["triple_idx"]
        assert sorted(batch.keys()) == sorted(set(want_keys))
        for k, v in batch.items():
            _same(v, g[p + f"b{j}_{k}"], k)


def _make(n_entity, n_rel, n_shard, n_triple, mode, bps, shard_bs, K, scheme, flat, seed, typed=False, local=False,
          placeholder=False, **kw):
    rng = np.random.default_rng(seed)
    triples = np.stack([rng.integers(n_entity, size=n_triple), rng.integers(n_rel, size=n_triple),
                        rng.integers(n_entity, size=n_triple)], axis=1)
    types = np.array([0, n_entity // 5, n_entity // 2]) if typed else None
    ds = _dataset(n_entity, n_rel, triples)
    rigid = kw.pop("rigid", False)
    sharding = Sharding.create(n_entity, n_shard, seed=seed, type_offsets=types)
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding, partition_mode=mode)

    def build():
        if placeholder:
            ns = PlaceholderNegativeSampler(corruption_scheme=scheme, seed=seed)
        elif typed:
            tt = np.random.default_rng(seed + 3).integers(len(types), size=(pts.triples.shape[0], 2)).astype(np.int32)
            ns = TypeBasedShardedNegativeSampler(triple_types=tt, n_negative=K, sharding=sharding,
                                                 corruption_scheme=scheme, local_sampling=local, seed=seed + 1)
        else:
            ns = RandomShardedNegativeSampler(n_negative=K, sharding=sharding, seed=seed + 1, corruption_scheme=scheme,
                                              local_sampling=local, flat_negative_format=flat)
        cls = RigidShardedBatchSampler if rigid else RandomShardedBatchSampler
        return cls(partitioned_triple_set=pts, negative_sampler=ns, shard_bs=shard_bs, batches_per_step=bps,
                   seed=seed + 2, **kw)

    return build(), build()


CASES = {
    "c2_like": dict(n_entity=9377, n_rel=51, n_shard=1, n_triple=50_000, mode="ht_shardpair", bps=3, shard_bs=512,
                    K=256, scheme="t", flat=False, seed=1),
    "odd_sizes": dict(n_entity=1001, n_rel=7, n_shard=3, n_triple=5_000, mode="ht_shardpair", bps=5, shard_bs=33,
                      K=7, scheme="h", flat=False, seed=2),
    "flat_ht": dict(n_entity=1000, n_rel=7, n_shard=4, n_triple=5_000, mode="ht_shardpair", bps=2, shard_bs=64,
                    K=33, scheme="ht", flat=True, seed=3),
    "h_shard": dict(n_entity=1000, n_rel=7, n_shard=4, n_triple=5_000, mode="h_shard", bps=3, shard_bs=21, K=5,
                    scheme="t", flat=False, seed=4, return_triple_idx=True),
    "typed": dict(n_entity=1200, n_rel=5, n_shard=4, n_triple=4_000, mode="ht_shardpair", bps=3, shard_bs=40, K=9,
                  scheme="ht", flat=False, seed=5, typed=True),
    "typed_local": dict(n_entity=1200, n_rel=5, n_shard=2, n_triple=4_000, mode="t_shard", bps=2, shard_bs=30, K=11,
                        scheme="h", flat=False, seed=6, typed=True, local=True),
    "weights_dup": dict(n_entity=800, n_rel=5, n_shard=2, n_triple=3_000, mode="ht_shardpair", bps=2, shard_bs=48,
                        K=3, scheme="t", flat=True, seed=7, hrt_freq_weighting=True, weight_smoothing=0.25,
                        duplicate_batch=True),
    "placeholder": dict(n_entity=800, n_rel=5, n_shard=2, n_triple=3_000, mode="h_shard", bps=2, shard_bs=48, K=1,
                        scheme="t", flat=True, seed=8, placeholder=True),
    "rigid": dict(n_entity=800, n_rel=5, n_shard=2, n_triple=3_000, mode="ht_shardpair", bps=2, shard_bs=48, K=5,
                  scheme="t", flat=False, seed=9, rigid=True),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_device_sampler_matches_host_stream(dev, case):
    """Several consecutive steps (odd element counts leave a buffered half-word
    between calls), then hand the stream back to the host sampler."""
    from besskge.device_sampler import DeviceBatchSampler

    host, twin = _make(**dict(CASES[case]))
    dbs = DeviceBatchSampler(twin, dev)
    order = list(host.get_dataloader_sampler(shuffle=False))
    for step in range(4):
        idx = order[step % len(order)]
        want = host[idx]
        got = dbs.sample(idx)
        assert list(got.keys()) == list(want.keys())
        for k in want:
            _same(got[k], want[k], f"{case}/{step}/{k}")
    dbs.sync_host()
    idx = order[0]
    a, b = host[idx], twin[idx]
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["odd_sizes", "flat_ht", "typed", "h_shard"])
def test_device_sampler_rank_slices(dev, case):
    """Each rank produces only `[:, rank]` of every tensor by jumping to its own
    part of the common stream."""
    from besskge.device_sampler import DeviceBatchSampler

    cfg = dict(CASES[case])
    n = cfg["n_shard"]
    host, _ = _make(**cfg)
    ranks = []
    for r in range(n):
        _, twin = _make(**cfg)
        ranks.append(DeviceBatchSampler(twin, dev, shards=[r]))
    _, twin = _make(**cfg)
    pair = DeviceBatchSampler(twin, dev, shards=[1, 2]) if n >= 3 else None
    order = list(host.get_dataloader_sampler(shuffle=False))
    for step in range(3):
        want = host[order[step % len(order)]]
        for r in range(n):
            got = ranks[r].sample(order[step % len(order)])
            for k in want:
                _same(got[k], want[k][:, r: r + 1], f"{case}/{step}/{r}/{k}")
        if pair is not None:
            got = pair.sample(order[step % len(order)])
            for k in want:
                _same(got[k], want[k][:, 1:3], f"{case}/{step}/pair/{k}")


TRIPLE_BASED = [  # (scheme, flat, mask_on_gather, return_sort_idx, partition mode, n_shard)
    ("t", False, False, False, "ht_shardpair", 3),
    ("h", False, True, True, "ht_shardpair", 2),
    ("t", True, False, True, "h_shard", 4),
    ("h", True, True, False, "t_shard", 2),
    ("ht", False, False, True, "ht_shardpair", 3),
    ("ht", False, True, False, "h_shard", 2),
    ("ht", True, False, True, "ht_shardpair", 4),
    ("ht", True, True, False, "ht_shardpair", 2),
]


def _triple_based(scheme, flat, on_gather, sort_idx, mode, n_shard, seed=21):
    n_entity, n_rel, n_triple, n_cand = 700, 5, 2400, 37
    rng = np.random.default_rng(seed)
    triples = np.stack([rng.integers(n_entity, size=n_triple), rng.integers(n_rel, size=n_triple),
                        rng.integers(n_entity, size=n_triple)], axis=1)
    ds = _dataset(n_entity, n_rel, triples)
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding, partition_mode=mode)
    n_list = 1 if flat else pts.triples.shape[0]
    crng = np.random.default_rng(seed + 1)
    # skewed candidates: some shards own few of a list's entities, so the padding (and its mask) matters
    heads = (crng.integers(n_entity, size=(n_list, n_cand)) // 3).astype(np.int32)
    tails = crng.integers(n_entity, size=(n_list, n_cand)).astype(np.int32)

    def build():
        ns = TripleBasedShardedNegativeSampler(
            negative_heads=heads if scheme in ("h", "ht") else None, negative_tails=tails if scheme in ("t", "ht") else None,
            sharding=sharding, corruption_scheme=scheme, seed=seed, mask_on_gather=on_gather, return_sort_idx=sort_idx)
        return RigidShardedBatchSampler(partitioned_triple_set=pts, negative_sampler=ns, shard_bs=24, batches_per_step=2,
                                        seed=seed + 2)

    return build(), build()


@pytest.mark.gpu
@pytest.mark.parametrize("scheme,flat,on_gather,sort_idx,mode,n_shard", TRIPLE_BASED)
def test_device_sampler_triple_based_candidate_lists(dev, scheme, flat, on_gather, sort_idx, mode, n_shard):
    """`TripleBasedShardedNegativeSampler` on the device: candidate lists resident in HBM,
    per-step look-up + layout by `bess_gather_candidate_lists`; every tensor equals the host sampler's, and
    a rank asked for its own shard gets the slice it needs."""
    from besskge.device_sampler import DeviceBatchSampler

    host, twin = _triple_based(scheme, flat, on_gather, sort_idx, mode, n_shard)
    dbs = DeviceBatchSampler(twin, dev)
    own = DeviceBatchSampler(twin, dev, shards=[n_shard - 1])
    order = list(host.get_dataloader_sampler(shuffle=False))
    for step in range(3):
        idx = order[step % len(order)]
        want = host[idx]
        got = dbs.sample(idx)
        assert sorted(got.keys()) == sorted(want.keys())
        assert "negative_mask" in want and ("negative_sort_idx" in want) == sort_idx
        for k in want:
            _same(got[k], want[k], f"{scheme}/{flat}/{step}/{k}")
        mine = own.sample(idx)
        for k in want:
            _same(mine[k], want[k][:, n_shard - 1:], f"own/{scheme}/{flat}/{step}/{k}")
    assert not bool(want["negative_mask"].all()) or flat  # the skewed lists do get padded


@pytest.mark.gpu
def test_device_triple_based_sampler_feeds_the_step(dev):
    """Validation against fixed candidate tails (the wikikg2 setup, notebooks/3_wikikg2_fp16.ipynb:962-995):
    ScoreMoving inference fed by the device sampler gives the scores of the run fed by the host sampler."""
    from besskge import runtime
    from besskge.bess import ScoreMovingBessKGE
    from besskge.device_sampler import DeviceBatchSampler
    from besskge.scoring import TransE

    host, twin = _triple_based("t", False, False, False, "ht_shardpair", 2)
    sharding = host.negative_sampler.sharding
    out = []
    for sampler in (host, DeviceBatchSampler(twin, dev)):
        torch.manual_seed(0)
        fn = TransE(False, 1, sharding, 5, 32)
        model = ScoreMovingBessKGE(negative_sampler=host.negative_sampler, score_fn=fn, return_scores=True)
        runner = runtime.inference_model(model, runtime.Options(device_iterations=2), device=dev)
        idx = list(host.get_dataloader_sampler(shuffle=False))[0]
        b = sampler[idx] if sampler is host else sampler.sample(idx)
        res = runner(**{k: v.flatten(end_dim=1) for k, v in b.items() if k in ("head", "relation", "tail", "negative",
                                                                                 "negative_mask", "triple_mask")})
        out.append({k: v.float().cpu() for k, v in res.items()})
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k
    assert float(out[0]["negative_score"].min()) < -40000  # padded candidates were masked
