"""Recorded steps hold no memset node, and the index they are built on is right.

On this ROCm a recorded training step whose side branch interleaves MEMSET nodes with kernels is not safe to replay
(round 3: a recorded clear left stale memory behind, Adam moments of 1e20; round 4: a memory-aperture fault in the
rocPRIM Onesweep kernel that reads the look-back state its memset nodes clear - profiles/r04/graph_memset_probe.md).
Every clear of the library is a kernel, the device-wide sort of the segment index is the library's own
(csrc/segments.hip: k_rsort_*), and torch's `zero_()` / `fill_()` record kernels too - so a recorded step is made
of kernel nodes (+ RCCL's), which `bess_graph_node_counts` lets this file assert."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


@pytest.mark.parametrize("n,n_rows", [(15_361, 700), (20_000, 1 << 9), (100_003, 93_773), (1_048_576 + 37, 93_773),
                                      (300_000, 312_576), (70_000, 62_500_000), (65_536, 3), (40_000, (1 << 27) + 5)])
def test_segment_index_is_the_stable_sort_by_row(dev, n, n_rows):
    """bess_build_segment_index above the one-workgroup size: references sorted by row, equal rows in reference
    order (stable: sums over a row's references are bitwise reproducible), unique rows ascending, offsets, count.
    One, two, three and four radix places (row ids of 2 .. 27+ bits)."""
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(n % 1000)
    # skewed ids: a hot row, runs of equal ids, the largest id
    idx = torch.randint(0, n_rows, (n,), generator=gen, dtype=torch.int64)
    idx[: n // 50] = int(idx[0])
    idx[n // 2: n // 2 + 300] = n_rows - 1
    idx = idx.to(torch.int32)
    seg = nat.SegmentIndex(idx.to(dev), n_rows)
    torch.cuda.synchronize()
    order = torch.sort(idx.long(), stable=True).indices
    assert torch.equal(seg.refs.cpu().long(), order)
    rows, counts = torch.unique_consecutive(idx.long()[order], return_counts=True)
    ns = int(seg.n_seg.item())
    assert ns == len(rows)
    assert torch.equal(seg.seg_rows[:ns].cpu().long(), rows)
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)])
    assert torch.equal(seg.seg_offsets[: ns + 1].cpu().long(), offs)
    n_long = int((counts > nat.SEGMENT_CAP).sum())
    assert int(seg.long_segs[0].item()) == n_long
    if n_long:
        got = sorted(seg.long_segs[1: 1 + n_long].cpu().tolist())
        assert got == (counts > nat.SEGMENT_CAP).nonzero().reshape(-1).tolist()


def _recorded(model_fn, S, K, flat, loss, augment, opt, dev, n_shard=1, scheme="t", steps=3):
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.negative_sampler import RandomShardedNegativeSampler

    fn = model_fn()
    sharding = fn.sharding
    n = n_shard
    ns = RandomShardedNegativeSampler(K, sharding, 0, scheme, local_sampling=False, flat_negative_format=flat)
    model = EmbeddingMovingBessKGE(ns, fn, loss, augment_negative=augment)
    runner = runtime.training_model(model, runtime.Options(use_graphs=True, keep_graph=True), opt, device=dev)
    rng = np.random.default_rng(0)
    M = int(sharding.shard_counts.min())
    ppp = S // n
    b = dict(head=rng.integers(M, size=(n, n, ppp)), relation=rng.integers(fn.relation_embedding.shape[0], size=(n, n, ppp)),
             tail=rng.integers(M, size=(n, n, ppp)), negative=rng.integers(M, size=(n, n, 1 if flat else S, K)))
    b = {k: torch.from_numpy(v.astype(np.int32)) for k, v in b.items()}
    for _ in range(steps):
        out = runner(**b)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out["loss"]).all())
    (counts,) = runner.graph_node_counts().values()
    return counts, model


CASES = {
    # the C2 training step (per-triple negatives of the own shard: device-wide index on the side stream), half size
    "c2_sgd": dict(scorer="ComplEx", d=256, rows=93_773, S=2048, K=128, flat=False, loss="ls", augment=False, opt="sgd"),
    "c2_adamw": dict(scorer="ComplEx", d=256, rows=93_773, S=2048, K=128, flat=False, loss="ls", augment=False, opt="adamw"),
    # the C4 notebook micro-batch and a larger point (fp16 shard, shared negatives, augmentation, sampled softmax)
    "c4_512x32": dict(scorer="TransE", d=256, rows=312_576, S=512, K=32, flat=True, loss="ssce", augment=True, opt="sgd", half=True),
    "c4_4096x256": dict(scorer="TransE", d=256, rows=312_576, S=4096, K=256, flat=True, loss="ssce", augment=True, opt="sgd", half=True),
    # two shards in one process, per-triple negatives through the exchange, d_query cleared + atomics (round 3's case)
    "em2_transe_adam": dict(scorer="TransE", d=32, rows=900, S=32, K=64, flat=False, loss="ls", augment=False, opt="adam",
                            n_shard=2, scheme="h"),
    "em2_complex_adam": dict(scorer="ComplEx", d=16, rows=900, S=32, K=6, flat=False, loss="ls", augment=False, opt="adam",
                             n_shard=2, scheme="h"),
}


@pytest.mark.parametrize("name", list(CASES))
def test_recorded_steps_hold_no_memset_node(dev, name):
    from besskge import runtime
    from besskge.loss import LogSigmoidLoss, SampledSoftmaxCrossEntropyLoss
    from besskge.scoring import ComplEx, TransE
    from besskge.sharding import Sharding

    c = CASES[name]
    n = c.get("n_shard", 1)
    sharding = Sharding.create(c["rows"] * n, n, seed=0)
    dtype = torch.float16 if c.get("half") else torch.float32

    def make():
        torch.manual_seed(0)
        if c["scorer"] == "ComplEx":
            return ComplEx(c["flat"], sharding, 51, c["d"], device=dev, dtype=dtype)
        return TransE(c["flat"], 1, sharding, 51, c["d"], device=dev, dtype=dtype)

    loss = (SampledSoftmaxCrossEntropyLoss(n_entity=c["rows"] * n) if c["loss"] == "ssce"
            else LogSigmoidLoss(margin=6.0, negative_adversarial_sampling=True))
    opt = dict(sgd=runtime.SGD(lr=1e-3), adamw=runtime.Adam(lr=1e-3, weight_decay=1e-2), adam=runtime.Adam(lr=1e-2))[c["opt"]]
    counts, _ = _recorded(make, c["S"], c["K"], c["flat"], loss, c["augment"], opt, dev, n_shard=n,
                          scheme=c.get("scheme", "t"))
    assert counts.get("kernel", 0) > 0
    assert "memset" not in counts, f"{name}: recorded step holds memset nodes: {counts}"
    assert set(counts) <= {"kernel", "memcpy", "empty"}, counts


def test_replays_follow_the_eager_trajectory(dev):
    """Six AdamW steps of the C2-shaped model replayed from one recording equal six eager steps (the side-stream
    index build - the library's own sort - re-runs correctly inside every replay)."""
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import ComplEx
    from besskge.sharding import Sharding

    sharding = Sharding.create(50_000, 1, seed=0)
    S, K = 1024, 64
    rng = np.random.default_rng(1)
    batches = []
    for _ in range(6):
        b = dict(head=rng.integers(50_000, size=(1, 1, S)), relation=rng.integers(11, size=(1, 1, S)),
                 tail=rng.integers(50_000, size=(1, 1, S)), negative=rng.integers(50_000, size=(1, 1, S, K)))
        batches.append({k: torch.from_numpy(v.astype(np.int32)) for k, v in b.items()})
    tables = []
    for graphs in (False, True):
        torch.manual_seed(3)
        fn = ComplEx(False, sharding, 11, 64, device=dev)
        ns = RandomShardedNegativeSampler(K, sharding, 0, "t", local_sampling=False, flat_negative_format=False)
        model = EmbeddingMovingBessKGE(ns, fn, LogSigmoidLoss(margin=6.0, negative_adversarial_sampling=True))
        runner = runtime.training_model(model, runtime.Options(use_graphs=graphs), runtime.Adam(lr=1e-2, weight_decay=1e-2),
                                        device=dev)
        for b in batches:
            runner(**b)
        torch.cuda.synchronize()
        tables.append(model.score_fn.entity_embedding.detach().clone())
    off = (tables[0] - tables[1]).abs()
    # (Adam normalises by |g|: a gradient that cancels to ~0 takes a +-lr step whose sign follows the atomics' order)
    assert float((off > 1e-5).float().mean()) < 0.01 and float(off.max()) <= 6 * 2 * 1e-2 * 1.01
