"""BASELINE.json configs as GPU parity cases (real widths, dtypes, schemes and
shard counts; entity / triple counts scaled so the CPU oracle finishes in
seconds), plus size-independent properties at the full C2 size.

  C1  synthetic 10k entities / 100k triples, TransE d=128 p=1 fp32, n_shard=1,
      RandomShardedBatchSampler S=512, flat K=64 't', LogSigmoid(12, adversarial)
  C2  ogbl-biokg-shaped ComplEx d=256 fp32, n_shard=1, per-triple and shared negatives
  C3  YAGO3-10-shaped RotatE d=200 p=1 fp32, n_shard=2, 'ht', sharing, K=1 per triple
  C4  ogbl-wikikg2-shaped TransE d=256 p=1 **fp16**, n_shard=8, flat K=32 't',
      augment_negative, SampledSoftmaxCrossEntropyLoss (3_wikikg2_fp16 notebook setup)
  C5  DistMult d=512 fp32, n_shard=8, per-triple K and flat K

Each case runs the product (real samplers -> runtime runner -> HIP kernels,
all replicas in lock-step on the one GPU) and compares scores, loss and one
sparse-SGD step with the CPU oracle's autograd.
"""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import kge  # noqa: E402

RTOL, ATOL = 1e-4, 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def run_config(dev, scorer, p, d, dtype, n_shard, n_entity, n_rel, n_triple, shard_bs, K, scheme, flat, sharing,
               augment, loss_name, model="EmbeddingMoving", sampler="random", train=True, seed=1234):
    from besskge import runtime
    from besskge.batch_sampler import RandomShardedBatchSampler, RigidShardedBatchSampler
    from besskge.bess import EmbeddingMovingBessKGE, ScoreMovingBessKGE
    from besskge.dataset import KGDataset
    from besskge.loss import LogSigmoidLoss, MarginRankingLoss, SampledSoftmaxCrossEntropyLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import ComplEx, DistMult, RotatE, TransE
    from besskge.sharding import PartitionedTripleSet, Sharding

    rng = np.random.default_rng(0)
    triples = np.stack([rng.integers(n_entity, size=n_triple), rng.integers(n_rel, size=n_triple),
                        rng.integers(n_entity, size=n_triple)], axis=1)
    ds = KGDataset(n_entity=n_entity, n_relation_type=n_rel, triples={"train": triples},
                   original_triple_ids={"train": np.arange(n_triple)})
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding)
    ns = RandomShardedNegativeSampler(K, sharding, seed, scheme, local_sampling=False, flat_negative_format=flat)
    bcls = RandomShardedBatchSampler if sampler == "random" else RigidShardedBatchSampler
    bs = bcls(pts, ns, shard_bs, 1, seed)
    torch.manual_seed(0)
    W = 2 * d if scorer in ("RotatE", "ComplEx") else d
    Wr = 2 * d if scorer == "ComplEx" else d
    # parity runs use randn tables (scores O(10), as the reference's tests do)
    ent = torch.randn(n_shard, sharding.max_entity_per_shard, W)
    rel = torch.randn(n_rel, Wr)
    if dtype == torch.float16:
        ent, rel = ent.half().float(), rel.half().float()
    ctor = dict(TransE=lambda: TransE(sharing, p, sharding, n_rel, d, ent, rel),
                RotatE=lambda: RotatE(sharing, p, sharding, n_rel, d, ent, rel),
                DistMult=lambda: DistMult(sharing, sharding, n_rel, d, ent, rel),
                ComplEx=lambda: ComplEx(sharing, sharding, n_rel, d, ent, rel))[scorer]
    fn = ctor()
    if loss_name == "logsigmoid":
        loss_fn, lkw = LogSigmoidLoss(12.0, True, 1.0), dict(kind="logsigmoid", margin=12.0, adversarial=True, adversarial_scale=1.0)
    elif loss_name == "margin":
        loss_fn, lkw = MarginRankingLoss(2.0, False), dict(kind="margin", margin=2.0)
    else:
        loss_fn, lkw = SampledSoftmaxCrossEntropyLoss(n_entity), dict(kind="ssce", n_entity=n_entity)
    cls = EmbeddingMovingBessKGE if model == "EmbeddingMoving" else ScoreMovingBessKGE
    m = cls(ns, fn, loss_fn, return_scores=True, augment_negative=augment)
    batch = bs[next(iter(bs.get_dataloader_sampler(shuffle=False)))]
    keys = ("head", "relation", "tail", "negative")
    flat_batch = {k: batch[k].flatten(end_dim=1) for k in keys}

    spec = kge.StepSpec(scorer, p, sharing, scheme, flat, augment=augment)
    t0 = ent.clone().requires_grad_(True)
    r0 = rel.clone().requires_grad_(True)
    # fp16 tables, TransE / RotatE with p = 1, shared negatives: the packed-fp16 kernels round the query to
    # fp16 before it meets the candidates (the reference's fp16 mode); the oracle does the same
    half_query = dtype == torch.float16 and scorer in ("TransE", "RotatE") and p == 1 and sharing and W % 32 == 0
    if os.environ.get("BESS_TEST_FP32_MATH", "0") == "1":  # same configs with the packed-fp16 kernels switched off
        fn.fp32_math, half_query = True, False
    import contextlib
    with (kge.half_queries() if half_query else contextlib.nullcontext()):
        want = kge.bess_step(spec, model, t0, r0, {k: batch[k][0] for k in keys}, lkw)
        if train:
            torch.stack(want["loss"]).sum().backward()

    lr = 0.05
    if train:
        runner = runtime.training_model(m, optimizer=runtime.SGD(lr=lr), device=dev, dtype=dtype)
    else:
        runner = runtime.inference_model(m, device=dev, dtype=dtype)
    res = runner(**flat_batch)
    S = want["positive_score"][0].shape[0]
    out_tol = dict(rtol=RTOL, atol=ATOL) if dtype == torch.float32 else dict(rtol=2e-3, atol=2e-2)
    shift = float(np.log(n_entity - 1) - np.log(want["negative_score"][0].shape[1])) \
        if (loss_name == "ssce" and dtype == torch.float32) else 0.0
    pos = res["positive_score"].float().cpu().reshape(n_shard, S)
    neg = res["negative_score"].float().cpu().reshape(n_shard, S, -1)
    scale = 2e-6 * float(torch.stack(want["negative_score"]).detach().abs().clamp(max=1e4).max())
    for r in range(n_shard):
        torch.testing.assert_close(pos[r], want["positive_score"][r].detach(), rtol=out_tol["rtol"],
                                   atol=max(out_tol["atol"], scale))
        torch.testing.assert_close(neg[r], want["negative_score"][r].detach() + shift, rtol=out_tol["rtol"],
                                   atol=max(out_tol["atol"], scale))
        torch.testing.assert_close(res["loss"].cpu().reshape(n_shard)[r], want["loss"][r].detach(), rtol=2e-4, atol=1e-3)
    if train:
        got_ent = m.score_fn.entity_embedding.detach().float().cpu()
        got_rel = m.score_fn.relation_embedding.detach().float().cpu()
        if dtype == torch.float32:
            torch.testing.assert_close(got_ent, ent - lr * t0.grad, rtol=1e-4, atol=2e-5)
            torch.testing.assert_close(got_rel, rel - lr * r0.grad, rtol=1e-4, atol=5e-5)
        else:
            # fp16 shard: contributions are coalesced per unique row in fp32 and the row is written once,
            # so every updated element is fp16(row - lr * fp32 gradient): within ONE fp16 ulp of the oracle's
            # value (one ulp, not zero: the two fp32 gradients differ in their last bits, which can move a
            # value across a rounding boundary), and untouched rows are bit-identical
            want_ent = (ent - lr * t0.grad).half().float()
            mag = want_ent.abs().clamp(min=2.0 ** -14)
            ulp = torch.exp2(torch.floor(torch.log2(mag)) - 10)
            err = (got_ent - want_ent).abs()
            # one fp16 ulp on top of the tolerance the fp32 tables are held to (rtol 1e-4, atol 2e-5: where row and
            # update nearly cancel, the fp32 difference of the two gradient sums is not small against the ulp of
            # the tiny result)
            slack = 1e-4 * want_ent.abs() + 2e-5
            bad = err > ulp * 1.001 + slack
            assert not bool(bad.any()), (int(bad.sum()), float((err / ulp).max()))
            assert float((err > 0).float().mean()) < 0.02  # and almost all of them are exactly that value
            untouched = t0.grad.abs().sum(-1) == 0
            assert torch.equal(got_ent[untouched], ent[untouched])
            want_rel = (rel - lr * r0.grad).half().float()
            ulp_r = torch.exp2(torch.floor(torch.log2(want_rel.abs().clamp(min=2.0 ** -14))) - 10)
            assert bool(((got_rel - want_rel).abs() <= ulp_r * 1.001 + 1e-4 * want_rel.abs() + 5e-5).all())
    return res


def test_c1_transe_cpu_reference_config(dev):
    run_config(dev, "TransE", 1, 128, torch.float32, 1, 10_000, 20, 100_000, 512, 64, "t", True, True, False, "logsigmoid")


@pytest.mark.parametrize("regime", ["per-triple", "shared"])
def test_c2_biokg_complex(dev, regime):
    if regime == "per-triple":
        run_config(dev, "ComplEx", 0, 256, torch.float32, 1, 20_000, 51, 50_000, 256, 64, "t", False, False, False, "logsigmoid")
    else:
        run_config(dev, "ComplEx", 0, 256, torch.float32, 1, 20_000, 51, 50_000, 256, 384, "t", True, True, False, "logsigmoid")


@pytest.mark.parametrize("model", ["EmbeddingMoving", "ScoreMoving"])
def test_c3_yago_rotate_two_shards(dev, model):
    run_config(dev, "RotatE", 1, 200, torch.float32, 2, 12_000, 37, 40_000, 256, 1, "ht", False, True, False, "logsigmoid",
               model=model, sampler="rigid")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_c4_wikikg2_transe_eight_shards(dev, dtype):
    run_config(dev, "TransE", 1, 256, dtype, 8, 40_000, 535, 200_000, 512, 32, "t", True, True, True, "ssce")


@pytest.mark.parametrize("regime", ["per-triple", "flat"])
def test_c5_distmult_eight_shards(dev, regime):
    if regime == "per-triple":
        run_config(dev, "DistMult", 0, 512, torch.float32, 8, 30_000, 100, 100_000, 128, 8, "h", False, False, False, "logsigmoid",
                   model="ScoreMoving")
    else:
        # (smooth loss: a hinge has knife-edge cases where 1[c > 0] flips with the summation order)
        run_config(dev, "DistMult", 0, 512, torch.float32, 8, 30_000, 100, 100_000, 256, 64, "h", True, True, False, "logsigmoid")


# ------------------------------------------------- full-size C2 properties ---
def test_c2_full_size_properties(dev):
    """S=4096 x K=256 per-triple negatives, 93,773 x 512 fp32 table (the bench
    launch).  Size-independent checks instead of a full CPU recomputation:
    permutation equivariance (bit-exact), duplicate consistency (bit-exact),
    exact scaling by 2 (linearity of the bilinear score), agreement of the
    per-triple and the shared (MFMA) kernels, and a sampled oracle check."""
    from besskge import _native as nat
    from besskge._native import RowSource

    g = torch.Generator().manual_seed(0)
    M, W, S, K = 93_773, 512, 4096, 256
    table = torch.randn(M, W, generator=g).to(dev)
    q = torch.randn(S, W, generator=g).to(dev)
    idx = torch.randint(M, (S, K), generator=g, dtype=torch.int32).to(dev)
    d = nat.make_desc(nat.COMPLEX, 0, table, W)
    out = nat.neg_score_pertriple_fwd(d, q, RowSource(table, idx.reshape(-1)), K)

    perm = torch.randperm(K, generator=g).to(dev)
    out_p = nat.neg_score_pertriple_fwd(d, q, RowSource(table, idx[:, perm].reshape(-1).contiguous()), K)
    assert torch.equal(out_p, out[:, perm])

    idx_dup = idx.clone()
    idx_dup[:, 1::2] = idx_dup[:, 0::2]
    out_d = nat.neg_score_pertriple_fwd(d, q, RowSource(table, idx_dup.reshape(-1)), K)
    assert torch.equal(out_d[:, 1::2], out_d[:, 0::2])

    out2 = nat.neg_score_pertriple_fwd(d, 2 * q, RowSource(table, idx.reshape(-1)), K)
    assert torch.equal(out2, 2 * out)

    # the shared-negative kernel on the same rows (first 64 queries against query 0's negatives)
    sh = nat.neg_score_shared_fwd(d, q[:64].contiguous(), RowSource(table, idx[0].contiguous()))
    torch.testing.assert_close(sh[0], out[0], rtol=1e-4, atol=2e-4)

    rows = torch.randint(S, (64,), generator=g)
    want = torch.einsum("sw,skw->sk", q[rows.to(dev)].cpu().double(), table.cpu()[idx[rows.to(dev)].cpu().long()].double())
    torch.testing.assert_close(out[rows.to(dev)].cpu().double(), want, rtol=1e-4, atol=2e-4)

    # distance scorer at the same size: translation invariance of the p-norm, exact for powers of two
    dt = nat.make_desc(nat.TRANSE, 1, table, W)
    o1 = nat.neg_score_pertriple_fwd(dt, q, RowSource(table, idx.reshape(-1)), K)
    o2 = nat.neg_score_pertriple_fwd(dt, 4 * q, RowSource((4 * table).contiguous(), idx.reshape(-1)), K)
    assert torch.equal(o2, 4 * o1)
    assert bool((o1 <= 0).all())


def test_c5_addressing_beyond_4gib(dev):
    """Config 5 is a 128 GB shard: row offsets must be 64-bit.  A 6.4 GB shard
    (3.1 M rows x 2 KiB) is filled with a pattern that identifies every row; rows
    beyond the 2^32-byte boundary are gathered, scored (per-triple and shared
    kernels), updated (segmented K9/K10 and atomic SGD) and checked."""
    from besskge import _native as nat
    from besskge._native import RowSource

    M, W = 3_100_000, 512
    table = torch.empty(M, W, dtype=torch.float32, device=dev)
    rid = torch.arange(M, device=dev, dtype=torch.float32)
    col = torch.arange(W, device=dev, dtype=torch.float32)
    for lo in range(0, M, 500_000):  # row r, column c holds (r mod 4099) / 4099 + c / 1024
        hi = min(M, lo + 500_000)
        table[lo:hi] = (rid[lo:hi, None] % 4099) / 4099 + col[None, :] / 1024
    g = torch.Generator().manual_seed(1)
    first_high = (1 << 32) // (W * 4) + 1  # first row that starts beyond 4 GiB
    idx = torch.cat([torch.randint(first_high, M, (2000,), generator=g), torch.tensor([M - 1, first_high, 0])]).to(torch.int32)

    def rows_cpu(i):
        i = i.float()
        return (i[:, None] % 4099) / 4099 + torch.arange(W, dtype=torch.float32)[None, :] / 1024

    got = nat.gather_rows(table, idx.to(dev))
    # rows differ by >= 1/4099 = 2.4e-4: 1e-6 identifies the row (and absorbs a 1-ulp division difference)
    torch.testing.assert_close(got.cpu(), rows_cpu(idx), rtol=0, atol=1e-6)
    S, K = 50, 40
    q = torch.randn(S, W, generator=g)
    nidx = idx[: S * K].reshape(S, K)
    d = nat.make_desc(nat.DISTMULT, 0, table, W)
    out = nat.neg_score_pertriple_fwd(d, q.to(dev), RowSource(table, nidx.reshape(-1).to(dev)), K)
    want = torch.einsum("sw,skw->sk", q.double(), rows_cpu(nidx.reshape(-1)).reshape(S, K, W).double())
    torch.testing.assert_close(out.cpu().double(), want, rtol=1e-5, atol=1e-3)
    sh = nat.neg_score_shared_fwd(d, q.to(dev), RowSource(table, idx[:300].to(dev)))
    torch.testing.assert_close(sh.cpu().double(), q.double() @ rows_cpu(idx[:300]).double().T, rtol=1e-5, atol=1e-3)
    # a contiguous window that straddles the boundary (the TopK / AllScores access pattern)
    w0 = first_high - 40
    win = nat.neg_score_shared_fwd(d, q.to(dev), RowSource(table[w0: w0 + 100]))
    torch.testing.assert_close(win.cpu().double(), q.double() @ rows_cpu(torch.arange(w0, w0 + 100)).double().T,
                               rtol=1e-5, atol=1e-3)
    # updates land in the right rows
    go = torch.randn(S, K, generator=g)
    seg = nat.SegmentIndex(nidx.reshape(-1).to(dev), M)
    before = nat.gather_rows(table, idx.to(dev)).cpu()
    nat.neg_pertriple_grad_segments(d, q.to(dev), table, K, go.to(dev), seg, fused_sgd_lr=0.5)
    uniq, inv = torch.unique(nidx.reshape(-1).long(), return_inverse=True)
    gsum = torch.zeros(uniq.numel(), W, dtype=torch.float64).index_add_(
        0, inv, (go.reshape(-1, 1).double() * q.double().repeat_interleave(K, dim=0)))
    after = nat.gather_rows(table, uniq.to(torch.int32).to(dev)).cpu()
    torch.testing.assert_close(after.double(), rows_cpu(uniq).double() - 0.5 * gsum, rtol=1e-5, atol=1e-4)
    untouched = ~torch.isin(idx.long(), uniq)
    assert torch.equal(nat.gather_rows(table, idx.to(dev)).cpu()[untouched], before[untouched])
    nat.sparse_sgd(table, torch.tensor([M - 1], dtype=torch.int32, device=dev), torch.ones(1, W, device=dev), 1.0)
    last = nat.gather_rows(table, torch.tensor([M - 1], dtype=torch.int32, device=dev)).cpu()
    base_last = rows_cpu(torch.tensor([M - 1]))
    if (uniq == M - 1).any():
        base_last = base_last - 0.5 * gsum[uniq == M - 1].float()
    torch.testing.assert_close(last, base_last - 1.0, rtol=1e-5, atol=1e-4)
