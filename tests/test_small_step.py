"""Launch-count work for notebook-size training steps (S = 512, K = 32: reference
notebooks/3_wikikg2_fp16.ipynb:251-256): `bess_step_prologue` (copy / fill jobs + the index of several row-id
lists in one launch), the one-launch loss, the pre-cleared targets of the shared backward - each against the
separate launches they replace (bit for bit), and the training step that uses them against the goldens."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


@pytest.mark.parametrize("sizes", [(5,), (512, 32, 512, 512), (1, 2000, 37, 1, 1, 900, 4, 3), (7000, 8000)])
def test_prologue_index_of_several_lists_equals_index_of_their_concatenation(dev, sizes):
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(len(sizes))
    n_rows = 3000
    lists = [torch.randint(n_rows, (n,), generator=gen, dtype=torch.int32).to(dev) for n in sizes]
    if len(sizes) > 2:
        lists[1][: lists[1].numel() // 2] = 17  # a long row (more than BESS_SEGMENT_CAP references)
    want = nat.SegmentIndex(torch.cat(lists), n_rows)
    src = torch.arange(1000, dtype=torch.float32, device=dev)
    dst = torch.full((1000,), -1.0, device=dev)
    zeros = torch.full((777, 3), 5.0, device=dev)
    ints = torch.full((33,), 9, dtype=torch.int32, device=dev)
    got = nat.step_prologue([(dst, src, 0), (zeros, None, 0), (ints, None, 0xFFFFFFFF)], lists, n_rows)
    torch.cuda.synchronize()
    n = int(want.n_seg)
    assert int(got.n_seg) == n and got.n_refs == want.n_refs
    assert torch.equal(got.refs, want.refs)  # stable: equal rows keep reference order
    assert torch.equal(got.seg_rows[:n], want.seg_rows[:n])
    assert torch.equal(got.seg_offsets[: n + 1], want.seg_offsets[: n + 1])
    nl = int(want.long_segs[0])
    assert int(got.long_segs[0]) == nl and sorted(got.long_segs[1: 1 + nl].tolist()) == sorted(want.long_segs[1: 1 + nl].tolist())
    assert torch.equal(dst, src) and float(zeros.abs().max()) == 0.0 and bool((ints == -1).all())


def test_prologue_jobs_only_and_argument_checks(dev):
    from besskge import _native as nat

    a = torch.ones(100_000, device=dev)
    assert nat.step_prologue([(a, None, 0)]) is None
    torch.cuda.synchronize()
    assert float(a.abs().max()) == 0.0
    assert nat.step_prologue([], ()) is None
    with pytest.raises(ValueError):
        nat.step_prologue([(torch.ones(4, dtype=torch.float16, device=dev), None, 0)])
    with pytest.raises(ValueError):
        nat.step_prologue([], [torch.zeros(nat.SMALL_INDEX_MAX + 1, dtype=torch.int32, device=dev)], 10)
    with pytest.raises(ValueError):
        nat.step_prologue([(a, torch.ones(5, device=dev), 0)])


@pytest.mark.parametrize("kind", ["logsigmoid", "margin", "ssce"])
@pytest.mark.parametrize("S,N", [(1, 3), (5, 544), (512, 544), (4096, 4352), (1000, 6145)])
def test_one_launch_loss_is_reproducible_and_matches_the_oracle(dev, kind, S, N):
    """The sum formed by the launch's last workgroup: same bits on every call (fixed order), the oracle's value."""
    from besskge import _native as nat
    from besskge.loss import LogSigmoidLoss, MarginRankingLoss, SampledSoftmaxCrossEntropyLoss
    from oracle import kge

    gen = torch.Generator().manual_seed(S + N)
    pos, neg, w = torch.randn(S, generator=gen), torch.randn(S, N, generator=gen), torch.rand(S, generator=gen)
    fn = dict(logsigmoid=LogSigmoidLoss(margin=2.0, negative_adversarial_sampling=True),
              margin=MarginRankingLoss(margin=1.0, negative_adversarial_sampling=False),
              ssce=SampledSoftmaxCrossEntropyLoss(n_entity=50_000))[kind]
    kw = dict(logsigmoid=dict(kind="logsigmoid", margin=2.0, adversarial=True, adversarial_scale=1.0),
              margin=dict(kind="margin", margin=1.0, adversarial=False), ssce=dict(kind="ssce", n_entity=50_000))[kind]
    ld = fn.kernel_desc(N)
    outs = [nat.loss_fwd_bwd(ld, pos.to(dev), neg.to(dev), w.to(dev), True) for _ in range(4)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[2], outs[0][2])
    p, n_ = pos.clone().requires_grad_(True), neg.clone().requires_grad_(True)
    want = kge.loss_value(pos=p, neg=n_, w=w, **kw)
    want.backward()
    torch.testing.assert_close(outs[0][0].cpu(), want.detach(), rtol=2e-5, atol=1e-4)
    torch.testing.assert_close(outs[0][1].cpu(), p.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(outs[0][2].cpu(), n_.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("S,N,W", [(512, 544, 256), (64, 70, 32), (300, 1000, 128)])
def test_shared_backward_into_a_precleared_buffer(dev, dtype, S, N, W):
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(0)
    table = torch.randn(2000, W, generator=gen).to(dtype).to(dev)
    q = torch.randn(S, W, generator=gen).to(dev)
    idx = torch.randint(2000, (N,), generator=gen, dtype=torch.int32).to(dev)
    d = nat.make_desc(nat.TRANSE, 1, table, W)
    src = nat.RowSource(table, idx)
    out = nat.neg_score_shared_fwd(d, q, src)
    go = torch.randn(S, N, generator=gen).to(dev)
    dq0, dn0 = nat.neg_score_shared_bwd(d, q, src, out, go)
    buf = nat.shared_bwd_buffer(d, S, N, dev)
    buf.fill_(123.0)
    nat.step_prologue([(buf, None, 0)])
    dq1, dn1 = nat.neg_score_shared_bwd(d, q, src, out, go, prezeroed=buf)
    torch.cuda.synchronize()
    assert dq1.data_ptr() == buf.data_ptr()
    # (partial sums meet in fp32 atomics: the order differs from call to call)
    torch.testing.assert_close(dq1, dq0, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(dn1, dn0, rtol=1e-5, atol=1e-4)


def test_training_step_dispatch_count_c4_notebook_shape(dev):
    """The C4 notebook micro-batch (TransE fp16, S = 512, K = 32, augmentation, sampled softmax, SGD) steps in at
    most 10 dispatches, torch's included (prologue, query + positive score, packed L1 scores, loss, both backward
    products, query / triple backward, update, relation update; rocprofv3's view is `profiles/r03/step_r03_c4s.txt`):
    counted here with the torch profiler."""
    from besskge import _native as nat
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import SampledSoftmaxCrossEntropyLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import TransE
    from besskge.sharding import Sharding

    S_, K_, M = 512, 32, 20_000
    sharding = Sharding.create(M, 1, seed=0)
    torch.manual_seed(0)
    fn = TransE(True, 1, sharding, 50, 256, device=dev, dtype=torch.float16)
    ns = RandomShardedNegativeSampler(K_, sharding, 0, "t", local_sampling=False, flat_negative_format=True)
    model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn, augment_negative=True,
                                   loss_fn=SampledSoftmaxCrossEntropyLoss(n_entity=M))
    rng = np.random.default_rng(0)
    batch = dict(head=rng.integers(M, size=(1, 1, S_)), relation=rng.integers(50, size=(1, 1, S_)),
                 tail=rng.integers(M, size=(1, 1, S_)), negative=rng.integers(M, size=(1, 1, 1, K_)))
    batch = {k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in batch.items()}
    runner = runtime.training_model(model, runtime.Options(), runtime.SGD(lr=1e-3), device=dev)
    runner(**batch)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        runner(**batch)
        torch.cuda.synchronize()
    kernels = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    assert 0 < len(kernels) <= 10, kernels
