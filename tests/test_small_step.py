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
    # round 4: query + positive score (+ the step's copy / fill jobs and the update's generation in spare workgroups),
    # packed L1 scores, loss rows, both backward products as partial sums (no atomics), query / triple backward (adds
    # the parts up; every gradient row lands in the shard's accumulator at its row id), direct update (+ relation step)
    assert 0 < len(kernels) <= 7, kernels
    assert not any("k_step_prologue" in k or "k_coalesced_update" in k for k in kernels), kernels
    assert any("k_direct_update" in k for k in kernels) and any("k_l1_bwd_parts" in k for k in kernels), kernels


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("opt_name", ["sgd", "sgdm", "adamw"])
def test_direct_update_equals_the_indexed_update(dev, dtype, opt_name):
    """bess_direct_update (gradient rows added into a [M, W] accumulator at their row ids, one wave per reference
    claims its row) against bess_build_segment_index + bess_coalesced_update on the same lists: duplicates inside
    and across lists, several steps (the accumulator and the generation stamps are left consistent)."""
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(7)
    M, W = 5000, 96
    table0 = torch.randn(M, W, generator=gen).to(dtype)
    tabs = [table0.clone().to(dev), table0.clone().to(dev)]
    states = [[torch.zeros(M, W, device=dev) for _ in range(2)] for _ in range(2)]
    scratch = nat.DirectAccumulator(tabs[0])
    rel = [torch.randn(40, W, generator=gen).to(dtype).to(dev) for _ in range(2)]
    rel[1] = rel[0].clone()
    ever = torch.zeros(M, dtype=torch.bool, device=dev)
    for step in range(1, 4):
        lists = [torch.randint(M, (n,), generator=gen, dtype=torch.int32) for n in (600, 37, 600)]
        lists[0][:50] = lists[2][:50]       # rows named by two lists
        lists[1][:] = int(lists[0][3])      # one row, many references
        grads = [torch.randn(len(x), W, generator=gen) * 0.1 for x in lists]
        lists = [x.to(dev) for x in lists]
        grads = [g.to(dev) for g in grads]
        rel_grad = torch.randn(40, W, generator=gen).to(dev)

        def desc():
            o = nat.OptDesc()
            o.kind = nat.OPT_ADAM if opt_name == "adamw" else nat.OPT_SGD
            o.step, o.lr = step, 0.05
            o.momentum = 0.9 if opt_name == "sgdm" else 0.0
            o.beta1, o.beta2, o.eps = 0.9, 0.999, 1e-8
            o.weight_decay = 0.01 if opt_name == "adamw" else 0.0
            return o

        n_state = dict(sgd=0, sgdm=1, adamw=2)[opt_name]
        st = [[s_ if i < n_state else None for i, s_ in enumerate(states[k])] for k in range(2)]
        # indexed
        seg = nat.SegmentIndex(torch.cat(lists), M)
        nat.coalesced_update(desc(), tabs[1], seg, grads, st[1][0], st[1][1], axpy=(rel[1], rel_grad, -0.05))
        # direct: what the backward kernels do (rows added at their ids), the generation bump, the update
        for x, g in zip(lists, grads):
            nat.sparse_sgd_lists(scratch.acc, [(x, g)], -1.0)
        nat.step_prologue([scratch.increment_job()])
        nat.direct_update(desc(), tabs[0], lists, scratch, st[0][0], st[0][1], axpy=(rel[0], rel_grad, -0.05))
        torch.cuda.synchronize()
        assert int(scratch.generation) == step + 1
        assert float(scratch.acc.abs().max()) == 0.0  # left zero for the next step
        tol = dict(rtol=1e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=1e-5, atol=1e-6)
        if opt_name == "adamw" and dtype == torch.float32:
            # (Adam divides by sqrt(v): where a row's gradient nearly cancels, the order of the fp32 additions - atomics
            # here, reference order there - shows a few 1e-6 of the 0.05 step)
            tol = dict(rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(tabs[0].float(), tabs[1].float(), **tol)
        torch.testing.assert_close(rel[0].float(), rel[1].float(), **tol)
        for a, b in zip(states[0][:n_state], states[1][:n_state]):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6 if opt_name != "adamw" else 1e-5)
        ever[torch.unique(torch.cat(lists)).long()] = True
        assert torch.equal(tabs[0][~ever], table0.to(dev)[~ever])  # lazy semantics: untouched rows do not move


@pytest.mark.parametrize("dtype,opt_name", [(torch.float16, "sgd"), (torch.float16, "adam"), (torch.float32, "sgdm")])
@pytest.mark.parametrize("scorer", ["TransE", "RotatE"])
def test_training_step_through_the_direct_update_equals_the_indexed_path(dev, dtype, opt_name, scorer):
    """The whole step both ways (same model, `direct_update_max_bytes = 0` switches the accumulator off): tables
    after three steps agree to the order of the fp32 additions."""
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import RotatE, TransE
    from besskge.sharding import Sharding

    S_, K_, M = 256, 32, 3000
    sharding = Sharding.create(M, 1, seed=0)
    rng = np.random.default_rng(0)
    batches = []
    for _ in range(3):
        b = dict(head=rng.integers(M, size=(1, 1, S_)), relation=rng.integers(9, size=(1, 1, S_)),
                 tail=rng.integers(M, size=(1, 1, S_)), negative=rng.integers(M, size=(1, 1, 1, K_)))
        batches.append({k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in b.items()})
    out = []
    for direct in (True, False):
        torch.manual_seed(1)
        fn = (TransE(True, 1, sharding, 9, 64, device=dev, dtype=dtype) if scorer == "TransE"
              else RotatE(True, 1, sharding, 9, 32, device=dev, dtype=dtype))
        ns = RandomShardedNegativeSampler(K_, sharding, 0, "t", local_sampling=False, flat_negative_format=True)
        model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn, augment_negative=True,
                                       loss_fn=LogSigmoidLoss(margin=4.0, negative_adversarial_sampling=True))
        if not direct:
            model.direct_update_max_bytes = 0
        opt = dict(sgd=runtime.SGD(lr=0.05), sgdm=runtime.SGD(lr=0.05, momentum=0.9), adam=runtime.Adam(lr=0.01))[opt_name]
        runner = runtime.training_model(model, runtime.Options(), opt, device=dev)
        losses = [float(runner(**b)["loss"]) for b in batches]
        torch.cuda.synchronize()
        used = "_direct_acc" in model.__dict__
        assert used == direct, "the direct update was (not) taken"
        out.append((model.score_fn.entity_embedding.detach().float().clone(),
                    model.score_fn.relation_embedding.detach().float().clone(), losses))
    tol = dict(rtol=2e-3, atol=3e-3) if dtype == torch.float16 else dict(rtol=1e-4, atol=1e-5)
    if opt_name == "adam":  # (a gradient that cancels to ~0 takes a +-lr step whose sign follows the order of the additions)
        off = (out[0][0] - out[1][0]).abs()
        assert float((off > 3e-3).float().mean()) < 0.01
    else:
        torch.testing.assert_close(out[0][0], out[1][0], **tol)
        torch.testing.assert_close(out[0][1], out[1][1], **tol)
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=2e-3)



@pytest.mark.parametrize("scorer,dtype,S,N,W", [("TransE", torch.float16, 512, 544, 256), ("TransE", torch.float32, 256, 288, 64),
                                                 ("RotatE", torch.float16, 1024, 1088, 128), ("TransE", torch.float16, 768, 800, 256)])
def test_backward_as_partial_sums_equals_the_atomic_form(dev, scorer, dtype, S, N, W):
    """bess_neg_score_shared_bwd_parts + bess_query_triple_bwd_parts (no atomics in the products: partial sums with plain
    stores, added up by their consumer, every gradient row landing in a row-space accumulator) against
    bess_neg_score_shared_bwd + bess_query_triple_bwd + a scatter of their dense rows."""
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(S + W)
    M, n_rel = 6000, 11
    code = nat.TRANSE if scorer == "TransE" else nat.ROTATE
    Wr = W if scorer == "TransE" else W // 2
    table = (torch.randn(M, W, generator=gen) * 0.3).to(dtype).to(dev)
    rel = (torch.randn(n_rel, Wr, generator=gen) * 0.3).to(dtype).to(dev)
    d = nat.make_desc(code, 1, table, Wr)
    head = nat.RowSource(table, torch.randint(M, (S,), generator=gen, dtype=torch.int32).to(dev))
    tail = nat.RowSource(table, torch.randint(M, (S,), generator=gen, dtype=torch.int32).to(dev))
    ridx = torch.randint(n_rel, (S,), generator=gen, dtype=torch.int32).to(dev)
    nidx = torch.randint(M, (N,), generator=gen, dtype=torch.int32)
    nidx[:40] = nidx[40:80]  # the same entity at several candidate positions
    nidx[100:110] = head.idx[:10].cpu()  # ... and as a head
    neg = nat.RowSource(table, nidx.to(dev))
    q, pos = nat.query_triple_fwd(d, nat.CORRUPT_TAIL, head, tail, rel, ridx)
    out = nat.neg_score_shared_fwd(d, q, neg)
    go = (torch.randn(S, N, generator=gen) * 0.1).to(dev)
    d_pos = torch.randn(S, generator=gen).to(dev)
    n_dq, n_de = nat.shared_bwd_parts_plan(d, S, N)
    assert n_dq >= 1 and n_de >= 1
    # reference: atomic form + dense rows scattered into an accumulator
    dq, dn = nat.neg_score_shared_bwd(d, q, neg, out, go)
    d_rel0 = torch.zeros(n_rel, Wr, device=dev)
    dh, dt = nat.query_triple_bwd(d, nat.CORRUPT_TAIL, head, tail, rel, ridx, d_pos, dq, d_rel0)
    want = torch.zeros(M, W, device=dev)
    nat.sparse_sgd_lists(want, [(neg.idx, dn), (head.idx, dh), (tail.idx, dt)], -1.0)
    # partial sums
    dqp, dep = nat.neg_score_shared_bwd_parts(d, q, neg, go)
    assert tuple(dqp.shape) == (n_dq, S, W) and tuple(dep.shape) == (n_de, N, W)
    torch.testing.assert_close(dqp.sum(0), dq, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dep.sum(0), dn, rtol=1e-4, atol=1e-4)
    acc = torch.zeros(M, W, device=dev)
    d_rel1 = torch.zeros(n_rel, Wr, device=dev)
    nat.query_triple_bwd_parts(d, nat.CORRUPT_TAIL, head, tail, rel, ridx, d_pos, dqp, dep, neg.idx, (acc, acc, acc), d_rel1)
    torch.cuda.synchronize()
    torch.testing.assert_close(acc, want, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(d_rel1, d_rel0, rtol=1e-4, atol=2e-4)
    # shapes without a partial-sum form say so
    small = nat.make_desc(nat.DISTMULT, 0, table, W)
    assert nat.shared_bwd_parts_plan(small, S, N) == (0, 0)
    assert nat.shared_bwd_parts_plan(d, 8, 16) == (0, 0)
    assert nat.shared_bwd_parts_plan(d, 4096, 4352) == (0, 0)  # (the slabs of d_query would be 570 MB: atomic form)


@pytest.mark.parametrize("scorer,dtype,S,N,W,loss", [
    ("ComplEx", torch.float32, 4096 + 3, 64, 128, "logsigmoid_adv"),  # 16 waves per workgroup, ragged last one
    ("TransE", torch.float16, 300, 1028, 64, "ssce"),
    ("RotatE", torch.float32, 257, 2048, 96, "margin_adv"),            # rows kept as 12 chunks per lane
    ("DistMult", torch.float16, 512, 32, 50, "margin"),                # W % 4 != 0
    ("ComplEx", torch.float32, 64, 256, 512, "logsigmoid"),
    ("TransE", torch.float32, 1100, 16, 1024, "ssce"),                 # widest rows of the fused forward: fewer waves
    ("DistMult", torch.float16, 40, 8, 2048, "logsigmoid_adv"),        # per workgroup (their rows share 32 KB of LDS)
])
def test_pertriple_tail_equals_the_four_launches(dev, scorer, dtype, S, N, W, loss):
    """`bess_pertriple_tail` (d loss / d query from the fused forward's partials + K8 + K3' + K6', one launch) against
    `bess_neg_score_pertriple_fwd_dq` + `bess_loss_fwd_bwd` + `bess_query_triple_bwd`: every output to the bit, the
    relation gradient (fp32 atomics in both) to rounding."""
    from besskge import _native as nat
    from besskge._native import RowSource

    g = torch.Generator().manual_seed(S + N)
    M, R = 5000, 11
    sc = dict(TransE=nat.TRANSE, RotatE=nat.ROTATE, DistMult=nat.DISTMULT, ComplEx=nat.COMPLEX)[scorer]
    Wr = W // 2 if scorer == "RotatE" else W
    table = (torch.randn(M, W, generator=g) * 0.3).to(dtype).to(dev)
    rel = (torch.randn(R, Wr, generator=g) * 0.3).to(dtype).to(dev)
    desc = nat.make_desc(sc, 1 if scorer in ("TransE", "RotatE") else 0, table, Wr)
    hi = torch.randint(M, (S,), generator=g, dtype=torch.int32).to(dev)
    ti = torch.randint(M, (S,), generator=g, dtype=torch.int32).to(dev)
    ri = torch.randint(R, (S,), generator=g, dtype=torch.int32).to(dev)
    neg = RowSource(table, torch.randint(M, (S * N,), generator=g, dtype=torch.int32).to(dev))
    w = (torch.rand(S, generator=g) + 0.5).to(dev) if S % 2 else torch.full((1,), 1.0 / S, device=dev)
    ld = nat.LossDesc()
    ld.kind = dict(logsigmoid=nat.LOSS_LOGSIGMOID, margin=nat.LOSS_MARGIN, ssce=nat.LOSS_SSCE)[loss.split("_")[0]]
    ld.margin = 2.0
    ld.adversarial = int(loss.endswith("_adv"))
    ld.adversarial_scale = 0.7
    ld.loss_scale = 3.0
    ld.ssce_shift = float(np.log(M - 1) - np.log(N))
    side = nat.CORRUPT_TAIL
    head, tail = RowSource(table, hi), RowSource(table, ti)
    q, pos = nat.query_triple_fwd(desc, side, head, tail, rel, ri)
    assert nat.pertriple_tail_supported(desc, N)
    # the four launches
    out_a, dq_a = nat.neg_score_pertriple_fwd_dq(desc, ld, q, neg, N, pos, w)
    loss_a, dp_a, dn_a = nat.loss_fwd_bwd(ld, pos, out_a, w, True)
    drel_a = torch.zeros(rel.shape, dtype=torch.float32, device=dev)
    dh_a, dt_a = nat.query_triple_bwd(desc, side, head, tail, rel, ri, dp_a, dq_a, drel_a)
    # forward with the partials left + the tail
    out_b, parts = nat.neg_score_pertriple_fwd_dq(desc, ld, q, neg, N, pos, w, defer=True)
    drel_b = torch.zeros(rel.shape, dtype=torch.float32, device=dev)
    loss_b, dp_b, dn_b, dh_b, dt_b, dq_b = nat.pertriple_tail(desc, ld, side, head, tail, rel, ri, parts, pos, out_b, w,
                                                              drel_b, want_d_query=True)
    torch.cuda.synchronize()
    for a, b, nm in ((out_a, out_b, "scores"), (dq_a, dq_b, "d_query"), (dp_a, dp_b, "d_pos"), (dn_a, dn_b, "d_neg"),
                     (dh_a, dh_b, "d_head"), (dt_a, dt_b, "d_tail")):
        assert torch.equal(a, b), nm
    assert float(loss_a) == float(loss_b) or abs(float(loss_a) - float(loss_b)) <= 1e-6 * abs(float(loss_a))
    torch.testing.assert_close(drel_a, drel_b, rtol=1e-4, atol=1e-4)
    # again: the ticket counter was left at zero, the sum is reproducible
    drel_c = torch.zeros_like(drel_b)
    loss_c = nat.pertriple_tail(desc, ld, side, head, tail, rel, ri, parts, pos, out_b, w, drel_c)[0]
    assert float(loss_c) == float(loss_b)
    with pytest.raises(RuntimeError, match="pertriple_tail"):
        nat.pertriple_tail(desc, ld, side, head, tail, rel, ri, parts, pos, torch.zeros(S, 4100, device=dev), w, drel_c)


@pytest.mark.parametrize("opt_name", ["sgd", "adam"])
@pytest.mark.parametrize("scorer,dtype", [("ComplEx", torch.float32), ("TransE", torch.float16)])
def test_training_step_through_the_pertriple_tail_equals_the_separate_launches(dev, scorer, dtype, opt_name):
    """Per-triple negatives of the own shard (the C2 shape in small): three steps with `pertriple_tail` on and off -
    same losses, same tables (plain SGD on fp32 rows adds with atomics: to rounding)."""
    from besskge import _native as nat
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import ComplEx, TransE
    from besskge.sharding import Sharding

    S_, K_, M = 192, 64, 3000
    sharding = Sharding.create(M, 1, seed=0)
    rng = np.random.default_rng(0)
    batches = []
    for _ in range(3):
        b = dict(head=rng.integers(M, size=(1, 1, S_)), relation=rng.integers(9, size=(1, 1, S_)),
                 tail=rng.integers(M, size=(1, 1, S_)), negative=rng.integers(M, size=(1, 1, S_, K_)))
        batches.append({k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in b.items()})
    out = []
    for tail_on in (True, False):
        torch.manual_seed(1)
        fn = (ComplEx(False, sharding, 9, 32, device=dev, dtype=dtype) if scorer == "ComplEx"
              else TransE(False, 1, sharding, 9, 64, device=dev, dtype=dtype))
        ns = RandomShardedNegativeSampler(K_, sharding, 0, "t", local_sampling=False, flat_negative_format=False)
        model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn,
                                       loss_fn=LogSigmoidLoss(margin=4.0, negative_adversarial_sampling=True))
        model.pertriple_tail = tail_on
        opt = dict(sgd=runtime.SGD(lr=0.05), adam=runtime.Adam(lr=0.01))[opt_name]
        runner = runtime.training_model(model, runtime.Options(), opt, device=dev)
        nat.start_kernel_timing(["bess_neg_score_pertriple_fwd_dq"])
        losses = [float(runner(**b)["loss"]) for b in batches]
        torch.cuda.synchronize()
        calls = {k: len(v) for k, v in nat.stop_kernel_timing().items()}
        assert calls.get("bess_neg_score_pertriple_fwd_dq", 0) == 3, calls  # the fused forward was taken
        out.append((model.score_fn.entity_embedding.detach().float().clone(),
                    model.score_fn.relation_embedding.detach().float().clone(), losses))
    tol = dict(rtol=2e-3, atol=3e-3) if dtype == torch.float16 else dict(rtol=1e-4, atol=1e-5)
    if opt_name == "adam":
        off = (out[0][0] - out[1][0]).abs()
        assert float((off > 3e-3).float().mean()) < 0.01
    else:
        torch.testing.assert_close(out[0][0], out[1][0], **tol)
        torch.testing.assert_close(out[0][1], out[1][1], **tol)
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=1e-5)
