"""Packed-fp16 L1 distance matrix (csrc/l1_f16.hip): TransE / RotatE, p = 1, fp16 tables, shared negatives.

Oracle for this path = the reference's fp16 mode restated: the query is an fp16 tensor when it meets
the candidates (`model.half()`; reference scoring.py:194-197, 342), everything after that in full
precision.  Checked here in float64 on fp16-rounded inputs:

  forward   out[q, j] = -sum_w |fp16(Q[q, w]) - E[j, w]|                  (tolerance: fp32 accumulation)
  backward  dQ[a, w] = -sum_b g[a, b] sgn(fp16(Q[a, w]) - E[b, w]),  sgn(0) = 0
            dE[b, w] = +sum_a g[a, b] sgn(fp16(Q[a, w]) - E[b, w])         (fp32 coefficients and sums)
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def desc(nat, W, fp32_math=False):
    d = nat.ModelDesc()
    d.scorer, d.norm_p, d.dtype, d.width, d.rel_width = nat.TRANSE, 1, nat.F16, W, W
    if fp32_math:
        d.reserved[0] = nat.FLAG_FP32_MATH
    return d


def problem(S, N, W, M, scale, seed, ties=False):
    g = torch.Generator().manual_seed(seed)
    Q = torch.randn(S, W, generator=g) * scale
    E = (torch.randn(M, W, generator=g) * scale).half()
    idx = torch.randint(0, M, (N,), generator=g, dtype=torch.int32)
    if ties:  # a quarter of the query entries equal the candidate they meet in column b = a % N
        sel = torch.rand(S, W, generator=g) < 0.25
        rows = E[idx.long()[torch.arange(S) % N]].float()
        Q = torch.where(sel, rows, Q)
    return Q, E, idx


@pytest.mark.parametrize("S,N,W,scale", [(64, 64, 32, 1.0), (70, 130, 96, 1.0), (256, 300, 256, 0.01),
                                         (33, 17, 64, 3e-5), (128, 64, 512, 30.0),
                                         (1000, 900, 64, 1.0)])  # 240 tiles: the 64-row tile; the others the 32-row one
def test_forward_matches_the_fp16_query_oracle(dev, S, N, W, scale):
    from besskge import _native as nat

    Q, E, idx = problem(S, N, W, 500, scale, 1)
    got = nat.neg_score_shared_fwd(desc(nat, W), Q.to(dev), nat.RowSource(E.to(dev), idx.to(dev))).cpu()
    q16 = Q.half().double()
    e = E[idx.long()].double()
    want = -(q16[:, None, :] - e[None, :, :]).abs().sum(-1)
    mag = q16.abs().sum(-1)[:, None] + e.abs().sum(-1)[None, :]  # the three fp32 sums that are combined
    err = (got.double() - want).abs()
    assert bool((err <= 4e-7 * mag + 1e-30).all()), float((err / mag).max())
    # subnormal fp16 operands (scale 3e-5) take part at full precision: nothing is flushed
    if scale < 1e-4:
        assert float(want.abs().min()) > 0 and float((err / want.abs()).max()) < 1e-5
    # identity indexing, and the fp32-math switch gives the unrounded-query scores
    got_id = nat.neg_score_shared_fwd(desc(nat, W), Q.to(dev), nat.RowSource(E[:N].contiguous().to(dev), None)).cpu()
    want_id = -(q16[:, None, :] - E[:N].double()[None, :, :]).abs().sum(-1)
    assert bool(((got_id.double() - want_id).abs() <= 4e-7 * (q16.abs().sum(-1)[:, None] + E[:N].double().abs().sum(-1)[None, :]) + 1e-30).all())
    got32 = nat.neg_score_shared_fwd(desc(nat, W, fp32_math=True), Q.to(dev), nat.RowSource(E.to(dev), idx.to(dev))).cpu()
    want32 = -(Q.double()[:, None, :] - e[None, :, :]).abs().sum(-1)
    torch.testing.assert_close(got32.double(), want32, rtol=1e-5, atol=1e-5 * float(mag.max()))


@pytest.mark.parametrize("ppp", [16, 512])  # 32-row / 64-row tile kernels
@pytest.mark.parametrize("variant", ["augment", "augment_ht", "mask1", "mask2_ht", "mask_rows", "augment_mask"])
def test_forward_with_the_kill_in_its_epilogue_equals_mask_scores(dev, variant, ppp):
    """bess_neg_score_shared_fwd_masked == scores, then bess_mask_scores (bit for bit)."""
    from besskge import _native as nat

    n, K, W = 2, 24 if ppp == 16 else 420, 64
    S = n * ppp
    ht = variant.endswith("ht")
    N = (S // 2 if ht else S) + n * K if variant.startswith("augment") else n * K
    Q, E, idx = problem(S, N, W, 300, 1.0, 5)
    g = torch.Generator().manual_seed(9)
    mask = None
    if "mask" in variant:
        rows = {"mask1": 1, "mask2_ht": 2, "mask_rows": S, "augment_mask": 1}[variant]
        mask = torch.rand(rows, n * K, generator=g) < 0.7
    diag = 1 if variant.startswith("augment") else 0
    kill = (diag, ht, ppp, mask.to(dev) if mask is not None else None)
    src = nat.RowSource(E.to(dev), idx.to(dev))
    fused = nat.neg_score_shared_fwd(desc(nat, W), Q.to(dev), src, kill=kill)
    plain = nat.neg_score_shared_fwd(desc(nat, W), Q.to(dev), src)
    nat.mask_scores(plain, diag, ht, ppp, kill[3])
    assert torch.equal(fused, plain)
    assert int((fused < -40000).sum()) > 0
    # a scorer without the fused epilogue takes the two-step route behind the same entry point
    fused32 = nat.neg_score_shared_fwd(desc(nat, W, fp32_math=True), Q.to(dev), src, kill=kill)
    plain32 = nat.neg_score_shared_fwd(desc(nat, W, fp32_math=True), Q.to(dev), src)
    nat.mask_scores(plain32, diag, ht, ppp, kill[3])
    assert torch.equal(fused32, plain32)


def exact_backward(Q, E, idx, g):
    q16, e = Q.half().double(), E[idx.long()].double()
    sg = torch.sign(q16[:, None, :] - e[None, :, :])  # [S, N, W], 0 at ties
    dq = -(g.double()[:, :, None] * sg).sum(1)
    de = (g.double()[:, :, None] * sg).sum(0)
    return dq, de


@pytest.mark.parametrize("S,N,W,scale,ties", [(64, 64, 32, 1.0, False), (70, 130, 96, 1.0, True),
                                             (200, 1100, 256, 0.01, False), (512, 96, 64, 1.0, True),
                                             # one launch for both products (k_l1_bwd_both): four-wave workgroups with
                                             # a last tile of one wave, columns past the end, eight-wave workgroups
                                             (256, 288, 64, 1.0, True), (512, 544, 256, 0.01, False),
                                             (264, 800, 96, 1.0, True), (1024, 2080, 128, 1.0, False)])
def test_backward_differentiates_the_rounded_query_function(dev, S, N, W, scale, ties):
    """The backward of the packed forward (fp32 kernel, query rounded to fp16 as it is loaded):
    sgn(fp16(q) - e) exactly, 0 at ties; fp32 coefficients and accumulation."""
    from besskge import _native as nat

    Q, E, idx = problem(S, N, W, 400, scale, 2, ties=ties)
    gen = torch.Generator().manual_seed(4)
    g = torch.softmax(torch.randn(S, N, generator=gen) * 3, dim=-1) * torch.rand(S, 1, generator=gen)
    d = desc(nat, W)
    src = nat.RowSource(E.to(dev), idx.to(dev))
    out = nat.neg_score_shared_fwd(d, Q.to(dev), src)
    dq, dn = nat.neg_score_shared_bwd(d, Q.to(dev), src, out, g.to(dev))
    want_q, want_e = exact_backward(Q, E, idx, g)
    if ties:
        assert float((Q.half()[:, None, :] == E[idx.long()][None, :, :]).float().mean()) > 1e-3
    torch.testing.assert_close(dq.cpu().double(), want_q, rtol=1e-5, atol=1e-6 * float(g.sum(1).max()))
    torch.testing.assert_close(dn.cpu().double(), want_e, rtol=1e-5, atol=1e-6 * float(g.sum(0).max()))
    # with the fp32-math switch the query is NOT rounded: signs follow q, not fp16(q)
    d32 = desc(nat, W, fp32_math=True)
    dq32, _ = nat.neg_score_shared_bwd(d32, Q.to(dev), src, nat.neg_score_shared_fwd(d32, Q.to(dev), src), g.to(dev))
    sg32 = torch.sign(Q.double()[:, None, :] - E[idx.long()].double()[None, :, :])
    torch.testing.assert_close(dq32.cpu().double(), -(g.double()[:, :, None] * sg32).sum(1), rtol=1e-5,
                               atol=1e-6 * float(g.sum(1).max()))


def test_transe_fp16_step_against_the_fp16_query_oracle(dev):
    """EmbeddingMoving, flat shared negatives, augmentation, sampled-softmax CE (the wikikg2 notebook's
    setup) on fp16 tables: scores, loss and the round-once SGD update against the oracle in half-query mode."""
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import SampledSoftmaxCrossEntropyLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import TransE
    from besskge.sharding import Sharding
    from oracle import kge

    torch.manual_seed(0)
    n_entity, n_rel, d, S, K = 3000, 11, 64, 128, 96
    sharding = Sharding.create(n_entity, 1, seed=3)
    ent = (torch.randn(1, sharding.max_entity_per_shard, d) * 0.5).half().float()
    rel = (torch.randn(n_rel, d) * 0.5).half().float()
    fn = TransE(True, 1, sharding, n_rel, d, ent, rel)
    ns = RandomShardedNegativeSampler(K, sharding, 5, "t", local_sampling=False, flat_negative_format=True)
    model = EmbeddingMovingBessKGE(ns, fn, SampledSoftmaxCrossEntropyLoss(n_entity), return_scores=True,
                                   augment_negative=True)
    rng = np.random.default_rng(2)
    batch = dict(head=rng.integers(n_entity, size=(1, 1, S)), relation=rng.integers(n_rel, size=(1, 1, S)),
                 tail=rng.integers(n_entity, size=(1, 1, S)), negative=rng.integers(n_entity, size=(1, 1, 1, K)))
    batch = {k: torch.from_numpy(v.astype(np.int32)) for k, v in batch.items()}
    spec = kge.StepSpec("TransE", 1, True, "t", True, augment=True)
    t0, r0 = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
    with kge.half_queries():
        want = kge.bess_step(spec, "EmbeddingMoving", t0, r0, batch, dict(kind="ssce", n_entity=n_entity))
        want["loss"][0].backward()
    lr = 0.05
    runner = runtime.training_model(model, optimizer=runtime.SGD(lr=lr), device=dev, dtype=torch.float16)
    res = runner(**batch)
    neg = res["negative_score"].float().cpu()
    wn = want["negative_score"][0].detach()
    live = wn > -40000
    # the fp16 return dtype of the module rounds the scores: compare in fp16 resolution
    torch.testing.assert_close(neg[live], wn[live].half().float(), rtol=2e-3, atol=2e-3)
    assert bool((neg[~live] < -40000).all())
    torch.testing.assert_close(res["loss"].float().cpu().reshape(()), want["loss"][0].detach(), rtol=2e-5, atol=1e-4)
    got = model.score_fn.entity_embedding.detach().float().cpu()
    want_ent = (ent - lr * t0.grad).half().float()
    ulp = torch.exp2(torch.floor(torch.log2(want_ent.abs().clamp(min=2.0 ** -14))) - 10)
    err = (got - want_ent).abs()
    # the two fp32 gradients differ in their last bits: at most a rounding boundary is crossed now and then
    slack = 1e-4 * want_ent.abs() + 2e-5  # the fp32 tables' tolerance (where row and update cancel, the gradients' fp32 difference shows)
    assert bool((err <= ulp * 1.001 + slack).all()), float((err / ulp).max())
    assert float((err > 0).float().mean()) < 0.02
