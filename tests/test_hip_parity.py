"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle and vs the
reference's own outputs (tests/golden/*.npz).

Tolerances.  Scores: the reference's own test tolerance for negative scores,
rtol=1e-4 / atol=1e-5 (reference tests/test_bess.py:245-246, 255-256, 273-274),
is applied to positives *and* negatives against the fp32 oracle.  Index / byte
movers (gather, scatter of exactly representable values) are bit-exact.
fp16 tables: the kernels accumulate in fp32 from fp16-rounded inputs, the oracle
is evaluated on the same fp16-rounded inputs in fp32 -> same tolerance.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import kge  # noqa: E402

from conftest import load_golden  # noqa: E402
from test_oracle import (  # noqa: E402
    AFFINE_LP, AFFINE_SCORERS, BOXE_LP, BOXE_SCORERS, LOSSES, SCORERS, T, bess_cases, load_bess_case, scoring_fixture,
    step_batch)

NATIVE_SCORERS = [s for s in SCORERS if s not in AFFINE_SCORERS + BOXE_SCORERS + AFFINE_LP + BOXE_LP]

RTOL, ATOL = 1e-4, 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a HIP device"
    return torch.device("cuda", 0)


def close(got, want, rtol=RTOL, atol=ATOL, scale=None):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    if scale is not None:  # absolute floor relative to the magnitude of the data
        atol = max(atol, scale * float(want.abs().max()))
    torch.testing.assert_close(got, want, rtol=rtol, atol=atol)


def make_scorer(name, p, sharing, n_rel, d, ent, rel, dev, dtype=torch.float32, sharding=None, net=None):
    from besskge.scoring import BoxE, ComplEx, ConvE, DistMult, InterHT, PairRE, RotatE, TranS, TransE, TripleRE
    from besskge.sharding import Sharding

    if sharding is None:  # only n_shard matters for the training / scoring step
        sharding = Sharding.create(ent.shape[0] * ent.shape[1], ent.shape[0], seed=0)
    if name in kge.BOXE_VARIANTS:
        tanh, per_dim = kge.BOXE_VARIANTS[name]
        fn = BoxE(sharing, p, sharding, n_rel, d, ent, rel, apply_tanh=tanh, dist_func_per_dim=per_dim)
    elif name == "ConvE":
        fn = ConvE(sharing, sharding, n_rel, d, d // 4, 4, ent, rel, inverse_relations=False, input_dropout=0.0,
                   feature_map_dropout=0.0, hidden_dropout=0.0)
        missing, unexpected = fn.load_state_dict(net, strict=False)
        assert not unexpected and set(missing) == {"entity_embedding", "relation_embedding"}
    elif name in kge.AFFINE_VARIANTS:
        cfg = kge.AFFINE_VARIANTS[name]
        cls = dict(PairRE=PairRE, TripleRE=TripleRE, InterHT=InterHT, TranS=TranS)[cfg["base"]]
        kw = dict(normalize_entities=cfg["normalize"])
        if "u" in cfg:
            kw["u"] = cfg["u"]
        if "offset" in cfg:
            kw["offset"] = cfg["offset"]
        fn = cls(sharing, p, sharding, n_rel, d, ent, rel, **kw)
    elif name == "TransE":
        fn = TransE(sharing, p, sharding, n_rel, d, ent, rel)
    elif name == "RotatE":
        fn = RotatE(sharing, p, sharding, n_rel, d, ent, rel)
    elif name == "DistMult":
        fn = DistMult(sharing, sharding, n_rel, d, ent, rel)
    else:
        fn = ComplEx(sharing, sharding, n_rel, d, ent, rel)
    fn = fn.to(dev)
    if dtype == torch.float16:
        fn = fn.half()
    return fn


def widths(name, d):
    return kge.entity_width(name, d), kge.relation_width(name, d)


# ---------------------------------------------------------------- movers ----
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("W", [128, 400, 512, 100, 6])
def test_gather_rows_bit_exact(dev, dtype, W):
    from besskge import _native as nat

    g = torch.Generator().manual_seed(W)
    table = torch.randn(1000, W, generator=g).to(dtype)
    idx = torch.randint(1000, (777,), generator=g, dtype=torch.int32)
    out = nat.gather_rows(table.to(dev), idx.to(dev))
    assert torch.equal(out.cpu(), table[idx.long()])
    empty = nat.gather_rows(table.to(dev), idx[:0].to(dev))
    assert empty.shape == (0, W)


def test_scatter_add_and_sgd(dev):
    from besskge import _native as nat

    g = torch.Generator().manual_seed(3)
    W = 64
    # small integers: sums are exact in fp32 whatever the order of the atomics
    src = torch.randint(-8, 9, (500, W), generator=g).float()
    idx = torch.randint(40, (500,), generator=g, dtype=torch.int32)
    dst = torch.zeros(40, W)
    want = dst.clone().index_add_(0, idx.long(), src)
    got = dst.to(dev)
    nat.scatter_add_rows(got, idx.to(dev), src.to(dev))
    assert torch.equal(got.cpu(), want)
    table = torch.randint(-8, 9, (40, W), generator=g).float()
    want_t = table - 0.5 * want
    t32 = table.to(dev)
    nat.sparse_sgd(t32, idx.to(dev), src.to(dev), 0.5)
    assert torch.equal(t32.cpu(), want_t)
    t16 = table.half().to(dev)
    nat.sparse_sgd(t16, idx.to(dev), src.to(dev), 0.5)
    close(t16, want_t, rtol=2e-3, atol=0.5)
    rel = table.clone().to(dev)
    nat.dense_sgd(rel, want.to(dev), 0.25)
    assert torch.equal(rel.cpu(), table - 0.25 * want)


def test_cpu_tensors_are_refused():
    from besskge import _native as nat

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nat.gather_rows(torch.zeros(4, 8), torch.zeros(2, dtype=torch.int32))


# --------------------------------------------------------------- scoring ----
@pytest.mark.parametrize("name,p", SCORERS)
@pytest.mark.parametrize("sharing", [True, False])
@pytest.mark.parametrize("B", [1, 10])
def test_scoring_golden(dev, name, p, sharing, B):
    """score_triple / score_heads / score_tails + gradients vs the reference's outputs."""
    g = scoring_fixture(name, p)
    S, N, d, n_rel, n_ent = (int(x) for x in g["args"])
    k = f"{name}_p{p}_"
    c = k + f"s{int(sharing)}_B{B}_"
    W, Wr = widths(name, d)
    fn = make_scorer(name, p, sharing, n_rel, d, torch.zeros(1, n_ent, W), T(g[k + "rel"]), dev)
    fn.relation_embedding.requires_grad_(True)
    h = T(g[k + "h"]).to(dev).requires_grad_(True)
    t = T(g[k + "t"]).to(dev).requires_grad_(True)
    rid = T(g[k + "rid"]).to(dev)
    neg = T(g[k + ("neg1" if B == 1 else "negS")]).to(dev).requires_grad_(True)

    pos = fn.score_triple(h, rid, t)
    close(pos, T(g[c + "pos"]))
    (pos * T(g[k + "g_pos"]).to(dev)).sum().backward()
    close(h.grad, T(g[c + "pos_dh"]))
    close(t.grad, T(g[c + "pos_dt"]))
    close(fn.relation_embedding.grad, T(g[c + "pos_drel"]))

    for method, ent, key, dkey in ((fn.score_heads, t, "heads", "heads_dt"), (fn.score_tails, h, "tails", "tails_dh")):
        for x in (h, t, neg, fn.relation_embedding):
            x.grad = None
        sc = method(neg, rid, t) if key == "heads" else method(h, rid, neg)
        close(sc, T(g[c + key]))
        (sc * T(g[c + key + "_g"]).to(dev)).sum().backward()
        close(neg.grad, T(g[c + key + "_dneg"]))
        close(ent.grad, T(g[c + dkey]))
        close(fn.relation_embedding.grad, T(g[c + key + "_drel"]))


@pytest.mark.parametrize("name,p", SCORERS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("d,S,N", [(64, 70, 33), (100, 9, 130), (256, 130, 65), (6, 5, 3)])
def test_scoring_vs_oracle(dev, name, p, dtype, d, S, N):
    """Ragged sizes, both table dtypes, both regimes, against the CPU oracle."""
    if d == 256 and name not in ("TransE", "RotatE", "DistMult", "ComplEx"):
        S, N = 34, 17  # the oracle's [S, S*N, d] broadcasts are slow (4 - 10 s a case); the wide-row kernel path is the same
    gen = torch.Generator().manual_seed(d * 1000 + S)
    W, Wr = widths(name, d)
    n_rel = 7
    rel = torch.randn(n_rel, Wr, generator=gen).to(dtype)
    h = torch.randn(S, W, generator=gen).to(dtype)
    t = torch.randn(S, W, generator=gen).to(dtype)
    rid = torch.randint(n_rel, (S,), generator=gen)
    negS = torch.randn(S, N, W, generator=gen).to(dtype)
    neg1 = torch.randn(1, N, W, generator=gen).to(dtype)
    scale = 2e-6  # |error| floor relative to the largest score (fp32 sum of W terms)
    import contextlib
    for sharing, neg in ((True, neg1), (True, negS), (False, negS)):
        fn = make_scorer(name, p, sharing, n_rel, d, torch.zeros(1, 4, W), rel.float(), dev, dtype)
        o = dict(scorer=name, p=p)
        close(fn.score_triple(h.to(dev), rid.to(dev), t.to(dev)),
              kge.score_triple(name, p, h.float(), rel.float(), rid, t.float()), scale=scale)
        # fp16 tables, TransE / RotatE with p = 1, shared negatives, W % 32 == 0: the packed-fp16 kernel scores the
        # query rounded to fp16 (the reference's fp16 mode) - so does the oracle inside half_queries()
        half = dtype == torch.float16 and name in ("TransE", "RotatE") and p == 1 and sharing and W % 32 == 0
        with (kge.half_queries() if half else contextlib.nullcontext()):
            close(fn.score_heads(neg.to(dev), rid.to(dev), t.to(dev)),
                  kge.score_candidates(sharing=sharing, side="h", ent=t.float(), rel_table=rel.float(), rid=rid,
                                       cand=neg.float(), **o), scale=scale)
            close(fn.score_tails(h.to(dev), rid.to(dev), neg.to(dev)),
                  kge.score_candidates(sharing=sharing, side="t", ent=h.float(), rel_table=rel.float(), rid=rid,
                                       cand=neg.float(), **o), scale=scale)


@pytest.mark.parametrize("name,p", SCORERS)
@pytest.mark.parametrize("sharing", [True, False])
def test_scoring_gradients_vs_oracle(dev, name, p, sharing):
    gen = torch.Generator().manual_seed(11)
    d, S, N, n_rel = 48, 37, 21, 5
    W, Wr = widths(name, d)
    rel = torch.randn(n_rel, Wr, generator=gen)
    h = torch.randn(S, W, generator=gen)
    t = torch.randn(S, W, generator=gen)
    rid = torch.randint(n_rel, (S,), generator=gen)
    neg = torch.randn(S, N, W, generator=gen)
    gp = torch.randn(S, generator=gen)
    fn = make_scorer(name, p, sharing, n_rel, d, torch.zeros(1, 4, W), rel, dev)
    fn.relation_embedding.requires_grad_(True)
    for side in ("h", "t"):
        hd, td, nd = (x.clone().to(dev).requires_grad_(True) for x in (h, t, neg))
        ho, to_, no, ro = (x.clone().requires_grad_(True) for x in (h, t, neg, rel))
        fn.relation_embedding.grad = None
        if side == "h":
            sc = fn.score_heads(nd, rid.to(dev), td)
            so = kge.score_candidates(name, p, sharing, "h", to_, ro, rid, no)
        else:
            sc = fn.score_tails(hd, rid.to(dev), nd)
            so = kge.score_candidates(name, p, sharing, "t", ho, ro, rid, no)
        gn = torch.randn(so.shape, generator=gen)
        pos = fn.score_triple(hd, rid.to(dev), td)
        po = kge.score_triple(name, p, ho, ro, rid, to_)
        ((sc * gn.to(dev)).sum() + (pos * gp.to(dev)).sum()).backward()
        ((so * gn).sum() + (po * gp).sum()).backward()
        close(hd.grad, ho.grad, scale=2e-6)
        close(td.grad, to_.grad, scale=2e-6)
        close(nd.grad, no.grad, scale=2e-6)
        close(fn.relation_embedding.grad, ro.grad, rtol=1e-4, scale=2e-6)


@pytest.mark.parametrize("name,p", [("BoxE", 1), ("BoxE", 2), ("BoxEnt", 3)])
@pytest.mark.parametrize("dtype,d", [(torch.float32, 320), (torch.float32, 512), (torch.float16, 400)])
def test_boxe_wide_embeddings_vs_oracle(dev, name, p, dtype, d):
    """BoxE beyond 256 dimensions (the reference has no limit - scoring.py:690-797; the kernels keep a box row in a
    16-lane group's registers: 512 dimensions since round 4, loud refusal above): scores and
    gradients of both regimes against the oracle."""
    gen = torch.Generator().manual_seed(d + p)
    S, N, n_rel = 6, 9, 3
    W, Wr = widths(name, d)
    rel = (torch.randn(n_rel, Wr, generator=gen) * 0.5).to(dtype)
    h = (torch.randn(S, W, generator=gen) * 0.5).to(dtype)
    t = (torch.randn(S, W, generator=gen) * 0.5).to(dtype)
    rid = torch.randint(n_rel, (S,), generator=gen)
    neg = (torch.randn(S, N, W, generator=gen) * 0.5).to(dtype)
    for sharing, cand in ((True, neg[:1]), (False, neg)):
        fn = make_scorer(name, p, sharing, n_rel, d, torch.zeros(1, 4, W), rel.float(), dev, dtype)
        hd, nd = (x.clone().to(dev).requires_grad_(True) for x in (h, cand))
        ho, no, ro = (x.float().clone().requires_grad_(True) for x in (h, cand, rel))
        fn.relation_embedding.grad = None
        sc = fn.score_tails(hd, rid.to(dev), nd)
        so = kge.score_candidates(name, p, sharing, "t", ho, ro, rid, no)
        close(sc, so, scale=4e-6 if dtype == torch.float32 else 2e-3)
        gn = torch.randn(so.shape, generator=gen)
        (sc.float() * gn.to(dev)).sum().backward()
        (so * gn).sum().backward()
        tol = dict(scale=4e-6) if dtype == torch.float32 else dict(rtol=2e-2, scale=4e-3)
        close(hd.grad, ho.grad, **tol)
        close(nd.grad, no.grad, **tol)
        close(fn.relation_embedding.grad, ro.grad, **(dict(rtol=1e-4, scale=4e-6) if dtype == torch.float32 else tol))
    if dtype == torch.float32:
        wide = make_scorer(name, p, True, n_rel, 520, torch.zeros(1, 4, kge.entity_width(name, 520)),
                           torch.randn(n_rel, kge.relation_width(name, 520)), dev, dtype)
        with pytest.raises(RuntimeError, match="too wide"):
            wide.score_tails(torch.randn(2, kge.entity_width(name, 520), device=dev), rid[:2].to(dev),
                             torch.randn(1, 3, kge.entity_width(name, 520), device=dev))


# ---------------------------------------------------------------- losses ----
@pytest.mark.parametrize("name", list(LOSSES))
@pytest.mark.parametrize("wname", ["w", "one"])
def test_loss_golden(dev, name, wname):
    from besskge.loss import LogSigmoidLoss, MarginRankingLoss, SampledSoftmaxCrossEntropyLoss

    g = load_golden("loss")
    kw = dict(LOSSES[name])
    kind = kw.pop("kind")
    if kind == "logsigmoid":
        fn = LogSigmoidLoss(kw["margin"], kw["adversarial"], kw.get("adversarial_scale", 1.0), kw.get("loss_scale", 1.0))
    elif kind == "margin":
        fn = MarginRankingLoss(kw["margin"], kw["adversarial"], kw.get("adversarial_scale", 1.0), kw.get("loss_scale", 1.0))
    else:
        fn = SampledSoftmaxCrossEntropyLoss(kw["n_entity"], kw.get("loss_scale", 1.0))
    pos = T(g["pos"]).to(dev).requires_grad_(True)
    neg = T(g["neg"]).to(dev).requires_grad_(True)
    w = (T(g["w"]) if wname == "w" else torch.tensor([1.0])).to(dev)
    loss = fn(pos, neg, w)
    loss.backward()
    close(loss, T(g[f"{name}_{wname}_loss"]), rtol=1e-5, atol=1e-5)
    close(pos.grad, T(g[f"{name}_{wname}_dpos"]), rtol=1e-4, atol=1e-6)
    close(neg.grad, T(g[f"{name}_{wname}_dneg"]), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("kind", ["logsigmoid", "margin", "ssce"])
@pytest.mark.parametrize("S,N", [(1, 1), (65, 64), (130, 1000), (7, 4097),
                                 (9, 2048), (5, 4352), (3, 8192)])  # rows held in 12 / 24 register chunks, streamed
def test_loss_vs_oracle(dev, kind, S, N):
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(S * 7 + N)
    pos = torch.randn(S, generator=gen) * 4
    neg = torch.randn(S, N, generator=gen) * 4
    neg[0, 0] = -50000.0
    w = torch.rand(S, generator=gen) + 0.5
    kw = dict(kind=kind, margin=2.5, adversarial=(kind != "ssce"), adversarial_scale=0.3, loss_scale=1.5, n_entity=5000)
    ld = nat.LossDesc()
    ld.kind = dict(logsigmoid=0, margin=1, ssce=2)[kind]
    ld.adversarial, ld.margin, ld.adversarial_scale, ld.loss_scale = int(kind != "ssce"), 2.5, 0.3, 1.5
    ld.ssce_shift = float(np.log(5000 - 1) - np.log(N))
    loss, dp, dn = nat.loss_fwd_bwd(ld, pos.to(dev), neg.to(dev), w.to(dev), True)
    po, no = pos.clone().requires_grad_(True), neg.clone().requires_grad_(True)
    lo = kge.loss_value(pos=po, neg=no, w=w, **kw)
    lo.backward()
    close(loss, lo, rtol=2e-5, atol=1e-4)
    close(dp, po.grad, rtol=1e-4, atol=1e-6)
    close(dn, no.grad, rtol=1e-4, atol=1e-6)
    # bitwise reproducible (fixed-order reduction)
    loss2, _, _ = nat.loss_fwd_bwd(ld, pos.to(dev), neg.to(dev), w.to(dev), False)
    assert torch.equal(loss, loss2)


# ------------------------------------------------------- whole BESS step ----
def build_model(c, dev, lr_loss=True):
    from besskge.bess import EmbeddingMovingBessKGE, ScoreMovingBessKGE
    from besskge.loss import LogSigmoidLoss, MarginRankingLoss, SampledSoftmaxCrossEntropyLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler, TripleBasedShardedNegativeSampler

    meta, spec = c["meta"], c["spec"]
    fn = make_scorer(spec.scorer, spec.p, spec.sharing, meta["n_rel"], meta["d"], c["table"], c["rel"],
                     torch.device("cpu"), net=c.get("net"))
    loss = None
    if c["loss_name"] == "logsigmoid":
        loss = LogSigmoidLoss(margin=3.0, negative_adversarial_sampling=True, negative_adversarial_scale=0.5)
    elif c["loss_name"] == "margin":
        loss = MarginRankingLoss(margin=1.0, negative_adversarial_sampling=False)
    elif c["loss_name"] == "ssce":
        loss = SampledSoftmaxCrossEntropyLoss(n_entity=meta["n_entity"])

    # BessKGE only reads three flags (and the class) of its negative sampler
    kind = TripleBasedShardedNegativeSampler if spec.triple_based else RandomShardedNegativeSampler
    ns = object.__new__(kind)
    ns.flat_negative_format = spec.flat
    ns.local_sampling = spec.local_sampling
    ns.corruption_scheme = spec.scheme
    cls = EmbeddingMovingBessKGE if c["model_cls"] == "EmbeddingMoving" else ScoreMovingBessKGE
    return cls(negative_sampler=ns, score_fn=fn, loss_fn=loss, return_scores=True, augment_negative=spec.augment)


@pytest.mark.parametrize("case", bess_cases())
def test_bess_forward_golden(dev, case):
    """BessKGE.forward of all replicas (lock-step on one GPU) reproduces the
    reference's positive_score / negative_score / loss for every golden case."""
    from besskge import runtime

    c = load_bess_case(case)
    meta = c["meta"]
    n, bps = meta["n_shard"], meta["bps"]
    model = build_model(c, dev)
    runner = runtime.inference_model(model, runtime.Options(device_iterations=bps), device=dev)
    if c["net"] is not None:
        model.train()  # the ConvE fixtures were produced in train mode (batch statistics, no dropout)
    keys = ("head", "relation", "tail", "negative", "negative_mask")
    res = runner(**{k: c["batch"][k].flatten(end_dim=1) for k in keys if k in c["batch"]})
    S = c["outs"]["positive_score"].shape[-1]
    close(res["positive_score"].reshape(bps, n, S), c["outs"]["positive_score"])
    close(res["negative_score"].reshape(bps, n, S, -1), c["outs"]["negative_score"], atol=2e-3 if c["loss_name"] == "ssce" else ATOL)
    if c["loss"] is not None:
        close(res["loss"].reshape(bps, n), c["outs"]["loss"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", [c for c in bess_cases() if c.startswith("tr_")])
def test_bess_train_step_golden(dev, case):
    """One sparse-SGD step moves the tables by -lr * (the reference's autograd gradient)."""
    from besskge import runtime

    c = load_bess_case(case)
    meta = c["meta"]
    n = meta["n_shard"]
    lr = 0.125
    model = build_model(c, dev)
    runner = runtime.training_model(model, runtime.Options(device_iterations=1), runtime.SGD(lr=lr), device=dev)
    keys = ("head", "relation", "tail", "negative", "negative_mask")
    res = runner(**{k: c["batch"][k][0] for k in keys if k in c["batch"]})
    close(res["loss"].reshape(n), c["outs"]["loss"][0], rtol=1e-4, atol=1e-4)
    want_ent = c["table"] - lr * c["grads"]["entity"]
    want_rel = c["rel"] - lr * c["grads"]["relation"].sum(0)
    close(model.score_fn.entity_embedding, want_ent, rtol=1e-4, atol=2e-5)
    close(model.score_fn.relation_embedding, want_rel, rtol=1e-4, atol=2e-5)
    for name, prm in model.score_fn.named_parameters():  # ConvE: the query network moves too
        if name in c["grads_net"]:
            close(prm, c["net"][name] - lr * c["grads_net"][name], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("sharing", [True, False])
@pytest.mark.parametrize("B", [1, 10])
def test_conve_scoring_golden(dev, mode, sharing, B):
    """ConvE: torch query network + dot-product kernels over [embedding | bias] rows, vs the
    reference's outputs and gradients (embeddings, relation table, network parameters)."""
    g = load_golden("scoring_conve")
    S, N, d, n_rel, n_ent = (int(x) for x in g["args"])
    net = {k[len("net_"):]: T(g[k]) for k in g.files if k.startswith("net_")}
    c = f"{mode}_s{int(sharing)}_B{B}_"
    rid = T(g["rid"]).to(dev)
    for what in ("pos", "tails"):
        fn = make_scorer("ConvE", 0, sharing, n_rel, d, torch.zeros(1, n_ent, d + 1), T(g["rel"]), dev, net=net)
        fn.train(mode == "train")
        fn.relation_embedding.requires_grad_(True)
        h = T(g["h"]).to(dev).requires_grad_(True)
        if what == "pos":
            t = T(g["t"]).to(dev).requires_grad_(True)
            sc = fn.score_triple(h, rid, t)
            (sc * T(g["g_pos"]).to(dev)).sum().backward()
            close(t.grad, T(g[c + "pos_dt"]))
        else:
            neg = T(g["neg1" if B == 1 else "negS"]).to(dev).requires_grad_(True)
            sc = fn.score_tails(h, rid, neg)
            (sc * T(g[c + "tails_g"]).to(dev)).sum().backward()
            close(neg.grad, T(g[c + "tails_dneg"]))
        close(sc, T(g[c + what]))
        close(h.grad, T(g[c + what + "_dh"]), rtol=1e-3, atol=1e-5)
        close(fn.relation_embedding.grad, T(g[c + what + "_drel"]), rtol=1e-3, atol=1e-5)
        for name, prm in fn.named_parameters():
            if name.startswith(("conv_layers", "fc_layers")):
                close(prm.grad, T(g[c + what + "_dnet_" + name]), rtol=1e-3, atol=1e-4)
    with pytest.raises(NotImplementedError):
        fn.score_heads(h, rid, h)


# --------------------------- the reference's own integration test, on HIP ----
@pytest.mark.parametrize("model_name", ["ScoreMoving", "EmbeddingMoving"])
@pytest.mark.parametrize("scheme,dup", [("h", False), ("t", False), ("ht", True)])
@pytest.mark.parametrize("flat", [True, False])
def test_bess_inference_like_reference(dev, model_name, scheme, dup, flat):
    """Mirror of reference tests/test_bess.py:54-275: 4 replicas, TransE d=128,
    500 entities, triple-specific negatives; sharded HIP scores == unsharded
    oracle scores re-ordered through triple_sort_idx / negative_sort_idx."""
    from besskge import runtime
    from besskge.batch_sampler import RigidShardedBatchSampler
    from besskge.bess import BAD_NEGATIVE_SCORE, EmbeddingMovingBessKGE, ScoreMovingBessKGE
    from besskge.dataset import KGDataset
    from besskge.negative_sampler import TripleBasedShardedNegativeSampler
    from besskge.scoring import TransE
    from besskge.sharding import PartitionedTripleSet, Sharding

    seed, n_entity, n_rel, n_shard, n_triple = 1234, 500, 10, 4, 1000
    bps, shard_bs, n_negative, d = 3, 48, 250, 128
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    entity_table = torch.randn(n_shard, sharding.max_entity_per_shard, d)
    relation_table = torch.randn(n_rel, d)
    th, tt_, tr = rng.integers(n_entity, size=n_triple), rng.integers(n_entity, size=n_triple), rng.integers(n_rel, size=n_triple)
    triples = {"test": np.stack([th, tr, tt_], axis=1)}
    outer = 1 if flat else n_triple
    neg_h = rng.integers(n_entity, size=(outer, n_negative)).astype(np.int32)
    neg_t = rng.integers(n_entity, size=(outer, n_negative)).astype(np.int32)
    ds = KGDataset(n_entity=n_entity, n_relation_type=n_rel, triples=triples,
                   original_triple_ids={"test": np.arange(n_triple)}, neg_heads={"test": neg_h}, neg_tails={"test": neg_t})
    pts = PartitionedTripleSet.create_from_dataset(ds, "test", sharding, partition_mode="ht_shardpair")
    score_fn = TransE(flat, 1, sharding, n_rel, d, entity_table, relation_table)
    ns = TripleBasedShardedNegativeSampler(pts.neg_heads, pts.neg_tails, sharding, corruption_scheme=scheme,
                                           seed=seed, return_sort_idx=True, mask_on_gather=False)
    bs = RigidShardedBatchSampler(pts, ns, shard_bs, bps, seed, duplicate_batch=dup, return_triple_idx=True)
    cls = EmbeddingMovingBessKGE if model_name == "EmbeddingMoving" else ScoreMovingBessKGE
    model = cls(negative_sampler=ns, score_fn=score_fn, return_scores=True)
    runner = runtime.inference_model(model, runtime.Options(device_iterations=bps), device=dev)

    flat_table = entity_table[sharding.entity_to_shard, sharding.entity_to_idx]
    he, te = flat_table[th], flat_table[tt_]
    true_pos = kge.score_triple("TransE", 1, he, relation_table, T(tr), te)
    true_nh = kge.score_candidates("TransE", 1, flat, "h", te, relation_table, T(tr), flat_table[neg_h.astype(np.int64)])
    true_nt = kge.score_candidates("TransE", 1, flat, "t", he, relation_table, T(tr), flat_table[neg_t.astype(np.int64)])
    sort_idx = T(pts.triple_sort_idx)

    for batch in bs.get_dataloader(shuffle=False):
        triple_idx = batch.pop("triple_idx")
        triple_mask = batch.pop("triple_mask")
        nsi = batch.pop("negative_sort_idx")
        res = runner(**{k: v.flatten(end_dim=1) for k, v in batch.items()})
        pos = res["positive_score"].cpu().reshape(bps, n_shard, n_shard, -1)
        negs = res["negative_score"].cpu()
        negs = negs[negs > 0.95 * BAD_NEGATIVE_SCORE].reshape(bps, n_shard, n_shard, -1, n_negative)
        nsi = nsi.reshape(bps, n_shard, n_shard, -1, n_negative).long()
        if dup:
            cut = pos.shape[-1] // 2
            triple_idx, pos, triple_mask = triple_idx[..., :cut], pos[..., :cut], triple_mask[..., :cut]
            n1, n2 = torch.split(negs, negs.shape[-2] // 2, dim=-2)
            s1, s2 = torch.split(nsi, cut, dim=-2)
        gidx = triple_idx[triple_mask]
        torch.testing.assert_close(true_pos[sort_idx][gidx], pos[triple_mask], rtol=RTOL, atol=ATOL)
        if dup:
            for true, got, s in ((true_nh, n1, s1), (true_nt, n2, s2)):
                torch.testing.assert_close(torch.take_along_dim(true[sort_idx][gidx], s[triple_mask], dim=-1),
                                           got[triple_mask], rtol=RTOL, atol=ATOL)
        else:
            true = true_nh if scheme == "h" else true_nt
            torch.testing.assert_close(torch.take_along_dim(true[sort_idx][gidx], nsi[triple_mask], dim=-1),
                                       negs[triple_mask], rtol=RTOL, atol=ATOL)


# ------------------------------------------- K9 segmented reduction (no atomics) ----
@pytest.mark.parametrize("name,p", NATIVE_SCORERS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_grad_segments_match_scatter_of_row_gradients(dev, name, p, dtype):
    """grad_seg == index_add of the per-reference row gradients of the plain backward
    kernel; the unique rows / offsets are exact; two runs are bitwise identical."""
    from besskge import _native as nat
    from besskge._native import RowSource

    gen = torch.Generator().manual_seed(5)
    M, d, S, N = 300, 40, 33, 17
    W, Wr = widths(name, d)
    table = torch.randn(M, W, generator=gen).to(dtype).to(dev)
    q = torch.randn(S, W, generator=gen).to(dev)
    idx = torch.randint(M, (S * N,), generator=gen, dtype=torch.int32).to(dev)
    go = torch.randn(S, N, generator=gen).to(dev)
    desc = nat.make_desc(dict(TransE=0, RotatE=1, DistMult=2, ComplEx=3)[name], max(p, 1), table, Wr)
    dq_ref, dn = nat.neg_score_pertriple_bwd(desc, q, RowSource(table, idx), N, go)
    dq, none = nat.neg_score_pertriple_bwd(desc, q, RowSource(table, idx), N, go, want_d_neg=False)
    assert none is None
    close(dq, dq_ref, rtol=1e-5, atol=1e-5)  # partial sums over negative blocks are combined atomically
    # d_neg alone (d_query came out of the fused forward): the same rows - DistMult / ComplEx without reading a candidate
    none, dn_only = nat.neg_score_pertriple_bwd(desc, q, RowSource(table, idx), N, go, want_d_query=False)
    assert none is None and torch.equal(dn_only, dn)
    # lists that name a row at most once: the gradient rows stored straight into a row-space matrix
    perm = torch.randperm(S * N + 50, generator=gen)[: S * N].to(torch.int32).to(dev)
    big = torch.randn(S * N + 50, W, generator=gen).to(dtype).to(dev)
    _, dn_p = nat.neg_score_pertriple_bwd(desc, q, RowSource(big, perm), N, go)
    by_row = torch.zeros((big.shape[0], W), dtype=torch.float32, device=dev)
    dq_p, none = nat.neg_score_pertriple_bwd(desc, q, RowSource(big, perm), N, go, d_neg_rows=by_row)
    assert none is None and torch.equal(by_row[perm.long()], dn_p)
    untouched = torch.ones(big.shape[0], dtype=torch.bool, device=dev)
    untouched[perm.long()] = False
    assert float(by_row[untouched].abs().max()) == 0.0
    seg = nat.SegmentIndex(idx, M)
    n_seg = int(seg.n_seg.item())
    uniq, counts = torch.unique(idx.cpu().long(), return_counts=True)
    assert n_seg == uniq.numel()
    assert torch.equal(seg.seg_rows[:n_seg].cpu().long(), uniq)
    assert torch.equal(seg.seg_offsets[: n_seg + 1].cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]))
    # stable: references of one row stay in reference order
    refs = seg.refs.cpu().long()
    assert torch.equal(idx.cpu().long()[refs], torch.sort(idx.cpu().long(), stable=True).values)
    assert torch.equal(refs, torch.sort(idx.cpu().long(), stable=True).indices)
    g1 = nat.neg_pertriple_grad_segments(desc, q, table, N, go, seg)
    g2 = nat.neg_pertriple_grad_segments(desc, q, table, N, go, nat.SegmentIndex(idx, M))
    assert torch.equal(g1[:n_seg], g2[:n_seg])
    want = torch.zeros(M, W, dtype=torch.float64).index_add_(0, idx.cpu().long(), dn.cpu().double())
    close(g1[:n_seg], want[uniq].float(), rtol=1e-4, atol=1e-5, scale=2e-6)
    t2 = table.clone()
    nat.apply_segments_sgd(t2, seg, g1, 0.5)
    want_t = table.float().cpu() - 0.5 * want.float()
    close(t2, want_t, rtol=2e-3 if dtype == torch.float16 else 1e-5, atol=2e-3 if dtype == torch.float16 else 1e-5)


# ------------------------------------------------------------ hipGraph replay ----
@pytest.mark.parametrize("case", ["tr_EM_TransE1_t_flat_n1", "tr_EM_ComplEx0_h_pt_n1", "tr_EM_aug_t_flat_n4",
                                  "tr_EM_RotatE2_ht_flat_n2", "tr_SM_ht_flat_n2", "inf_SM_t_0_n4"])
def test_graph_replay_matches_reference(dev, case):
    """Options.use_graphs: the captured step (forward, and forward+backward+SGD)
    replayed from static buffers gives the reference's outputs / gradient step."""
    from besskge import runtime

    c = load_bess_case(case)
    meta = c["meta"]
    n, bps = meta["n_shard"], meta["bps"]
    keys = ("head", "relation", "tail", "negative", "negative_mask")
    batch = {k: c["batch"][k].flatten(end_dim=1) for k in keys if k in c["batch"]}
    model = build_model(c, dev)
    runner = runtime.inference_model(model, runtime.Options(device_iterations=bps, use_graphs=True), device=dev)
    for _ in range(2):  # second call replays the cached graph
        res = runner(**batch)
    S = c["outs"]["positive_score"].shape[-1]
    close(res["positive_score"].reshape(bps, n, S), c["outs"]["positive_score"])
    close(res["negative_score"].reshape(bps, n, S, -1), c["outs"]["negative_score"],
          atol=2e-3 if c["loss_name"] == "ssce" else ATOL)
    if c["loss"] is not None:
        close(res["loss"].reshape(bps, n), c["outs"]["loss"], rtol=1e-4, atol=1e-4)
    if case.startswith("tr_"):
        lr = 0.125
        model = build_model(c, dev)
        runner = runtime.training_model(model, runtime.Options(device_iterations=1, use_graphs=True),
                                        runtime.SGD(lr=lr), device=dev)
        res = runner(**{k: v[:n] for k, v in batch.items()})
        close(res["loss"].reshape(n), c["outs"]["loss"][0], rtol=1e-4, atol=1e-4)
        close(model.score_fn.entity_embedding, c["table"] - lr * c["grads"]["entity"], rtol=1e-4, atol=2e-5)
        close(model.score_fn.relation_embedding, c["rel"] - lr * c["grads"]["relation"].sum(0), rtol=1e-4, atol=2e-5)


# ------------------------------------ fused training forward (scores + d loss / d query) ----
@pytest.mark.parametrize("name,p", NATIVE_SCORERS)
@pytest.mark.parametrize("loss", list(LOSSES))
@pytest.mark.parametrize("dtype,wname", [(torch.float32, "w"), (torch.float16, "one")])
def test_fused_forward_dq_matches_two_pass(dev, name, p, loss, dtype, wname):
    """One pass over the negative rows (online-softmax accumulation of the loss weights) gives the
    scores of the plain forward and the d_query of loss kernel + backward kernel."""
    from besskge import _native as nat
    from besskge._native import RowSource
    from besskge.loss import LogSigmoidLoss, MarginRankingLoss, SampledSoftmaxCrossEntropyLoss

    kw = dict(LOSSES[loss])
    kind = kw.pop("kind")
    if kind == "logsigmoid":
        fn = LogSigmoidLoss(kw["margin"], kw["adversarial"], kw.get("adversarial_scale", 1.0), kw.get("loss_scale", 1.0))
    elif kind == "margin":
        fn = MarginRankingLoss(kw["margin"], kw["adversarial"], kw.get("adversarial_scale", 1.0), kw.get("loss_scale", 1.0))
    else:
        fn = SampledSoftmaxCrossEntropyLoss(kw["n_entity"], kw.get("loss_scale", 1.0))
    gen = torch.Generator().manual_seed(7)
    for M, d, S, N, amp in ((300, 40, 33, 150, 0.5), (500, 256, 70, 256, 0.5), (50, 6, 5, 3, 0.5), (40, 8, 4, 1, 0.5),
                            (300, 16, 9, 200, 6.0)):  # amp 6: scores of +-1e3, the softmax is one-hot
        W, Wr = widths(name, d)
        table = (amp * torch.randn(M, W, generator=gen)).to(dtype).to(dev)
        q = (amp * torch.randn(S, W, generator=gen)).to(dev)
        idx = torch.randint(M, (S * N,), generator=gen, dtype=torch.int32).to(dev)
        pos = torch.randn(S, generator=gen).to(dev)
        w = (torch.rand(S, generator=gen) + 0.5).to(dev) if wname == "w" else torch.ones(1, device=dev)
        desc = nat.make_desc(dict(TransE=0, RotatE=1, DistMult=2, ComplEx=3)[name], max(p, 1), table, Wr)
        ld = fn.kernel_desc(N)
        neg = RowSource(table, idx)
        out, dq = nat.neg_score_pertriple_fwd_dq(desc, ld, q, neg, N, pos, w)
        ref = nat.neg_score_pertriple_fwd(desc, q, neg, N)
        assert torch.equal(out, ref)
        _, _, dn = nat.loss_fwd_bwd(ld, pos, ref, w, True)
        dq_ref, _ = nat.neg_score_pertriple_bwd(desc, q, neg, N, dn, want_d_neg=False)
        close(dq, dq_ref, rtol=2e-4, atol=1e-6, scale=4e-6)


@pytest.mark.parametrize("n_part,normalize,p", [(1, True, 1), (1, False, 2), (2, True, 2), (2, False, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_affine_grad_segments_match_scatter_of_row_gradients(dev, n_part, normalize, p, dtype):
    """K9 of the PairRE / TripleRE / InterHT / TranS family: the segmented reduction (per-reference
    gradient recomputed from the query, normalisation backward applied once per row) equals
    index_add of the backward kernel's row gradients."""
    from besskge import _native as nat
    from besskge._native import RowSource

    gen = torch.Generator().manual_seed(9)
    for M, d, S, N in ((300, 40, 33, 17), (64, 6, 9, 5)):
        W = n_part * d
        table = torch.randn(M, W, generator=gen).to(dtype).to(dev)
        q = torch.randn(S, (n_part + 1) * d, generator=gen).to(dev)
        idx = torch.randint(M, (S * N,), generator=gen, dtype=torch.int32).to(dev)
        go = torch.randn(S, N, generator=gen).to(dev)
        desc = nat.make_desc(nat.AFFINE, p, table, d)
        desc.reserved[0], desc.reserved[1] = n_part, int(normalize)
        _, dn = nat.neg_score_pertriple_bwd(desc, q, RowSource(table, idx), N, go)
        seg = nat.SegmentIndex(idx, M)
        n_seg = int(seg.n_seg.item())
        uniq = torch.unique(idx.cpu().long())
        g1 = nat.neg_pertriple_grad_segments(desc, q, table, N, go, seg)
        want = torch.zeros(M, W, dtype=torch.float64).index_add_(0, idx.cpu().long(), dn.cpu().double())
        close(g1[:n_seg], want[uniq].float(), rtol=1e-4, atol=1e-5, scale=4e-6)
        t2 = table.clone()
        nat.neg_pertriple_grad_segments(desc, q, t2, N, go, seg, fused_sgd_lr=0.5)
        tol = 2e-3 if dtype == torch.float16 else 1e-5
        close(t2, table.float().cpu() - 0.5 * want.float(), rtol=tol, atol=tol, scale=4e-6)


@pytest.mark.parametrize("name,p", AFFINE_SCORERS + AFFINE_LP)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_affine_query_kernels_match_torch_formulas(dev, name, p, dtype):
    """k_aff_query_fwd / bwd (the [U | V | R] transform used by the fused step) against the torch
    expressions of the public API and their autograd, both corruption sides; and the positive score
    as the tail-corruption score of the one true tail."""
    from besskge import _native as nat
    from besskge._native import RowSource

    gen = torch.Generator().manual_seed(21)
    for d, S, M in ((40, 37, 90), (6, 5, 11)):
        W, Wr = widths(name, d)
        n_rel = 5
        rel = torch.randn(n_rel, Wr, generator=gen).to(dtype)
        table = torch.randn(M, W, generator=gen).to(dtype)
        fn = make_scorer(name, p, False, n_rel, d, torch.zeros(1, 4, W), rel.float(), dev, dtype)
        tab = table.to(dev)
        idx = torch.randint(M, (S,), generator=gen, dtype=torch.int32).to(dev)
        rid = torch.randint(n_rel, (S,), generator=gen, dtype=torch.int32).to(dev)
        for side in (nat.CORRUPT_HEAD, nat.CORRUPT_TAIL):
            q, _ = fn.query_fwd(side, RowSource(tab, idx), rid)
            rows = tab[idx.long()].float().requires_grad_(True)
            relp = fn.relation_embedding.detach().float().requires_grad_(True)
            want = torch.cat(fn._uvr(side, fn._parts(rows), fn._rel(rid, relp)), dim=-1)
            close(q, want, scale=2e-6)
            dq = torch.randn(want.shape, generator=gen).to(dev)
            d_rel = torch.zeros(n_rel, Wr, device=dev)
            dx = fn.query_bwd(side, RowSource(tab, idx), rid, None, dq, d_rel)
            g_rows, g_rel = torch.autograd.grad(want, [rows, relp], dq)
            close(dx, g_rows, scale=4e-6)
            close(d_rel, g_rel, scale=4e-6)
        tidx = torch.randint(M, (S,), generator=gen, dtype=torch.int32).to(dev)
        pos, ctx = fn.triple_fwd(RowSource(tab, idx), RowSource(tab, tidx), rid)
        h = tab[idx.long()].float().requires_grad_(True)
        t = tab[tidx.long()].float().requires_grad_(True)
        relp = fn.relation_embedding.detach().float().requires_grad_(True)
        want = fn._score_norm(fn._delta(fn._parts(h), fn._rel(rid, relp), fn._parts(t)))
        close(pos, want, scale=2e-6)
        gp = torch.randn(S, generator=gen).to(dev)
        d_rel = torch.zeros(n_rel, Wr, device=dev)
        dh, dt = fn.triple_bwd(RowSource(tab, idx), RowSource(tab, tidx), rid, ctx, gp, d_rel)
        gh, gt, gr = torch.autograd.grad(want, [h, t, relp], gp)
        close(dh, gh, scale=4e-6)
        close(dt, gt, scale=4e-6)
        close(d_rel, gr, scale=4e-6)


@pytest.mark.parametrize("sharing", [True, False])
def test_public_broadcast_helpers(dev, sharing):
    """DistanceBasedScoreFunction.broadcasted_distance / MatrixDecompositionScoreFunction
    .broadcasted_dot_product / reduce_embedding / BoxE.boxe_score keep the reference's semantics
    (scoring.py:163-255, 1250-1340)."""
    gen = torch.Generator().manual_seed(3)
    S, N, d, n_rel = 9, 13, 20, 4
    q = torch.randn(S, d, generator=gen)
    neg = torch.randn(1 if sharing else S, N, d, generator=gen)
    for p in (1, 2):
        fn = make_scorer("TransE", p, sharing, n_rel, d, torch.zeros(1, 4, d), torch.zeros(n_rel, d), dev)
        got = fn.broadcasted_distance(q.to(dev), neg.to(dev))
        want = torch.norm(q[:, None, :] - (neg.reshape(1, -1, d) if sharing else neg), p=p, dim=-1)
        close(got, want, scale=2e-6)
        close(fn.reduce_embedding(q.to(dev)), torch.norm(q, p=p, dim=-1))
    fn = make_scorer("DistMult", 0, sharing, n_rel, d, torch.zeros(1, 4, d), torch.zeros(n_rel, d), dev)
    got = fn.broadcasted_dot_product(q.to(dev), neg.to(dev))
    want = (q[:, None, :] * (neg.reshape(1, -1, d) if sharing else neg)).sum(-1)
    close(got, want, scale=2e-6)
    close(fn.reduce_embedding(q.to(dev)), q.sum(-1))
    for name, p in (("BoxE", 1), ("BoxEnt", 2)):
        box = make_scorer(name, p, sharing, n_rel, d, torch.zeros(1, 4, 2 * d), torch.zeros(n_rel, 4 * d + 2), dev)
        bumped = torch.randn(S, N, 2, d, generator=gen)
        rel = torch.randn(S, 1, 4 * d + 2, generator=gen)
        got = box.boxe_score(bumped.to(dev), rel[..., : 2 * d].reshape(S, 1, 2, d).to(dev),
                             rel[..., 2 * d: 4 * d].reshape(S, 1, 2, d).to(dev), rel[..., 4 * d:].to(dev))
        close(got, kge.boxe_score(name, p, bumped, rel), scale=2e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("S,N,W,use_idx,scale", [
    (2048, 2048, 256, True, 0.1),      # exactly 256 tiles
    (2051, 2309, 100, True, 1.0),      # ragged rows, columns and k (W not a multiple of 4)
    (1700, 33000, 64, False, 3e-4),    # dense candidates (all-entities scoring), tiny values
    (4099, 1100, 136, True, 50.0),     # W % 8 == 0 but not % 32, large values
    (130, 40000, 8, False, 1.0),       # a single K slice, fewer query rows than one 256-row tile
    (2048, 2048, 8, True, 1.0),        # a single K slice through the 128 x 128 kernel
])
def test_split_fp16_gemm_matches_float64_product(dev, dtype, S, N, W, use_idx, scale):
    """csrc/gemm_split.hip: the bilinear shared-negative forward on the fp16 matrix cores must be
    as close to the exact product as the fp32 MFMA kernel it replaces (both within 2e-6 of
    max|out|; the reference tolerance is rtol 1e-4 / atol 1e-5), for every shape class the
    dispatcher sends there; without a workspace the same entry point is the fp32 kernel."""
    import ctypes

    from besskge import _native as nat

    g = torch.Generator().manual_seed(S + N + W)
    M = max(N, 5000)
    table = (torch.randn(M, W, generator=g) * scale).to(dtype).to(dev)
    q = (torch.randn(S, W, generator=g) * scale).to(dev)
    d = nat.make_desc(nat.DISTMULT, 0, table, W)
    assert nat.load().bess_neg_score_shared_workspace(ctypes.byref(d), S, N) > 0, "shape must take the split path"
    idx = torch.randint(M, (N,), generator=g, dtype=torch.int32).to(dev) if use_idx else None
    neg = nat.RowSource(table, idx) if use_idx else nat.RowSource(table[:N].contiguous(), None)
    got = nat.neg_score_shared_fwd(d, q, neg)
    rows = table[idx.long()] if use_idx else table[:N]
    exact = nat.load()  # the entry point without scratch = exact fp32 MFMA kernel
    out32 = torch.empty((S, N), dtype=torch.float32, device=dev)
    rc = exact.bess_neg_score_shared_fwd(ctypes.byref(d), q.data_ptr(), S, neg.base.data_ptr(),
                                         idx.data_ptr() if use_idx else None, N, out32.data_ptr(), N, None)
    assert rc == 0
    torch.cuda.synchronize()
    for lo in range(0, S, 1024):  # float64 reference in row blocks
        ref = q[lo:lo + 1024].double() @ rows.double().T
        bound = 2e-6 * float(ref.abs().max())
        assert float((got[lo:lo + 1024].double() - ref).abs().max()) <= bound
        assert float((out32[lo:lo + 1024].double() - ref).abs().max()) <= bound


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("S,N,W,scale", [
    (4096, 4096, 256, 0.1),    # k split 4 / 4
    (2051, 2309, 100, 1.0),    # ragged everything, W < 128
    (1300, 9000, 384, 3e-3),   # different k splits for the two products
])
def test_split_fp16_gemm_backward_matches_float64_products(dev, dtype, S, N, W, scale):
    """Backward of the bilinear shared-negative scores through csrc/gemm_split.hip (transposed split
    images, k split with a fixed-order sum): d_query = G E and d_neg = G^T Q within 2e-6 of their
    largest element, like the exact fp32 MFMA kernels they replace; and bit-identical when repeated."""
    import ctypes

    from besskge import _native as nat

    g = torch.Generator().manual_seed(S * 3 + N + W)
    M = 12000
    table = (torch.randn(M, W, generator=g) * scale).to(dtype).to(dev)
    q = (torch.randn(S, W, generator=g) * scale).to(dev)
    go = torch.randn(S, N, generator=g).to(dev)
    idx = torch.randint(M, (N,), generator=g, dtype=torch.int32).to(dev)
    d = nat.make_desc(nat.DISTMULT, 0, table, W)
    assert nat.load().bess_neg_score_shared_bwd_workspace(ctypes.byref(d), S, N) > 0, "shape must take the split path"
    neg = nat.RowSource(table, idx)
    out = torch.zeros(S, N, device=dev)  # unused by dot products
    dq, dn = nat.neg_score_shared_bwd(d, q, neg, out, go)
    dq2, dn2 = nat.neg_score_shared_bwd(d, q, neg, out, go)
    assert torch.equal(dq, dq2) and torch.equal(dn, dn2)
    rows = table[idx.long()].double()
    rq = go.double() @ rows
    rn = go.double().T @ q.double()
    assert float((dq.double() - rq).abs().max()) <= 2e-6 * float(rq.abs().max())
    assert float((dn.double() - rn).abs().max()) <= 2e-6 * float(rn.abs().max())


def test_split_gemm_training_step_under_graph_capture(dev):
    """A whole training step whose shared-negative products take the split-fp16 path (scratch from
    torch's allocator, k-split backward): captured in a hipGraph and replayed it must give the first loss of the eager run bit for bit, and the
    later losses and the tables up to the rounding order of the atomic row updates."""
    import ctypes

    from besskge import _native as nat
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.embedding import init_KGE_normal
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import DistMult
    from besskge.sharding import Sharding

    n_ent, n_rel, d, S, K = 6000, 11, 256, 2048, 2048
    sharding = Sharding.create(n_ent, 1, seed=0)
    ns = RandomShardedNegativeSampler(K, sharding, 0, "t", local_sampling=False, flat_negative_format=True)
    rng = np.random.default_rng(3)
    batch = dict(head=rng.integers(n_ent, size=(1, 1, S)), relation=rng.integers(n_rel, size=(1, 1, S)),
                 tail=rng.integers(n_ent, size=(1, 1, S)), negative=rng.integers(n_ent, size=(1, 1, 1, K)))
    batch = {k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in batch.items()}

    def run(graphs):
        torch.manual_seed(0)
        fn = DistMult(True, sharding, n_rel, d, [init_KGE_normal], [init_KGE_normal])
        with torch.no_grad():
            fn.entity_embedding.mul_(30.0)  # scores of O(1)
            fn.relation_embedding.mul_(30.0)
        model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn,
                                       loss_fn=LogSigmoidLoss(margin=1.0, negative_adversarial_sampling=True))
        runner = runtime.training_model(model, runtime.Options(device_iterations=1, use_graphs=graphs),
                                        runtime.SGD(lr=0.05), device=dev)
        losses = [runner(**batch)["loss"].clone() for _ in range(3)]
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), fn.entity_embedding.detach().cpu().clone(), \
            fn.relation_embedding.detach().cpu().clone()

    d_ = nat.ModelDesc()
    d_.scorer, d_.norm_p, d_.dtype, d_.width, d_.rel_width = nat.DISTMULT, 0, 0, d, d
    assert nat.load().bess_neg_score_shared_workspace(ctypes.byref(d_), S, K) > 0
    assert nat.load().bess_neg_score_shared_bwd_workspace(ctypes.byref(d_), S, K) > 0
    l_e, ent_e, rel_e = run(False)
    l_g, ent_g, rel_g = run(True)
    assert torch.isfinite(l_e).all() and float(l_e[2]) < float(l_e[0])  # it trains
    assert torch.equal(l_e[0], l_g[0])  # forward and loss from identical tables: deterministic kernels only
    torch.testing.assert_close(l_e, l_g, rtol=1e-6, atol=0)  # later steps see the rounding of the atomic updates
    # the row updates go through float atomics (duplicate rows among heads / tails / negatives): order-dependent rounding
    torch.testing.assert_close(ent_e, ent_g, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rel_e, rel_g, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name,p", NATIVE_SCORERS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_grad_segments_with_rows_that_many_references_point_at(dev, name, p, dtype):
    """Rows with more than BESS_SEGMENT_CAP references (padded candidate lists, hot entities) are
    reduced by the whole device through partial sums; the result - gradient rows and the fused SGD
    step - is that of the per-reference backward scattered by index_add, as for ordinary rows."""
    from besskge import _native as nat
    from besskge._native import RowSource

    gen = torch.Generator().manual_seed(11)
    M, d, S, N = 500, 24, 150, 40   # 6000 references
    W, Wr = widths(name, d)
    table = torch.randn(M, W, generator=gen).to(dtype).to(dev)
    q = torch.randn(S, W, generator=gen).to(dev)
    idx = torch.randint(M, (S * N,), generator=gen, dtype=torch.int32)
    sel = torch.rand(S * N, generator=gen)
    idx[sel < 0.25] = 3          # ~1500 references: 6 slices of 256
    idx[(sel >= 0.25) & (sel < 0.31)] = 77   # ~360 references: 2 slices
    idx = idx.to(dev)
    go = (torch.randn(S, N, generator=gen) * 0.1).to(dev)
    desc = nat.make_desc(dict(TransE=0, RotatE=1, DistMult=2, ComplEx=3)[name], max(p, 1), table, Wr)
    _, dn = nat.neg_score_pertriple_bwd(desc, q, RowSource(table, idx), N, go)
    seg = nat.SegmentIndex(idx, M)
    n_seg = int(seg.n_seg.item())
    assert int(seg.long_segs[0].item()) == 2
    assert sorted(seg.seg_rows[seg.long_segs[1:3].long()].cpu().tolist()) == [3, 77]
    uniq = torch.unique(idx.cpu().long())
    want = torch.zeros(M, W, dtype=torch.float64).index_add_(0, idx.cpu().long(), dn.cpu().double())
    g1 = nat.neg_pertriple_grad_segments(desc, q, table, N, go, seg)
    close(g1[:n_seg], want[uniq].float(), rtol=1e-4, atol=1e-5, scale=2e-6)
    t2 = table.clone()
    nat.neg_pertriple_grad_segments(desc, q, t2, N, go, seg, fused_sgd_lr=0.5)
    want_t = table.float().cpu() - 0.5 * want.float()
    close(t2, want_t, rtol=2e-3 if dtype == torch.float16 else 1e-5, atol=4e-3 if dtype == torch.float16 else 1e-5)


@pytest.mark.parametrize("n_part,normalize,p", [(1, True, 1), (2, True, 2), (2, False, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_affine_grad_segments_with_rows_that_many_references_point_at(dev, n_part, normalize, p, dtype):
    """The long-row tier of the affine family: slices summed through atomics, the normalisation
    backward applied once to the total - same result as index_add of the per-reference backward."""
    from besskge import _native as nat
    from besskge._native import RowSource

    gen = torch.Generator().manual_seed(13)
    M, d, S, N = 400, 24, 150, 40
    W = n_part * d
    table = torch.randn(M, W, generator=gen).to(dtype).to(dev)
    q = torch.randn(S, (n_part + 1) * d, generator=gen).to(dev)
    idx = torch.randint(M, (S * N,), generator=gen, dtype=torch.int32)
    sel = torch.rand(S * N, generator=gen)
    idx[sel < 0.2] = 5
    idx[(sel >= 0.2) & (sel < 0.27)] = 91
    idx = idx.to(dev)
    go = (torch.randn(S, N, generator=gen) * 0.1).to(dev)
    desc = nat.make_desc(nat.AFFINE, p, table, d)
    desc.reserved[0], desc.reserved[1] = n_part, int(normalize)
    _, dn = nat.neg_score_pertriple_bwd(desc, q, RowSource(table, idx), N, go)
    seg = nat.SegmentIndex(idx, M, width=W)
    n_seg = int(seg.n_seg.item())
    assert int(seg.long_segs[0].item()) == 2
    uniq = torch.unique(idx.cpu().long())
    want = torch.zeros(M, W, dtype=torch.float64).index_add_(0, idx.cpu().long(), dn.cpu().double())
    g1 = nat.neg_pertriple_grad_segments(desc, q, table, N, go, seg)
    close(g1[:n_seg], want[uniq].float(), rtol=1e-4, atol=1e-5, scale=4e-6)
    t2 = table.clone()
    nat.neg_pertriple_grad_segments(desc, q, t2, N, go, seg, fused_sgd_lr=0.5)  # second use of the scratch rows
    tol = 4e-3 if dtype == torch.float16 else 1e-5
    close(t2, table.float().cpu() - 0.5 * want.float(), rtol=tol, atol=tol, scale=4e-6)


@pytest.mark.parametrize("tanh,per_dim,p", [(True, True, 1), (True, False, 2), (False, True, 2), (False, False, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_boxe_grad_segments_match_scatter_of_row_gradients(dev, tanh, per_dim, p, dtype):
    """K9 of BoxE (csrc/boxe.hip k_box_grad_segments, incl. the long-row tier): the per-row sums of
    d score / d e recomputed from the six query vectors equal index_add of the backward kernel's
    per-reference gradients; the fused SGD step is the same update."""
    from besskge import _native as nat
    from besskge._native import RowSource

    gen = torch.Generator().manual_seed(17)
    for M, d, S, N, hot in ((300, 40, 33, 17, False), (400, 24, 150, 40, True), (64, 6, 9, 5, False)):
        W = 2 * d
        table = (torch.randn(M, W, generator=gen) * 0.7).to(dtype).to(dev)
        q = torch.randn(S, 6 * d, generator=gen)
        q[:, 2 * d:3 * d].abs_()   # half widths H_0, H_1 are positive
        q[:, 5 * d:].abs_()
        q = q.to(dev)
        idx = torch.randint(M, (S * N,), generator=gen, dtype=torch.int32)
        if hot:
            idx[torch.rand(S * N, generator=gen) < 0.2] = 5
        idx = idx.to(dev)
        go = (torch.randn(S, N, generator=gen) * 0.1).to(dev)
        desc = nat.make_desc(nat.BOXE, p, table, W)
        desc.reserved[0] = int(tanh) | (int(per_dim) << 1)
        _, dn = nat.neg_score_pertriple_bwd(desc, q, RowSource(table, idx), N, go)
        seg = nat.SegmentIndex(idx, M, width=W)
        n_seg = int(seg.n_seg.item())
        assert int(seg.long_segs[0].item()) == (1 if hot else 0)
        uniq = torch.unique(idx.cpu().long())
        want = torch.zeros(M, W, dtype=torch.float64).index_add_(0, idx.cpu().long(), dn.cpu().double())
        g1 = nat.neg_pertriple_grad_segments(desc, q, table, N, go, seg)
        close(g1[:n_seg], want[uniq].float(), rtol=1e-4, atol=1e-5, scale=4e-6)
        t2 = table.clone()
        nat.neg_pertriple_grad_segments(desc, q, t2, N, go, seg, fused_sgd_lr=0.5)
        tol = 4e-3 if dtype == torch.float16 else 1e-5
        close(t2, table.float().cpu() - 0.5 * want.float(), rtol=tol, atol=tol, scale=4e-6)


@pytest.mark.parametrize("scorer", ["TransE", "RotatE", "DistMult", "ComplEx"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("side", [0, 1])
def test_query_and_positive_score_in_one_launch(dev, scorer, dtype, side):
    """bess_query_triple_fwd / _bwd == bess_query_fwd + bess_score_triple_fwd (and the sum of their
    backwards on the row the query was built from), bit for bit in the forward."""
    from besskge import _native as nat

    torch.manual_seed(5)
    S, M, d, n_rel = 77, 200, 48, 9
    W = 2 * d if scorer in ("RotatE", "ComplEx") else d
    Wr = W if scorer != "RotatE" else d
    table = torch.randn(M, W).to(dtype).to(dev)
    other = torch.randn(S, W).to(dtype).to(dev)  # tails that came through an exchange: identity rows
    rel = torch.randn(n_rel, Wr).to(dtype).to(dev)
    hidx = torch.randint(0, M, (S,), dtype=torch.int32, device=dev)
    ridx = torch.randint(0, n_rel, (S,), dtype=torch.int32, device=dev)
    dsc = nat.make_desc(dict(TransE=nat.TRANSE, RotatE=nat.ROTATE, DistMult=nat.DISTMULT, ComplEx=nat.COMPLEX)[scorer],
                        1, table, Wr)
    head, tail = nat.RowSource(table, hidx), nat.RowSource(other, None)
    ent = head if side == nat.CORRUPT_TAIL else tail
    q, pos = nat.query_triple_fwd(dsc, side, head, tail, rel, ridx)
    assert torch.equal(q, nat.query_fwd(dsc, side, ent, rel, ridx))
    assert torch.equal(pos, nat.score_triple_fwd(dsc, head, tail, rel, ridx))
    d_pos, dq = torch.randn(S, device=dev), torch.randn(S, W, device=dev)
    dr = torch.zeros(n_rel, Wr, device=dev)
    dh, dt = nat.query_triple_bwd(dsc, side, head, tail, rel, ridx, d_pos, dq, dr)
    dr2 = torch.zeros_like(dr)
    dh2, dt2 = nat.score_triple_bwd(dsc, head, tail, rel, ridx, d_pos, dr2)
    dx = nat.query_bwd(dsc, side, ent, rel, ridx, dq, dr2)
    if side == nat.CORRUPT_TAIL:
        dh2 = dh2 + dx
    else:
        dt2 = dt2 + dx
    torch.testing.assert_close(dh, dh2, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(dt, dt2, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(dr, dr2, rtol=1e-5, atol=1e-5)  # atomics: order of the adds differs


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_sparse_sgd_of_several_lists_in_one_launch(dev, dtype):
    from besskge import _native as nat

    torch.manual_seed(6)
    M, W = 120, 64
    table = torch.randn(M, W).to(dtype).to(dev)
    lists = [(torch.randint(0, M, (n,), dtype=torch.int32, device=dev), torch.randn(n, W, device=dev) * 0.1)
             for n in (50, 1, 33)]
    a, b = table.clone(), table.clone()
    nat.sparse_sgd_lists(a, lists, 0.5)
    for idx, g in lists:
        nat.sparse_sgd(b, idx, g, 0.5)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=5e-3, atol=5e-3)
    torch.testing.assert_close(a.float(), b.float(), **tol)
    want = table.float().cpu().double()
    for idx, g in lists:
        want.index_add_(0, idx.cpu().long(), -0.5 * g.cpu().double())
    torch.testing.assert_close(a.float().cpu().double(), want, **tol)


@pytest.mark.parametrize("where", ["query", "candidate", "score_gradient", "inf"])
def test_split_fp16_gemm_falls_back_to_fp32_outside_the_fp16_range(dev, where):
    """An operand of 65504 or more in magnitude (or a non-finite one) cannot be split into fp16 pairs: the
    pre-pass raises a flag, the split kernels return at once and the exact fp32 MFMA kernels queued behind them
    compute the product - same call, no host synchronisation, same accuracy as for in-range operands."""
    import ctypes

    from besskge import _native as nat

    g = torch.Generator().manual_seed(17)
    S, N, W, M = 2048, 2304, 128, 6000
    table = torch.randn(M, W, generator=g)
    q = torch.randn(S, W, generator=g)
    go = torch.randn(S, N, generator=g)
    idx = torch.randint(M, (N,), generator=g, dtype=torch.int32)
    big = 3.0e5 if where != "inf" else float("inf")
    if where in ("query", "inf"):
        q[S - 3, 5] = big
    elif where == "candidate":
        table[int(idx[N - 1]), W - 1] = -big  # reached through the index, in the last chunk of rows
    else:
        go[7, N - 2] = big
    table, q, go, idx = table.to(dev), q.to(dev), go.to(dev), idx.to(dev)
    d = nat.make_desc(nat.DISTMULT, 0, table, W)
    assert nat.load().bess_neg_score_shared_workspace(ctypes.byref(d), S, N) > 0
    neg = nat.RowSource(table, idx)
    rows = table[idx.long()].double()
    got = nat.neg_score_shared_fwd(d, q, neg)
    ref = q.double() @ rows.T
    if where == "inf":
        assert bool(torch.isinf(got[S - 3]).any() or torch.isnan(got[S - 3]).any())  # what fp32 arithmetic gives
        keep = torch.ones(S, dtype=torch.bool, device=dev)
        keep[S - 3] = False
        assert float((got[keep].double() - ref[keep]).abs().max()) <= 2e-6 * float(ref[keep].abs().max())
        return
    if where != "score_gradient":
        assert float((got.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    dq, dn = nat.neg_score_shared_bwd(d, q, neg, got, go)
    rq, rn = go.double() @ rows, go.double().T @ q.double()
    # (the fp32 chain over ~2000 terms is a little less accurate than the split path it stands in for)
    assert float((dq.double() - rq).abs().max()) <= 4e-6 * float(rq.abs().max())
    assert float((dn.double() - rn).abs().max()) <= 4e-6 * float(rn.abs().max())
    # and the descriptor flag asks for the fp32 kernels outright
    d.reserved[0] = nat.FLAG_FP32_MATH
    assert nat.load().bess_neg_score_shared_workspace(ctypes.byref(d), S, N) == 0
    got32 = nat.neg_score_shared_fwd(d, q, neg)
    assert float((got32.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize("p,dtype", [(1, torch.float32), (2, torch.float32), (1, torch.float16), (2, torch.float16)])
@pytest.mark.parametrize("S,N,W", [(512, 1024, 256),   # 64-row tiles, reduction split 16 / 8 ways
                                   (512, 544, 256),    # the notebook's micro-batch: 32-row tiles
                                   (70, 1300, 100),    # 32-row tiles, ragged rows and columns, scalar loads
                                   (1024, 70, 400),    # one slice only on the dE side
                                   (1024, 1030, 64),   # 272 forward tiles: the 64-row forward tile (the others: 32-row)
                                   (33, 17, 64)])
def test_shared_distance_backward_tile_variants(dev, p, dtype, S, N, W):
    """k_neg_shared_bwd (distance scorers, shared negatives): both tile heights and every split the launcher's
    planner picks, against the float64 gradients of -||q - e||_p (sgn(0) = 0, the p = 2 gradient 0 at zero
    distance; reference scoring.py:94-128 + torch autograd)."""
    from besskge import _native as nat

    g = torch.Generator().manual_seed(S * 7 + N)
    M = 2000
    table = (torch.randn(M, W, generator=g) * 0.5).to(dtype).to(dev)
    q = (torch.randn(S, W, generator=g) * 0.5).half().float().to(dev)  # fp16-exact queries: any rounding mode agrees
    idx = torch.randint(M, (N,), generator=g, dtype=torch.int32).to(dev)
    go = (torch.softmax(torch.randn(S, N, generator=g) * 2, -1) * torch.rand(S, 1, generator=g)).to(dev)
    d = nat.make_desc(nat.TRANSE, p, table, W)
    src = nat.RowSource(table, idx)
    out = nat.neg_score_shared_fwd(d, q, src)
    dq, dn = nat.neg_score_shared_bwd(d, q, src, out, go)
    rows = table[idx.long()].double()
    qd = q.double()
    want_q = torch.zeros(S, W, dtype=torch.float64, device=dev)
    want_n = torch.zeros(N, W, dtype=torch.float64, device=dev)
    want_out = torch.zeros(S, N, dtype=torch.float64, device=dev)
    for a0 in range(0, S, 64):  # [64, N, W] float64 at a time
        diff = qd[a0:a0 + 64, None, :] - rows[None, :, :]
        if p == 1:
            coef = torch.sign(diff)
            want_out[a0:a0 + 64] = -diff.abs().sum(-1)
        else:
            nrm = diff.norm(dim=-1, keepdim=True)
            coef = torch.where(nrm > 0, diff / nrm.clamp(min=1e-300), torch.zeros_like(diff))
            want_out[a0:a0 + 64] = -nrm[..., 0]
        t = go[a0:a0 + 64].double()[:, :, None] * coef
        want_q[a0:a0 + 64] = -t.sum(1)
        want_n += t.sum(0)
    tol = 2e-5 if p == 2 else 1e-5  # p = 2 divides by the fp32 forward score
    assert float((out.double() - want_out).abs().max()) <= 2e-6 * float(want_out.abs().max())
    assert float((dq.double() - want_q).abs().max()) <= tol * float(want_q.abs().max())
    assert float((dn.double() - want_n).abs().max()) <= tol * float(want_n.abs().max())


@pytest.mark.parametrize("case,fused", [("tr_SM_ComplEx0_t_pt_n2", True),   # log-sigmoid, per-triple negatives
                                        ("tr_SM_ht_pt_n2", True),           # sampled softmax, two groups per shard
                                        ("tr_SM_ht_flat_n2", False)])       # margin ranking: needs the positive score
def test_score_moving_training_forward_keeps_its_partials(dev, case, fused):
    """ScoreMoving training with per-triple negatives: every shard scores the gathered queries against its own
    rows ONCE (`bess_neg_score_pertriple_fwd_partials`) and later rescales the partials it kept
    (`bess_combine_dq_partials`) - no second pass over the negative rows (`bess_neg_score_pertriple_bwd`) -
    and still moves the tables by the reference's gradients (reference bess.py:490-603 + autograd)."""
    from besskge import _native as nat
    from besskge import runtime

    c = load_bess_case(case)
    lr = 0.125
    model = build_model(c, dev)
    runner = runtime.training_model(model, runtime.Options(device_iterations=1), runtime.SGD(lr=lr), device=dev)
    keys = ("head", "relation", "tail", "negative", "negative_mask")
    watched = ["bess_neg_score_pertriple_fwd_partials", "bess_neg_score_pertriple_bwd", "bess_neg_score_pertriple_fwd"]
    nat.start_kernel_timing(watched)
    runner(**{k: c["batch"][k][0] for k in keys if k in c["batch"]})
    calls = {k: len(v) for k, v in nat.stop_kernel_timing().items()}
    if fused:
        assert calls.get("bess_neg_score_pertriple_fwd_partials", 0) > 0, calls
        assert calls.get("bess_neg_score_pertriple_bwd", 0) == 0 and calls.get("bess_neg_score_pertriple_fwd", 0) == 0, calls
    else:
        assert calls.get("bess_neg_score_pertriple_fwd_partials", 0) == 0, calls
    close(model.score_fn.entity_embedding, c["table"] - lr * c["grads"]["entity"], rtol=1e-4, atol=2e-5)
    close(model.score_fn.relation_embedding, c["rel"] - lr * c["grads"]["relation"].sum(0), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("case,fused", [("tr_EM_ComplEx0_ht_pt_n2", True), ("tr_EM_DistMult0_ht_pt_n2", True),
                                        ("tr_EM_TransE1_ht_pt_n2", False)])
def test_embedding_moving_fused_forward_over_received_rows(dev, case, fused):
    """EmbeddingMoving on two shards, per-triple negatives that arrived through the all-to-all: the bilinear scorers
    take scores and d loss / d query from ONE pass over the received rows (`bess_neg_score_pertriple_fwd_dq`) and
    their backward writes d_neg = coefficient x query without reading a candidate row; the distance scorers keep
    the two-pass form (sgn(q - e) needs the rows again).  Tables move by the reference's gradients either way
    (reference bess.py:340-468 + autograd)."""
    from besskge import _native as nat
    from besskge import runtime

    c = load_bess_case(case)
    lr = 0.125
    model = build_model(c, dev)
    runner = runtime.training_model(model, runtime.Options(device_iterations=1), runtime.SGD(lr=lr), device=dev)
    keys = ("head", "relation", "tail", "negative", "negative_mask")
    watched = ["bess_neg_score_pertriple_fwd_dq", "bess_neg_score_pertriple_fwd", "bess_neg_score_pertriple_bwd"]
    nat.start_kernel_timing(watched)
    runner(**{k: c["batch"][k][0] for k in keys if k in c["batch"]})
    calls = {k: len(v) for k, v in nat.stop_kernel_timing().items()}
    if fused:
        assert calls.get("bess_neg_score_pertriple_fwd_dq", 0) > 0 and calls.get("bess_neg_score_pertriple_fwd", 0) == 0, calls
    else:
        assert calls.get("bess_neg_score_pertriple_fwd_dq", 0) == 0 and calls.get("bess_neg_score_pertriple_fwd", 0) > 0, calls
    assert calls.get("bess_neg_score_pertriple_bwd", 0) > 0, calls
    close(model.score_fn.entity_embedding, c["table"] - lr * c["grads"]["entity"], rtol=1e-4, atol=2e-5)
    close(model.score_fn.relation_embedding, c["rel"] - lr * c["grads"]["relation"].sum(0), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("kind,adv", [("logsigmoid", True), ("logsigmoid", False), ("ssce", False)])
@pytest.mark.parametrize("name,p", [("ComplEx", 0), ("TransE", 1), ("RotatE", 2)])
def test_partials_of_two_shards_combine_to_the_one_pass_gradient(dev, name, p, kind, adv):
    """The negatives of every query split over two 'shards': scores + partials per part
    (`bess_neg_score_pertriple_fwd_partials`), the loss kernel's row normalisation over the concatenated scores
    (`bess_loss_fwd_bwd_norm`), each part rescaled (`bess_combine_dq_partials`) - the parts add up to the
    d loss / d query of the one-pass fused forward over all negatives, and the norm is what its header says."""
    from besskge import _native as nat
    from besskge.loss import LogSigmoidLoss, SampledSoftmaxCrossEntropyLoss

    torch.manual_seed(5)
    S, K1, K2, d_emb, M = 70, 40, 24, 32, 500
    W, Wr = widths(name, d_emb)
    table = (0.3 * torch.randn(M, W)).to(dev)
    desc = nat.make_desc(dict(TransE=0, RotatE=1, DistMult=2, ComplEx=3)[name], max(p, 1), table, Wr)
    q = torch.randn(S, W, device=dev) * 0.3
    idx = torch.randint(M, (S, K1 + K2), device=dev, dtype=torch.int32)
    pos = torch.randn(S, device=dev)
    w = torch.rand(S, device=dev) + 0.1
    loss_fn = (SampledSoftmaxCrossEntropyLoss(n_entity=10 * M) if kind == "ssce"
               else LogSigmoidLoss(margin=1.5, negative_adversarial_sampling=adv, negative_adversarial_scale=0.7))
    ld = loss_fn.kernel_desc(K1 + K2)
    full = nat.RowSource(table, idx.reshape(-1).contiguous())
    sc_full, dq_full = nat.neg_score_pertriple_fwd_dq(desc, ld, q, full, K1 + K2, pos, w)
    parts = []
    for lo, hi in ((0, K1), (K1, K1 + K2)):
        src = nat.RowSource(table, idx[:, lo:hi].reshape(-1).contiguous())
        parts.append(nat.neg_score_pertriple_fwd_partials(desc, ld, q, src, hi - lo))
    sc = torch.cat([parts[0][0], parts[1][0]], dim=1).contiguous()
    assert torch.equal(sc, sc_full)
    _, _, _, norm = nat.loss_fwd_bwd(ld, pos, sc, w, True, want_norm=True)
    dq = nat.combine_dq_partials(parts[0][1], norm) + nat.combine_dq_partials(parts[1][1], norm)
    torch.testing.assert_close(dq, dq_full, rtol=2e-5, atol=1e-6 * float(dq_full.abs().max()))
    # the normalisation itself, in float64
    beta = 1.0 if kind == "ssce" else (0.7 if adv else 0.0)
    z = beta * (sc.double() + (float(ld.ssce_shift) if kind == "ssce" else 0.0))
    m = z.max(1).values
    if kind == "ssce":
        m = torch.maximum(m, pos.double())
    big_l = torch.exp(z - m[:, None]).sum(1) + (torch.exp(pos.double() - m) if kind == "ssce" else 0.0)
    c = (1.0 if kind == "ssce" else 0.5) * float(ld.loss_scale) * w.double()
    torch.testing.assert_close(norm[:, 0].double(), m, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(norm[:, 1].double(), big_l / c, rtol=1e-5, atol=0)


@pytest.mark.parametrize("mask_rows", ["one", "per_triple"])
@pytest.mark.parametrize("kind", ["logsigmoid", "ssce", "margin"])
@pytest.mark.parametrize("name,p,dtype", [("ComplEx", 0, torch.float32), ("TransE", 1, torch.float16), ("RotatE", 2, torch.float32)])
def test_fused_forward_with_the_mask_inside_matches_mask_then_two_pass(dev, name, p, dtype, kind, mask_rows):
    """bess_neg_score_pertriple_fwd_dq_masked: the padding mask of triple-specific negatives applied inside the fused
    pass == scores, bess_mask_scores, loss kernel, backward kernel (bess.py:182-245 then loss.py:28-251)."""
    from besskge import _native as nat
    from besskge.loss import LogSigmoidLoss, MarginRankingLoss, SampledSoftmaxCrossEntropyLoss

    gen = torch.Generator().manual_seed(21)
    S, N, cols, d_emb, M = 70, 96, 80, 24, 400
    W, Wr = widths(name, d_emb)
    table = (0.4 * torch.randn(M, W, generator=gen)).to(dtype).to(dev)
    desc = nat.make_desc(dict(TransE=0, RotatE=1, DistMult=2, ComplEx=3)[name], max(p, 1), table, Wr)
    q = (0.4 * torch.randn(S, W, generator=gen)).to(dev)
    idx = torch.randint(M, (S * N,), generator=gen, dtype=torch.int32).to(dev)
    pos = torch.randn(S, generator=gen).to(dev)
    w = (torch.rand(S, generator=gen) + 0.5).to(dev)
    rows = 1 if mask_rows == "one" else S
    mask = (torch.rand(rows, cols, generator=gen) < 0.7).to(dev)  # over the LAST `cols` columns
    fn = {"logsigmoid": LogSigmoidLoss(1.0, True, 0.8), "margin": MarginRankingLoss(2.0, True, 0.8),
          "ssce": SampledSoftmaxCrossEntropyLoss(10 * M)}[kind]
    ld = fn.kernel_desc(N)
    neg = nat.RowSource(table, idx)
    out, dq = nat.neg_score_pertriple_fwd_dq(desc, ld, q, neg, N, pos, w, mask=mask)
    ref = nat.neg_score_pertriple_fwd(desc, q, neg, N)
    nat.mask_scores(ref, 0, False, 0, mask)
    assert torch.equal(out, ref)
    assert int((out < -40000).sum()) > 0
    _, _, dn = nat.loss_fwd_bwd(ld, pos, ref, w, True)
    dq_ref, _ = nat.neg_score_pertriple_bwd(desc, q, neg, N, dn, want_d_neg=False)
    close(dq, dq_ref, rtol=2e-4, atol=1e-6, scale=4e-6)


@pytest.mark.parametrize("rows", ["one", "per_triple"])
def test_training_step_with_a_negative_mask_takes_the_fused_forward(dev, rows):
    """EmbeddingMoving, one shard, per-triple negatives + negative_mask: the fused training forward applies the mask
    inside its pass (no second pass over the negative rows) and moves the tables like the two-pass path does."""
    from besskge import _native as nat
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import ComplEx
    from besskge.sharding import Sharding

    S, K, n_ent, n_rel, d_emb = 48, 20, 600, 7, 16
    rng = np.random.default_rng(3)
    batch = dict(head=rng.integers(n_ent, size=(1, 1, S)), relation=rng.integers(n_rel, size=(1, 1, S)),
                 tail=rng.integers(n_ent, size=(1, 1, S)), negative=rng.integers(n_ent, size=(1, 1, S, K)))
    batch = {k: torch.from_numpy(v.astype(np.int32)) for k, v in batch.items()}
    batch["negative_mask"] = torch.from_numpy(rng.random((1, 1 if rows == "one" else S, 1, K)) < 0.7)

    def run(fused: bool):
        torch.manual_seed(1)
        sharding = Sharding.create(n_ent, 1, seed=2)
        fn = ComplEx(False, sharding, n_rel, d_emb)
        ns = RandomShardedNegativeSampler(K, sharding, 0, "h", local_sampling=False, flat_negative_format=False)
        model = EmbeddingMovingBessKGE(ns, fn, LogSigmoidLoss(margin=1.0, negative_adversarial_sampling=True))
        if not fused:
            model._fusable = lambda b: None  # type: ignore[method-assign]
        runner = runtime.training_model(model, optimizer=runtime.SGD(lr=0.1), device=dev)
        nat.start_kernel_timing(["bess_neg_score_pertriple_fwd_dq", "bess_neg_score_pertriple_bwd"])
        res = runner(**batch)
        calls = {k: len(v) for k, v in nat.stop_kernel_timing().items()}
        return res, model.score_fn.entity_embedding.detach().clone(), model.score_fn.relation_embedding.detach().clone(), calls

    res_f, ent_f, rel_f, calls_f = run(True)
    res_t, ent_t, rel_t, calls_t = run(False)
    assert calls_f.get("bess_neg_score_pertriple_fwd_dq", 0) == 1 and calls_f.get("bess_neg_score_pertriple_bwd", 0) == 0, calls_f
    assert calls_t.get("bess_neg_score_pertriple_fwd_dq", 0) == 0 and calls_t.get("bess_neg_score_pertriple_bwd", 0) == 1, calls_t
    torch.testing.assert_close(res_f["loss"], res_t["loss"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ent_f, ent_t, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(rel_f, rel_t, rtol=1e-5, atol=1e-7)
    assert float((ent_f - ent_t).abs().max()) < 1e-5 and float(ent_f.abs().max()) > 0


@pytest.mark.parametrize("name,p,dtype,d_emb", [("ComplEx", 0, torch.float32, 1000),   # W = 2000: two windows
                                                ("RotatE", 1, torch.float32, 1000),    # W = 2000, relation 1000
                                                ("DistMult", 0, torch.float32, 1027),  # odd width: 256-scalar windows
                                                ("TransE", 1, torch.float16, 2560)])   # f16: windows of 2048
def test_rows_wider_than_a_groups_registers_are_scored_in_column_windows(dev, name, p, dtype, d_emb):
    """Per-triple kernels on rows of more than 1024 f32 / 2048 f16 scalars (RotatE / ComplEx with embedding_size
    1000, the literature's setting): forward, backward and the segmented K9 against float64 torch; the fused
    training forward and the p = 2 distance say so loudly instead (BESS_EUNSUPPORTED)."""
    from besskge import _native as nat
    from besskge._native import RowSource
    from besskge.loss import LogSigmoidLoss

    gen = torch.Generator().manual_seed(d_emb)
    M, S, N = 200, 40, 24
    W, Wr = widths(name, d_emb)
    table = (0.2 * torch.randn(M, W, generator=gen)).to(dtype).to(dev)
    q = (0.2 * torch.randn(S, W, generator=gen)).to(dev)
    idx = torch.randint(M, (S * N,), generator=gen, dtype=torch.int32).to(dev)
    go = torch.randn(S, N, generator=gen).to(dev)
    desc = nat.make_desc(dict(TransE=0, RotatE=1, DistMult=2, ComplEx=3)[name], max(p, 1), table, Wr)
    assert not nat.row_fits_registers(desc)
    src = RowSource(table, idx)
    out = nat.neg_score_pertriple_fwd(desc, q, src, N)
    rows = table[idx.long()].double().view(S, N, W).requires_grad_(True)
    qq = q.double()[:, None, :].requires_grad_(True)
    ref = (qq * rows).sum(-1) if p == 0 else -(qq - rows).abs().sum(-1)
    ref.backward(go.double())
    close(out, ref.detach().float(), rtol=1e-5, atol=1e-5, scale=2e-6)
    dq, dn = nat.neg_score_pertriple_bwd(desc, q, src, N, go)
    close(dq, qq.grad[:, 0].float(), rtol=1e-5, atol=1e-5, scale=2e-6)
    close(dn, rows.grad.reshape(S * N, W).float(), rtol=1e-5, atol=1e-5, scale=2e-6)
    seg = nat.SegmentIndex(idx, M)
    n_seg = int(seg.n_seg.item())
    g1 = nat.neg_pertriple_grad_segments(desc, q, table, N, go, seg)
    want = torch.zeros(M, W, dtype=torch.float64, device=dev).index_add_(0, idx.long(), rows.grad.reshape(S * N, W))
    close(g1[:n_seg], want[seg.seg_rows[:n_seg].long()].float(), rtol=1e-4, atol=1e-5, scale=2e-6)
    # loud refusals
    ld = LogSigmoidLoss(1.0, True).kernel_desc(N)
    with pytest.raises(RuntimeError, match="fused training forward"):
        nat.neg_score_pertriple_fwd_dq(desc, ld, q, src, N, torch.zeros(S, device=dev), torch.ones(1, device=dev))
    if name in ("TransE", "RotatE"):
        d2 = nat.make_desc(dict(TransE=0, RotatE=1)[name], 2, table, Wr)
        with pytest.raises(RuntimeError, match="p = 2"):
            nat.neg_score_pertriple_fwd(d2, q, src, N)


def test_training_step_on_embedding_size_1000_matches_the_oracle(dev):
    """ComplEx with embedding_size 1000 (rows of 2000 scalars: column windows in the per-triple kernels, two-pass
    training forward), per-triple negatives, log-sigmoid loss, one SGD step against the CPU restatement."""
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import ComplEx
    from besskge.sharding import Sharding
    from oracle import kge

    torch.manual_seed(0)
    n_entity, n_rel, d_emb, S, K = 500, 9, 1000, 32, 12
    sharding = Sharding.create(n_entity, 1, seed=3)
    ent = torch.randn(1, sharding.max_entity_per_shard, 2 * d_emb) * 0.05
    rel = torch.randn(n_rel, 2 * d_emb) * 0.05
    fn = ComplEx(False, sharding, n_rel, d_emb, ent, rel)
    ns = RandomShardedNegativeSampler(K, sharding, 5, "h", local_sampling=False, flat_negative_format=False)
    model = EmbeddingMovingBessKGE(ns, fn, LogSigmoidLoss(margin=1.0, negative_adversarial_sampling=True))
    rng = np.random.default_rng(2)
    batch = dict(head=rng.integers(n_entity, size=(1, 1, S)), relation=rng.integers(n_rel, size=(1, 1, S)),
                 tail=rng.integers(n_entity, size=(1, 1, S)), negative=rng.integers(n_entity, size=(1, 1, S, K)))
    batch = {k: torch.from_numpy(v.astype(np.int32)) for k, v in batch.items()}
    spec = kge.StepSpec("ComplEx", 0, False, "h", False)
    t0, r0 = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
    want = kge.bess_step(spec, "EmbeddingMoving", t0, r0, batch,
                         dict(kind="logsigmoid", margin=1.0, adversarial=True, adversarial_scale=1.0))
    want["loss"][0].backward()
    lr = 0.5
    runner = runtime.training_model(model, optimizer=runtime.SGD(lr=lr), device=dev)
    res = runner(**batch)
    torch.testing.assert_close(res["loss"].float().cpu().reshape(()), want["loss"][0].detach(), rtol=2e-5, atol=1e-5)
    close(model.score_fn.entity_embedding, ent - lr * t0.grad, rtol=1e-4, atol=2e-6)
    close(model.score_fn.relation_embedding, rel - lr * r0.grad, rtol=1e-4, atol=2e-6)
    assert float(t0.grad.abs().max()) > 0
