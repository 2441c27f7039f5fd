"""next-1 / next-2 on the GPU: TopKQueryBessKGE, AllScoresBESS, AllScoresPipeline
and Evaluation against vectors produced by the reference (tests/golden/{topk,
allscores,metric}.npz), the reference's hand-computed metric answers
(reference tests/test_metric.py:14-75) and the unsharded CPU oracle
(the comparison the reference's tests/test_bess.py:278-423 and
tests/test_pipeline.py:42-206 make)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import kge  # noqa: E402

from conftest import load_golden  # noqa: E402
from test_hip_parity import make_scorer  # noqa: E402
from test_oracle import T  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


# ------------------------------------------------------------------ metrics ---
@pytest.mark.parametrize("worst_rank_infty", [True, False])
def test_metric_known_answers(dev, worst_rank_infty):
    from besskge.metric import Evaluation

    names = ["mrr", "hits@1", "hits@5"]
    pos = torch.tensor([2.1, 5.0, 5.9, 2.0], device=dev)
    neg = torch.tensor([[2.1, 3.1, 2.1, 5.2, 8.4], [9.8, 5.0, 1.0, 3.2, 5.0], [4.0, 2.3, 5.9, 3.1, 4.5],
                        [4.0, 2.3, 5.9, 3.1, 4.5]], device=dev)
    worst = 0.0 if worst_rank_infty else 1.0 / 6
    res = Evaluation(names, mode="pessimistic", worst_rank_infty=worst_rank_infty)
    out = res.dict_metrics_from_ranks(res.ranks_from_scores(pos.clone(), neg))
    torch.testing.assert_close(out["hits@1"].cpu(), torch.tensor([0.0, 0.0, 0.0, 0.0]))
    torch.testing.assert_close(out["hits@5"].cpu(), torch.tensor([0.0, 1.0, 1.0, 0.0]))
    torch.testing.assert_close(out["mrr"].cpu(), torch.tensor([worst, 1.0 / 4, 1.0 / 2, worst]))
    res = Evaluation(names, mode="optimistic", worst_rank_infty=worst_rank_infty)
    out = res.dict_metrics_from_ranks(res.ranks_from_scores(pos.clone(), neg))
    torch.testing.assert_close(out["hits@1"].cpu(), torch.tensor([0.0, 0.0, 1.0, 0.0]))
    torch.testing.assert_close(out["hits@5"].cpu(), torch.tensor([1.0, 1.0, 1.0, 0.0]))
    torch.testing.assert_close(out["mrr"].cpu(), torch.tensor([1.0 / 4, 1.0 / 2, 1.0, worst]))
    truth = torch.tensor([6, 0, 2], device=dev)
    cand = torch.tensor([[6, 1, 45, 33, 28], [5, 2, 12, 0, 44], [27, 9, 1, 6, 17]], device=dev)
    res = Evaluation(names, worst_rank_infty=worst_rank_infty)
    out = res.dict_metrics_from_ranks(res.ranks_from_indices(truth, cand))
    torch.testing.assert_close(out["hits@1"].cpu(), torch.tensor([1.0, 0.0, 0.0]))
    torch.testing.assert_close(out["hits@5"].cpu(), torch.tensor([1.0, 1.0, 0.0]))
    torch.testing.assert_close(out["mrr"].cpu(), torch.tensor([1.0, 1.0 / 4, worst]))


@pytest.mark.parametrize("mode", ["optimistic", "pessimistic", "average"])
@pytest.mark.parametrize("winf", [False, True])
@pytest.mark.parametrize("red", ["none", "sum"])
def test_metric_golden(dev, mode, winf, red):
    from besskge.metric import Evaluation

    g = load_golden("metric")
    ev = Evaluation(["mrr", "hits@1", "hits@5", "hits@10"], mode=mode, worst_rank_infty=winf, reduction=red,
                    return_ranks=True)
    key = f"{mode}_{int(winf)}_{red}_"
    assert list(ev.metrics.keys())[:3] == [str(x) for x in g[key + "names"]][:3]
    order = [list(ev.metrics.keys()).index(str(x)) for x in g[key + "names"]]
    r1 = ev.ranks_from_scores(T(g["pos"]).to(dev), T(g["cand"]).to(dev))
    r2 = ev.ranks_from_indices(T(g["truth"]).to(dev), T(g["ids"]).to(dev))
    torch.testing.assert_close(r1.cpu(), T(g[key + "ranks_scores"]))
    torch.testing.assert_close(r2.cpu(), T(g[key + "ranks_indices"]))
    s1 = ev.stacked_metrics_from_ranks(r1, T(g["mask"]).to(dev))
    s2 = ev.stacked_metrics_from_ranks(r2)
    torch.testing.assert_close(s1.cpu()[:, order], T(g[key + "stacked_scores"]))
    torch.testing.assert_close(s2.cpu()[:, order], T(g[key + "stacked_indices"]))


# ------------------------------------------------------- streaming top-k -----
@pytest.mark.parametrize("rows,L,kk", [(1, 1, 1), (7, 50, 6), (130, 1000, 11), (5, 4097, 64), (33, 63, 20),
                                       (6200, 1030, 11), (40, 9001, 33),
                                       (9, 3000, 65), (70, 5000, 101), (6200, 600, 128), (3, 100, 128)])  # two registers per lane
def test_topk_update_matches_torch(dev, rows, L, kk):
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(rows * L + kk)
    best_s = torch.full((rows, kk), -50000.0)
    best_i = torch.full((rows, kk), -1, dtype=torch.int32)
    all_s, all_i = [best_s.clone()], [best_i.clone()]
    bs, bi = best_s.to(dev), best_i.to(dev)
    base = 0
    for w in range(3):  # three windows merged one after the other
        sc = torch.randn(rows, L, generator=gen) * 10
        ids = (torch.randperm(L, generator=gen)[None, :] + base).to(torch.int32) if w == 1 else None
        mask = (torch.rand(rows, L, generator=gen) > 0.2) if w == 2 else None
        nat.topk_update(sc.to(dev), bs, bi, ids=None if ids is None else ids.to(dev), id_base=base,
                        mask=None if mask is None else mask.to(dev))
        eff = sc if mask is None else sc + (-50000.0) * (~mask).float()
        all_s.append(eff)
        all_i.append((torch.arange(L, dtype=torch.int32)[None, :] + base).expand(rows, L) if ids is None
                     else ids.expand(rows, L))
        base += L
    cat_s, cat_i = torch.cat(all_s, dim=1), torch.cat(all_i, dim=1)
    want = torch.topk(cat_s, kk, dim=1)
    torch.testing.assert_close(bs.cpu(), want.values)
    valid = want.values > -40000  # ids of the (tied) sentinel / masked tail are unspecified
    # ... and so is the order of exactly equal scores (torch.topk; with 1e7 random floats a few rows have one):
    # an id may differ only where its score equals a neighbour's (or the first score left out)
    v = want.values
    nxt = torch.topk(cat_s, min(kk + 1, cat_s.shape[1]), dim=1).values[:, -1:]
    tied = torch.zeros_like(valid)
    tied[:, 1:] |= v[:, 1:] == v[:, :-1]
    tied[:, :-1] |= v[:, :-1] == v[:, 1:]
    tied[:, -1:] |= v[:, -1:] == nxt
    check = valid & ~tied
    assert torch.equal(bi.cpu()[check], torch.take_along_dim(cat_i, want.indices, dim=1)[check])
    assert float(tied.float().mean()) < 0.01


@pytest.mark.parametrize("kk", [17, 100])
@pytest.mark.parametrize("rows,L,pad", [(9, 2051, True), (9, 2051, False), (6200, 700, True), (3, 5000, True)])
def test_topk_update_ties_keep_column_order(dev, rows, L, pad, kk):
    """Equal scores keep their left-to-right (then earlier-window) order on every code path of the
    kernel: 16-B loads (row-aligned, padded leading dimension) and scalar loads, four waves per
    row and one.  torch.topk leaves the order of ties unspecified; a stable sort is the reference."""
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(L + rows)
    bs = torch.full((rows, kk), -50000.0, device=dev)
    bi = torch.full((rows, kk), -1, dtype=torch.int32, device=dev)
    cols = []
    for w in range(2):
        sc = torch.randint(0, 40, (rows, L), generator=gen).float()  # many ties
        cols.append(sc)
        if pad:
            buf = torch.zeros(rows, (L + 3) // 4 * 4, device=dev)
            buf[:, :L] = sc.to(dev)
            view = buf[:, :L]
        else:
            view = sc.to(dev)
        nat.topk_update(view, bs, bi, id_base=w * L)
    allc = torch.cat(cols, dim=1)
    order = torch.sort(allc, dim=1, descending=True, stable=True).indices[:, :kk]
    assert torch.equal(bi.cpu().long(), order)
    torch.testing.assert_close(bs.cpu(), torch.take_along_dim(allc, order, dim=1))


# ------------------------------------------------------------- goldens ------
def load_query_case(fix, case):
    g = load_golden(fix)
    p = case + "_"
    meta = dict(zip((str(k) for k in g[p + "meta_keys"]), (int(v) for v in g[p + "meta_vals"])))
    kind, scorer, scheme, cand = (str(s) for s in g[p + "strs"])
    batch = {k[len(p + "batch_"):]: T(g[k]) for k in g.files if k.startswith(p + "batch_")}
    outs = {k[len(p + "out_"):]: T(g[k]) for k in g.files if k.startswith(p + "out_")}
    return dict(meta=meta, scorer=scorer, scheme=scheme, cand=cand, batch=batch, outs=outs,
                table=T(g[p + "entity_table"]), rel=T(g[p + "relation_table"]), triples=g[p + "triples"],
                neg_heads=g[p + "neg_heads"], neg_tails=g[p + "neg_tails"], sort_idx=g[p + "triple_sort_idx"])


def query_cases(fix):
    return [str(c) for c in load_golden(fix)["cases"]]


def rank_inputs(c, rows, shard_bs):
    """Seeded `rank_truth` [rows, shard_bs] and `rank_filter` [rows, 5, 2] for an allscores golden case (rows =
    micro-batches x shards): the rank-counting mode of AllScoresBESS, single-process and one process per shard."""
    gen = torch.Generator().manual_seed(7)
    n_entity = c["meta"]["n_entity"]
    truth = torch.randint(0, n_entity, (rows, shard_bs), generator=gen, dtype=torch.int32)
    qi = torch.stack([torch.randperm(shard_bs, generator=gen)[:5] for _ in range(rows)]).to(torch.int32)
    ent = torch.randint(0, n_entity, (rows, 5), generator=gen, dtype=torch.int32)
    ent = torch.where(ent == torch.gather(truth, 1, qi.long()), torch.full_like(ent, -1), ent)
    ent[:, 4] = -1  # padding
    return truth, torch.stack([torch.where(ent < 0, torch.full_like(qi, -1), qi), ent], dim=-1).contiguous()


def candidate_sampler(c):
    from besskge.negative_sampler import PlaceholderNegativeSampler, TripleBasedShardedNegativeSampler

    if c["cand"] == "all":
        return PlaceholderNegativeSampler(corruption_scheme=c["scheme"])
    ns = object.__new__(TripleBasedShardedNegativeSampler)
    ns.corruption_scheme, ns.local_sampling, ns.mask_on_gather = c["scheme"], False, True
    ns.flat_negative_format = c["cand"] == "flat"
    return ns


@pytest.mark.parametrize("case", query_cases("topk"))
def test_topk_query_golden(dev, case):
    from besskge import runtime
    from besskge.bess import TopKQueryBessKGE
    from besskge.metric import Evaluation

    c = load_query_case("topk", case)
    m = c["meta"]
    n, bps = m["n_shard"], m["bps"]
    flat = bool(m["flat"])
    from besskge.sharding import Sharding

    fn = make_scorer(c["scorer"], m["norm"], flat, m["n_rel"], m["d"], c["table"], c["rel"], torch.device("cpu"),
                     sharding=Sharding.create(m["n_entity"], n, seed=1234))
    ev = Evaluation(["mrr", "hits@1", "hits@5"], mode="average", reduction="none", return_ranks=True)
    model = TopKQueryBessKGE(k=m["k"], candidate_sampler=candidate_sampler(c), score_fn=fn, evaluation=ev,
                             return_scores=True, window_size=m["window"])
    runner = runtime.inference_model(model, runtime.Options(device_iterations=bps), device=dev)
    keys = ("relation", "head", "tail", "negative", "triple_mask", "negative_mask")
    res = runner(**{k: c["batch"][k].flatten(end_dim=1) for k in keys if k in c["batch"]})
    want_ids = c["outs"]["topk_global_id"].reshape(bps * n, -1, m["k"])
    want_sc = c["outs"]["topk_scores"].reshape(bps * n, -1, m["k"])
    got_ids = res["topk_global_id"].cpu().reshape(bps * n, -1, m["k"])
    got_sc = res["topk_scores"].float().cpu().reshape(bps * n, -1, m["k"])
    torch.testing.assert_close(got_sc, want_sc, rtol=1e-4, atol=1e-4)
    # ids must agree wherever neighbouring scores are not (numerically) tied
    gap = (want_sc[..., :-1] - want_sc[..., 1:]).abs()
    clear = torch.ones_like(want_sc, dtype=torch.bool)
    clear[..., :-1] &= gap > 1e-3
    clear[..., 1:] &= gap > 1e-3
    assert torch.equal(got_ids[clear].long(), want_ids[clear].long())
    # (candidate lists contain duplicate entities: exact ties, but of equal ids)
    assert float((got_ids.long() == want_ids.long()).float().mean()) > 0.98
    torch.testing.assert_close(res["ranks"].cpu().reshape(-1), c["outs"]["ranks"].reshape(-1))
    torch.testing.assert_close(res["metrics"].cpu().reshape(bps * n, 3, -1),
                               c["outs"]["metrics"].reshape(bps * n, 3, -1))


@pytest.mark.parametrize("case", query_cases("allscores"))
def test_all_scores_golden(dev, case):
    from besskge import runtime
    from besskge.bess import AllScoresBESS

    c = load_query_case("allscores", case)
    m = c["meta"]
    n, bps = m["n_shard"], m["bps"]
    from besskge.sharding import Sharding

    fn = make_scorer(c["scorer"], m["norm"], True, m["n_rel"], m["d"], c["table"], c["rel"], torch.device("cpu"),
                     sharding=Sharding.create(m["n_entity"], n, seed=1234))
    model = AllScoresBESS(candidate_sampler(c), fn, window_size=m["window"])
    runner = runtime.inference_model(model, runtime.Options(device_iterations=bps), device=dev)
    known = "tail" if c["scheme"] == "h" else "head"
    inp = {k: c["batch"][k].flatten(end_dim=1) for k in ("relation", known)}
    want = c["outs"]["scores"]  # [bps, n_step, n, shard_bs, n * ws]
    assert want.shape[1] == model.n_step
    for step in range(model.n_step):
        got = runner(step=torch.full((bps * n, 1), step, dtype=torch.int32), **inp)
        torch.testing.assert_close(got.float().cpu().reshape(bps, n, *want.shape[3:]), want[:, step], rtol=1e-4, atol=1e-4)


# --------------------------------------- pipeline vs the unsharded CPU oracle --
@pytest.mark.parametrize("scheme", ["h", "t"])
@pytest.mark.parametrize("filtered", [False, True])
def test_all_scores_pipeline_vs_oracle(dev, scheme, filtered):
    """Mirror of reference tests/test_pipeline.py:42-206 (ComplEx, 4 shards)."""
    from besskge.batch_sampler import RigidShardedBatchSampler
    from besskge.dataset import KGDataset
    from besskge.metric import Evaluation
    from besskge.negative_sampler import PlaceholderNegativeSampler
    from besskge.pipeline import AllScoresPipeline
    from besskge.scoring import ComplEx
    from besskge.sharding import PartitionedTripleSet, Sharding

    seed, n_entity, n_rel, n_shard, n_triple, d = 1234, 5000, 50, 4, 640, 64
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    ent = torch.randn(n_shard, sharding.max_entity_per_shard, 2 * d)
    rel = torch.randn(n_rel, 2 * d)
    triples = np.stack([rng.integers(n_entity, size=n_triple), rng.integers(n_rel, size=n_triple),
                        rng.integers(n_entity, size=n_triple)], axis=1)
    extra = np.stack([rng.integers(n_entity, size=3000), rng.integers(n_rel, size=3000),
                      rng.integers(n_entity, size=3000)], axis=1)
    extra[:600, :2] = triples[rng.integers(n_triple, size=600), :2]  # some share (h, r) with test queries
    extra[600:1200, 1:] = triples[rng.integers(n_triple, size=600), 1:]
    ds = KGDataset(n_entity=n_entity, n_relation_type=n_rel, triples={"test": triples},
                   original_triple_ids={"test": np.arange(n_triple)})
    mode = "h_shard" if scheme == "t" else "t_shard"
    pts = PartitionedTripleSet.create_from_dataset(ds, "test", sharding, partition_mode=mode)
    fn = ComplEx(True, sharding, n_rel, d, ent, rel)
    bs = RigidShardedBatchSampler(pts, PlaceholderNegativeSampler(scheme), shard_bs=80, batches_per_step=2,
                                  seed=seed, return_triple_idx=True)
    ev = Evaluation(["mrr", "hits@10"], mode="average", reduction="sum", return_ranks=True)
    cand_ents = np.sort(rng.choice(n_entity, size=4000, replace=False)) if filtered else None
    pipe = AllScoresPipeline(bs, scheme, fn, evaluation=ev, filter_triples=[extra] if filtered else None,
                             candidate_ents=cand_ents, return_scores=True, return_topk=True, k=10, window_size=500,
                             device=dev)
    out = pipe()
    order = pts.triple_sort_idx[out["triple_idx"].numpy()]
    tr = triples[order]
    flat = ent[sharding.entity_to_shard, sharding.entity_to_idx]
    if scheme == "t":
        full = kge.score_candidates("ComplEx", 0, True, "t", flat[tr[:, 0]], rel, T(tr[:, 1]), flat[None])
        truth = tr[:, 2]
    else:
        full = kge.score_candidates("ComplEx", 0, True, "h", flat[tr[:, 2]], rel, T(tr[:, 1]), flat[None])
        truth = tr[:, 0]
    want = full.clone()
    rows = torch.arange(len(tr))
    true_sc = want[rows, T(truth)].clone()
    if filtered:
        off = np.setdiff1d(np.arange(n_entity), cand_ents)
        want[:, T(off)] = -torch.inf
        true_sc = want[rows, T(truth)].clone()
        col, other = (0, 2) if scheme == "t" else (2, 0)
        for i, (a, r_) in enumerate(zip(tr[:, col], tr[:, 1])):
            hit = extra[(extra[:, col] == a) & (extra[:, 1] == r_)][:, other]
            want[i, T(hit)] = -torch.inf
    # reference quirk kept: Evaluation.ranks_from_scores replaces non-finite positive scores *in place*
    # (metric.py:152, `nan_to_num_(-inf)` also maps -inf to the lowest finite float), and the pipeline
    # writes that tensor back into the returned scores (pipeline.py:266-271)
    want[rows, T(truth)] = torch.nan_to_num(true_sc, neginf=torch.finfo(torch.float32).min)
    torch.testing.assert_close(out["scores"], want, rtol=1e-4, atol=1e-3)
    masked = want.clone()
    masked[rows, T(truth)] = -torch.inf
    ts = torch.nan_to_num(true_sc, neginf=torch.finfo(torch.float32).min)[:, None]  # same quirk
    gt = (masked > ts).sum(-1).float()
    ge = (masked >= ts).sum(-1).float()
    want_rank = 1 + 0.5 * (gt + ge)
    assert float((out["ranks"] != want_rank).float().mean()) < 0.01  # rank flips only at numerical ties
    torch.testing.assert_close(out["metrics"]["mrr"], (1 / want_rank).sum(), rtol=1e-3, atol=1e-3)
    top = torch.topk(want, 10, dim=-1)
    finite = torch.isfinite(top.values).all(-1)
    agree = (out["topk_global_id"][finite] == top.indices[finite]).float().mean()
    assert float(agree) > 0.99
    assert set(out["metrics_avg"]) == {"mrr", "hits@10"} and 0 < float(out["metrics_avg"]["mrr"]) <= 1


# ----------------------------------------------------------------- pruned score tiles (next-1 as the survey wrote it)
@pytest.mark.parametrize("scorer,dtype,W", [("ComplEx", torch.float32, 128), ("DistMult", torch.float16, 96),
                                            ("TransE", torch.float16, 64), ("TransE", torch.float32, 64),
                                            ("RotatE", torch.float16, 128)])
def test_pruned_scoring_writes_exactly_the_blocks_that_matter(dev, scorer, dtype, W):
    """bess_neg_score_shared_fwd_pruned: a (row, 64-column block) is flagged iff one of its scores beats the row's
    threshold (kernels with a pruning epilogue) - never unflagged when one does (all kernels); flagged blocks hold
    the scores of the plain kernel; bess_topk_update_flagged over them equals the dense update."""
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(W)
    nq, n_ent = 300, 40_000
    table = (torch.randn(n_ent, W, generator=gen) * 0.3).to(dtype).to(dev)
    q = (torch.randn(nq, W, generator=gen) * 0.3).to(dev)
    code = dict(ComplEx=nat.COMPLEX, DistMult=nat.DISTMULT, TransE=nat.TRANSE, RotatE=nat.ROTATE)[scorer]
    d = nat.make_desc(code, 1 if scorer in ("TransE", "RotatE") else 0, table, W // 2 if scorer == "RotatE" else W)
    src = nat.RowSource(table[: n_ent - 37])  # a ragged last block
    n = len(src)
    dense = nat.neg_score_shared_fwd(d, q, src, pad_ld=True)
    kk = 11
    thr = dense.topk(kk, dim=1).values[:, -1].contiguous() - 1e-3  # a few candidates per row beat it
    thr[::7] = -1e30  # rows without a threshold yet: everything is written
    thr[3::7] = 1e30  # nothing can enter
    sc, flags = nat.neg_score_shared_fwd_pruned(d, q, src, thr)
    torch.cuda.synchronize()
    nb = (n + 63) // 64
    pad = torch.full((nq, nb * 64), -float("inf"), device=dev)
    pad[:, :n] = dense[:, :n]
    hit = (pad.reshape(nq, nb, 64) > thr[:, None, None]).any(dim=2)
    got = flags[:, :nb].bool()
    assert bool((got | ~hit).all()), "a block holding a candidate was not flagged"
    prunes = (scorer in ("ComplEx", "DistMult")) or (dtype == torch.float16 and W % 32 == 0)
    if prunes:
        # (the two kernels round differently only in the last bits: a score within 1e-5 of the threshold may fall on
        # either side)
        near = ((pad.reshape(nq, nb, 64) - thr[:, None, None]).abs() < 1e-4).any(dim=2)
        assert bool(((got == hit) | near).all())
        assert float(got.float().mean()) < 0.6
    else:
        assert bool(got.all())
    cols = torch.arange(nb * 64, device=dev) // 64
    written = got[:, cols][:, :n]
    torch.testing.assert_close(sc[:, :n][written], dense[:, :n][written], rtol=1e-5, atol=1e-5)
    # the flagged top-k update equals the dense one
    bs0 = torch.full((nq, kk), -50000.0, device=dev)
    bs0[:, :] = thr[:, None].clamp(min=-50000.0, max=50000.0)
    bi0 = torch.full((nq, kk), n_ent, dtype=torch.int32, device=dev)
    want_s, want_i = bs0.clone(), bi0.clone()
    nat.topk_update(dense[:, :n] if False else dense, want_s, want_i, id_base=5)
    got_s, got_i = bs0.clone(), bi0.clone()
    nat.topk_update(sc, got_s, got_i, id_base=5, flags=flags)
    torch.cuda.synchronize()
    ok = thr < 1e29
    torch.testing.assert_close(got_s[ok], want_s[ok], rtol=1e-5, atol=1e-5)
    same = (got_i == want_i) | ((got_s - want_s).abs() < 1e-4)
    assert bool(same[ok].all())


@pytest.mark.parametrize("n_shard", [1, 2])
@pytest.mark.parametrize("scorer", ["ComplEx", "TransE16"])
def test_topk_over_all_entities_with_and_without_pruning(dev, scorer, n_shard):
    """TopKQueryBessKGE over every entity: pruned score tiles (geometric tile schedule) give the ids and scores
    of the unpruned pass."""
    from besskge import runtime, scoring
    from besskge.bess import TopKQueryBessKGE
    from besskge.negative_sampler import PlaceholderNegativeSampler
    from besskge.sharding import Sharding

    torch.manual_seed(0)
    n_entity, n_rel, d = 30_011, 7, 64
    sharding = Sharding.create(n_entity, n_shard, seed=3)
    if scorer == "ComplEx":
        fn = scoring.ComplEx(True, sharding, n_rel, d, device=dev)
        fn.entity_embedding.data.normal_(0, 0.5)
    else:
        fn = scoring.TransE(True, 1, sharding, n_rel, d, device=dev, dtype=torch.float16)
        fn.entity_embedding.data.normal_(0, 0.5)
    rng = np.random.default_rng(1)
    bsz = 96
    M = int(sharding.shard_counts.min())
    batch = dict(relation=torch.from_numpy(rng.integers(n_rel, size=(n_shard, bsz)).astype(np.int32)),
                 head=torch.from_numpy(rng.integers(M, size=(n_shard, bsz)).astype(np.int32)))
    outs = {}
    for prune in (False, True):
        model = TopKQueryBessKGE(k=10, candidate_sampler=PlaceholderNegativeSampler("t"), score_fn=fn,
                                 return_scores=True)
        model.prune_scores = prune
        model.first_tile = 512
        runner = runtime.inference_model(model, runtime.Options(device_iterations=1), device=dev)
        outs[prune] = runner(**batch)
    torch.testing.assert_close(outs[True]["topk_scores"].float(), outs[False]["topk_scores"].float(), rtol=1e-4, atol=1e-4)
    close = (outs[True]["topk_scores"].float() - outs[False]["topk_scores"].float()).abs() < 1e-4
    same = (outs[True]["topk_global_id"] == outs[False]["topk_global_id"]) | close
    assert float(same.float().mean()) > 0.999


@pytest.mark.parametrize("k", [127, 200, 500])
def test_topk_lists_longer_than_the_kernel_keeps(dev, k):
    """The reference's `torch.topk` takes any k (bess.py:807-814): TopKQueryBessKGE with k + 1 > 128 merges its
    lists with torch.topk on the device; ids and scores equal a direct top-k over all entities."""
    from besskge import runtime, scoring
    from besskge.bess import TopKQueryBessKGE
    from besskge.negative_sampler import PlaceholderNegativeSampler
    from besskge.sharding import Sharding

    torch.manual_seed(1)
    n_entity, n_rel, d, n_shard, bsz = 5003, 5, 32, 2, 40
    sharding = Sharding.create(n_entity, n_shard, seed=3)
    fn = scoring.DistMult(True, sharding, n_rel, d, device=dev)
    fn.entity_embedding.data.normal_(0, 1.0)
    fn.relation_embedding.data.normal_(0, 1.0)
    rng = np.random.default_rng(1)
    M = int(sharding.shard_counts.min())
    head = rng.integers(M, size=(n_shard, bsz)).astype(np.int32)
    rel = rng.integers(n_rel, size=(n_shard, bsz)).astype(np.int32)
    model = TopKQueryBessKGE(k=k, candidate_sampler=PlaceholderNegativeSampler("t"), score_fn=fn, return_scores=True)
    model.score_tile_bytes = 1 << 18  # several tiles
    runner = runtime.inference_model(model, runtime.Options(device_iterations=1), device=dev)
    out = runner(relation=torch.from_numpy(rel), head=torch.from_numpy(head))
    # direct: every entity's score against the queries, unsharded
    table = fn.entity_embedding.detach().float().cpu()  # [n, M, W]
    relt = fn.relation_embedding.detach().float().cpu()
    glob = torch.zeros(n_entity, d)
    s2e = torch.from_numpy(np.asarray(sharding.shard_and_idx_to_entity))
    for sh in range(n_shard):
        cnt = int(sharding.shard_counts[sh])
        glob[s2e[sh, :cnt].long()] = table[sh, :cnt]
    for sh in range(n_shard):
        q = table[sh][torch.from_numpy(head[sh]).long()] * relt[torch.from_numpy(rel[sh]).long()]
        want_s, want_i = torch.topk(q @ glob.T, k, dim=1)
        got_s = out["topk_scores"][sh * bsz:(sh + 1) * bsz].float().cpu()
        got_i = out["topk_global_id"][sh * bsz:(sh + 1) * bsz].cpu().long()
        torch.testing.assert_close(got_s, want_s, rtol=1e-4, atol=1e-4)
        assert float(((got_i == want_i) | ((got_s - want_s).abs() < 1e-4)).float().mean()) > 0.999


# ------------------------------------------------- ranks in the scoring epilogue (next-2 as the survey wrote it)
@pytest.mark.parametrize("scorer,dtype,W,indexed", [("ComplEx", torch.float32, 128, False),
                                                    ("DistMult", torch.float16, 96, True),
                                                    ("TransE", torch.float16, 64, False),
                                                    ("TransE", torch.float16, 64, True),
                                                    ("TransE", torch.float32, 64, False),
                                                    ("RotatE", torch.float16, 128, True)])
def test_counts_in_the_scoring_epilogue_equal_counts_of_the_stored_scores(dev, scorer, dtype, W, indexed):
    """bess_neg_score_shared_fwd_counts: per row the number of candidates above / equal to the row's threshold,
    the excluded position left out - exactly what counting the stored score matrix of the same kernel gives
    (matrix-core product, packed L1 kernel, and the tile + count fallback)."""
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(W + int(indexed))
    nq, n_ent = 300, 40_000
    table = (torch.randn(n_ent, W, generator=gen) * 0.3).to(dtype).to(dev)
    q = (torch.randn(nq, W, generator=gen) * 0.3).to(dev)
    code = dict(ComplEx=nat.COMPLEX, DistMult=nat.DISTMULT, TransE=nat.TRANSE, RotatE=nat.ROTATE)[scorer]
    d = nat.make_desc(code, 1 if scorer in ("TransE", "RotatE") else 0, table, W // 2 if scorer == "RotatE" else W)
    n_cand = n_ent - 37  # a ragged last block
    if indexed:
        idx = torch.randperm(n_ent, generator=gen)[:n_cand].to(torch.int32).to(dev)
        src = nat.RowSource(table, idx)
    else:
        src = nat.RowSource(table[:n_cand])
    sc = nat.neg_score_shared_fwd(d, q, src)
    # thresholds: the score of the excluded candidate (ties with duplicates of it below), or of none of them
    excl = torch.randint(0, n_cand, (nq,), generator=gen).to(torch.int32)
    excl[::7] = -1
    excl[5], excl[6] = 0, n_cand - 1  # first and last column
    excl = excl.to(dev)
    rows = torch.arange(nq, device=dev)
    thr = torch.where(excl >= 0, sc[rows, excl.clamp(min=0).long()], sc[rows, 17])
    thr[3] = float("inf")
    thr[4] = -float("inf")
    counts = nat.neg_score_shared_counts(d, q, src, thr.contiguous(), excl)
    keep = torch.ones_like(sc, dtype=torch.bool)
    keep[rows[excl >= 0], excl[excl >= 0].long()] = False
    want_gt = ((sc > thr[:, None]) & keep).sum(-1)
    want_eq = ((sc == thr[:, None]) & keep).sum(-1)
    assert torch.equal(counts[:, 0].long(), want_gt)
    assert torch.equal(counts[:, 1].long(), want_eq)
    assert int(want_eq[excl < 0].min()) >= 1  # (rows without an exclusion tie with their column 17)
    # accumulation over windows: two halves of the candidates into the same counters
    half = (n_cand // 2) // 64 * 64 + 13
    if indexed:
        a, b = nat.RowSource(table, idx[:half].contiguous()), nat.RowSource(table, idx[half:].contiguous())
    else:
        a, b = nat.RowSource(table[:half]), nat.RowSource(table[half:n_cand])
    c2 = nat.neg_score_shared_counts(d, q, a, thr, torch.where(excl < half, excl, torch.full_like(excl, -1)))
    ex_b = torch.where(excl >= half, excl - half, torch.full_like(excl, -1))
    c2 = nat.neg_score_shared_counts(d, q, b, thr, ex_b, counts=c2)
    assert torch.equal(c2, counts)
    # a half-precision model ranks fp16 scores: many ties
    sc16, thr16 = sc.half().float(), thr.half().float().contiguous()
    c16 = nat.neg_score_shared_counts(d, q, src, thr16, excl, round_f16=True)
    assert torch.equal(c16[:, 0].long(), ((sc16 > thr16[:, None]) & keep).sum(-1))
    assert torch.equal(c16[:, 1].long(), ((sc16 == thr16[:, None]) & keep).sum(-1))
    assert int(c16[:, 1].sum()) > int(counts[:, 1].sum())


@pytest.mark.parametrize("scorer,dtype,n_entity,n_shard,shard_bs", [
    ("ComplEx", torch.float32, 60_000, 2, 160),  # split-fp16 product with the counting epilogue
    ("TransE", torch.float16, 20_000, 2, 80),    # packed L1 kernel with the counting epilogue
    ("RotatE", torch.float32, 6_000, 3, 40),     # score tiles + count kernel
    ("DistMult", torch.float32, 5_000, 4, 80),   # small shards: the fp32 product, tiles + count kernel
])
@pytest.mark.parametrize("scheme,filtered,mode", [("t", False, "average"), ("h", True, "optimistic"),
                                                  ("t", True, "pessimistic")])
def test_pipeline_ranks_by_counting_equal_ranks_from_the_score_matrix(dev, scorer, dtype, n_entity, n_shard, shard_bs,
                                                                      scheme, filtered, mode):
    """AllScoresPipeline with only metrics / ranks asked for never assembles the score matrix
    (`AllScoresBESS.rank_counts_replicas`): same ranks as the matrix path and as the CPU oracle."""
    from besskge.batch_sampler import RigidShardedBatchSampler
    from besskge.dataset import KGDataset
    from besskge.metric import Evaluation
    from besskge.negative_sampler import PlaceholderNegativeSampler
    from besskge.pipeline import AllScoresPipeline
    from besskge.sharding import PartitionedTripleSet, Sharding

    seed, n_rel, d = 99, 30, 64
    n_triple = 3 * n_shard * shard_bs - 17  # a padded last batch
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    ew, rw = kge.entity_width(scorer, d), kge.relation_width(scorer, d)
    ent = torch.randn(n_shard, sharding.max_entity_per_shard, ew) * 0.3
    rel = torch.randn(n_rel, rw) * 0.3
    if dtype == torch.float16:
        ent, rel = ent.half().float(), rel.half().float()
    triples = np.stack([rng.integers(n_entity, size=n_triple), rng.integers(n_rel, size=n_triple),
                        rng.integers(n_entity, size=n_triple)], axis=1)
    extra = np.stack([rng.integers(n_entity, size=4000), rng.integers(n_rel, size=4000),
                      rng.integers(n_entity, size=4000)], axis=1)
    extra[:1500, :2] = triples[rng.integers(n_triple, size=1500), :2]  # share (h, r) with test queries
    extra[1500:3000, 1:] = triples[rng.integers(n_triple, size=1500), 1:]  # share (r, t)
    extra[3000:3200] = triples[:200]  # the test triples themselves (filtered and true completion at once)
    extra[3200:3400] = extra[:200]    # duplicates in the filter set
    ds = KGDataset(n_entity=n_entity, n_relation_type=n_rel, triples={"test": triples},
                   original_triple_ids={"test": np.arange(n_triple)})
    pts = PartitionedTripleSet.create_from_dataset(ds, "test", sharding,
                                                   partition_mode="h_shard" if scheme == "t" else "t_shard")
    p = 1 if scorer in ("TransE", "RotatE") else 0
    fn = make_scorer(scorer, p, True, n_rel, d, ent, rel, dev, dtype=dtype, sharding=sharding)
    bs = RigidShardedBatchSampler(pts, PlaceholderNegativeSampler(scheme), shard_bs=shard_bs, batches_per_step=2,
                                  seed=seed, return_triple_idx=True)
    ev = Evaluation(["mrr", "hits@10"], mode=mode, reduction="sum", return_ranks=True)
    # (the matrix path scores window by window: give it windows that take the same kernel as the all-entity pass -
    # the split-fp16 product needs 256 output tiles - so that the two paths can be compared to the last bit)
    window = sharding.max_entity_per_shard if scorer == "ComplEx" else 1000
    # (filtered runs also rank against a subset of the entities: `candidate_ents`)
    cand_ents = np.sort(rng.choice(n_entity, size=int(0.8 * n_entity), replace=False)) if filtered else None
    kw = dict(evaluation=ev, filter_triples=[extra] if filtered else None, candidate_ents=cand_ents, window_size=window,
              device=dev)
    fused = AllScoresPipeline(bs, scheme, fn, **kw)
    assert fused.fused_ranks
    plain = AllScoresPipeline(bs, scheme, fn, fused_ranks=False, **kw)
    assert not plain.fused_ranks
    a, b = fused(), plain()
    assert torch.equal(a["triple_idx"], b["triple_idx"]) and len(a["ranks"]) == n_triple
    # (the positives' and the filtered completions' scores come from the matrix kernel's arithmetic on both paths:
    # bess_neg_score_shared_fwd_pairs - the same ranks to the last bit)
    assert torch.equal(a["ranks"], b["ranks"])
    torch.testing.assert_close(a["metrics"]["mrr"], b["metrics"]["mrr"], rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(a["metrics"]["hits@10"], b["metrics"]["hits@10"], rtol=0, atol=2)
    # and the unsharded oracle
    order = pts.triple_sort_idx[a["triple_idx"].numpy()]
    tr = triples[order]
    flat = ent[sharding.entity_to_shard, sharding.entity_to_idx]
    known, truth = (tr[:, 0], tr[:, 2]) if scheme == "t" else (tr[:, 2], tr[:, 0])
    import contextlib

    half = dtype == torch.float16  # (the fp16 mode: query rounded to fp16, fp16 scores ranked - ties included)
    with (kge.half_queries() if half else contextlib.nullcontext()):
        full = kge.score_candidates(scorer, p, True, scheme, flat[known], rel, T(tr[:, 1]), flat[None])
    if half:
        full = full.half().float()
    rows = torch.arange(len(tr))
    if filtered:
        full[:, T(np.setdiff1d(np.arange(n_entity), cand_ents))] = -torch.inf
    # (reference quirk: a -inf positive - a true completion outside the candidates - becomes the lowest finite float)
    true_sc = torch.nan_to_num(full[rows, T(truth)].clone(), neginf=torch.finfo(torch.float32).min)
    if filtered:
        col, other = (0, 2) if scheme == "t" else (2, 0)
        for i, (e_, r_) in enumerate(zip(tr[:, col], tr[:, 1])):
            full[i, T(extra[(extra[:, col] == e_) & (extra[:, 1] == r_)][:, other])] = -torch.inf
    full[rows, T(truth)] = -torch.inf
    gt, ge = (full > true_sc[:, None]).sum(-1).float(), (full >= true_sc[:, None]).sum(-1).float()
    want = 1 + dict(optimistic=gt, pessimistic=ge, average=0.5 * (gt + ge))[mode]
    assert float((a["ranks"] != want).float().mean()) < (0.05 if half else 0.03)  # (rank flips at numerical ties: fp16 rounding boundaries; RotatE's sin / cos)


@pytest.mark.parametrize("n_shard", [2, 3])
def test_counting_falls_back_when_an_operand_leaves_the_fp16_range(dev, n_shard):
    """the split-fp16 product poisons a shard's counts with INT32_MIN when an operand is
    outside the fp16 range.  A query row goes into EVERY shard's product, so all n shards are poisoned at once and
    n * INT32_MIN wraps to 0 for even n (and INT32_MIN - 1 wraps in the filter correction): the sentinel used to
    vanish in the sum and every rank came out as 1.  Now the flag travels next to the counts (`out_of_range`) and
    the pipeline takes the matrix path for that batch: same ranks as the pipeline that never counts."""
    from besskge.batch_sampler import RigidShardedBatchSampler
    from besskge.dataset import KGDataset
    from besskge.metric import Evaluation
    from besskge.negative_sampler import PlaceholderNegativeSampler
    from besskge.pipeline import AllScoresPipeline
    from besskge.sharding import PartitionedTripleSet, Sharding

    seed, n_rel, d, n_entity, shard_bs = 5, 11, 64, 60_000, 160
    n_triple = 2 * n_shard * shard_bs
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    ent = torch.randn(n_shard, sharding.max_entity_per_shard, 2 * d) * 0.3
    rel = torch.randn(n_rel, 2 * d) * 0.3
    triples = np.stack([rng.integers(n_entity, size=n_triple), rng.integers(n_rel, size=n_triple),
                        rng.integers(n_entity, size=n_triple)], axis=1)
    h0 = int(triples[7, 0])  # the head of one query: its row makes that query's row of the product huge
    ent[sharding.entity_to_shard[h0], sharding.entity_to_idx[h0], :4] = 3.0e5
    extra = np.stack([rng.integers(n_entity, size=2000), rng.integers(n_rel, size=2000),
                      rng.integers(n_entity, size=2000)], axis=1)
    extra[:1500, :2] = triples[rng.integers(n_triple, size=1500), :2]
    ds = KGDataset(n_entity=n_entity, n_relation_type=n_rel, triples={"test": triples},
                   original_triple_ids={"test": np.arange(n_triple)})
    pts = PartitionedTripleSet.create_from_dataset(ds, "test", sharding, partition_mode="h_shard")
    fn = make_scorer("ComplEx", 0, True, n_rel, d, ent, rel, dev, sharding=sharding)
    bs = RigidShardedBatchSampler(pts, PlaceholderNegativeSampler("t"), shard_bs=shard_bs, batches_per_step=1,
                                  seed=seed, return_triple_idx=True)
    ev = Evaluation(["mrr"], mode="average", reduction="sum", return_ranks=True)
    kw = dict(evaluation=ev, filter_triples=[extra], window_size=sharding.max_entity_per_shard, device=dev)
    fused = AllScoresPipeline(bs, "t", fn, **kw)
    assert fused.fused_ranks
    seen = []
    inner = fused._ranks_by_counting
    fused._ranks_by_counting = lambda *a, **k: (seen.append(inner(*a, **k)), seen[-1])[1]
    a = fused()
    b = AllScoresPipeline(bs, "t", fn, fused_ranks=False, **kw)()
    assert any(r is None for r in seen), "the out-of-range flag was lost: no batch fell back to the score matrix"
    assert torch.equal(a["ranks"], b["ranks"])
    assert float((a["ranks"] == 1).float().mean()) < 0.5  # (the lost sentinel made every rank 1)
    # and the module's own outputs say which queries were flagged
    mod = fused.bess_module
    batch = next(iter(fused.dl))
    rows = batch["head"].flatten(end_dim=1)
    res = fused.runner(step=torch.zeros((rows.shape[0], 1), dtype=torch.int32), head=rows,
                       relation=batch["relation"].flatten(end_dim=1),
                       rank_truth=batch["tail"].flatten(end_dim=1).to(torch.int32))
    assert bool(res["out_of_range"].any()) and bool((res["counts"][res["out_of_range"]] == -1).all())
    assert not bool((res["counts"][~res["out_of_range"]] < 0).any())
    del mod


@pytest.mark.parametrize("scorer,dtype,W,n_cand", [("ComplEx", torch.float32, 128, 40_000), ("DistMult", torch.float16, 96, 40_000),
                                                   ("TransE", torch.float16, 64, 40_000), ("TransE", torch.float32, 64, 3_000),
                                                   ("RotatE", torch.float16, 128, 2_000), ("ComplEx", torch.float32, 64, 900)])
def test_pair_scores_equal_the_matrix_kernels_elements(dev, scorer, dtype, W, n_cand):
    """bess_neg_score_shared_fwd_pairs: score(query i, candidate c_i) bit-for-bit the element [i, c_i] of the
    all-entity pass (every kernel family: split-fp16 product, packed L1, fp32 product, generic tiles)."""
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(W)
    nq = 300
    table = (torch.randn(n_cand + 50, W, generator=gen) * 0.3).to(dtype).to(dev)
    q = (torch.randn(nq, W, generator=gen) * 0.3).to(dev)
    code = dict(ComplEx=nat.COMPLEX, DistMult=nat.DISTMULT, TransE=nat.TRANSE, RotatE=nat.ROTATE)[scorer]
    d = nat.make_desc(code, 1 if scorer in ("TransE", "RotatE") else 0, table, W // 2 if scorer == "RotatE" else W)
    sc = nat.neg_score_shared_fwd(d, q, nat.RowSource(table[:n_cand]))
    n_pair = 2500  # (more than two blocks of 1024, a ragged last one)
    rows = torch.randint(0, nq, (n_pair,), generator=gen).to(dev)
    cols = torch.randint(0, n_cand, (n_pair,), generator=gen).to(torch.int32).to(dev)
    got = nat.neg_score_shared_pairs(d, q[rows].contiguous(), nat.RowSource(table, cols), nq, n_cand)
    assert torch.equal(got, sc[rows, cols.long()])
