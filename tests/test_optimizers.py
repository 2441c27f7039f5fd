"""Row-sparse optimisers (K10) and the generic coalescing step (K9) on the GPU.

Kernel level: `bess_segment_sum_rows` + `bess_apply_segments_opt` against plain
torch formulas (lazy semantics: only touched rows and their state move) over
several steps with duplicate rows.  End to end: a BessKGE training step with
Adagrad / Adam / SGD-momentum equals oracle autograd + torch.optim on the dense
tables (one step from zero state, where dense and lazy semantics coincide)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import kge  # noqa: E402

from test_oracle import load_bess_case  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def lazy_reference(kind, p, rows, g, state, hp, step):
    """One lazy step on CPU.  rows: unique touched rows, g: summed gradients."""
    p = p.clone()
    if kind == "sgd":
        gg = g + hp["weight_decay"] * p[rows]
        if hp["momentum"]:
            state[0][rows] = hp["momentum"] * state[0][rows] + gg
            gg = state[0][rows]
        p[rows] -= hp["lr"] * gg
    elif kind == "adagrad":
        gg = g + hp["weight_decay"] * p[rows]
        state[0][rows] += gg * gg
        p[rows] -= hp["lr"] * gg / (state[0][rows].sqrt() + hp["eps"])
    else:
        p[rows] -= hp["lr"] * hp["weight_decay"] * p[rows]
        state[0][rows] = hp["beta1"] * state[0][rows] + (1 - hp["beta1"]) * g
        state[1][rows] = hp["beta2"] * state[1][rows] + (1 - hp["beta2"]) * g * g
        b1, b2 = 1 - hp["beta1"] ** step, 1 - hp["beta2"] ** step
        p[rows] -= hp["lr"] * np.sqrt(b2) / b1 * state[0][rows] / (state[1][rows].sqrt() + hp["eps"])
    return p


@pytest.mark.parametrize("kind", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_apply_segments_opt_matches_lazy_formulas(dev, kind, dtype):
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(0)
    M, W = 200, 48
    table = torch.randn(M, W, generator=gen).to(dtype)
    p_ref = table.float()
    p_dev = table.to(dev)
    hp = dict(lr=0.05, momentum=0.9, weight_decay=0.01, eps=1e-8 if kind == "adam" else 1e-10, beta1=0.9, beta2=0.99)
    s_ref = [torch.zeros(M, W), torch.zeros(M, W)]
    s_dev = [torch.zeros(M, W, device=dev), torch.zeros(M, W, device=dev)]
    o = nat.OptDesc()
    o.kind = dict(sgd=nat.OPT_SGD, adagrad=nat.OPT_ADAGRAD, adam=nat.OPT_ADAM)[kind]
    o.lr, o.momentum, o.weight_decay, o.eps, o.beta1, o.beta2 = hp["lr"], hp["momentum"], hp["weight_decay"], hp["eps"], hp["beta1"], hp["beta2"]
    for step in range(1, 4):
        idx = torch.randint(M, (300,), generator=gen, dtype=torch.int32)
        src = torch.randn(300, W, generator=gen)
        seg = nat.SegmentIndex(idx.to(dev), M)
        gseg = nat.segment_sum_rows(src.to(dev), seg)
        n = int(seg.n_seg.item())
        rows = torch.unique(idx.long())
        want_g = torch.zeros(M, W, dtype=torch.float64).index_add_(0, idx.long(), src.double())[rows].float()
        torch.testing.assert_close(gseg[:n].cpu(), want_g, rtol=1e-5, atol=1e-5)
        assert torch.equal(seg.seg_rows[:n].cpu().long(), rows)
        o.step = step
        nat.apply_segments_opt(o, p_dev, seg, gseg, s_dev[0], s_dev[1] if kind == "adam" else None)
        p_ref = lazy_reference(kind, p_ref, rows, want_g, s_ref, hp, step)
        if dtype == torch.float16:
            p_ref = p_ref.half().float()  # the shard is stored in fp16 after every step
        tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-3, atol=2e-3)
        torch.testing.assert_close(p_dev.float().cpu(), p_ref, **tol)
        torch.testing.assert_close(s_dev[0].cpu(), s_ref[0], rtol=1e-4, atol=1e-5)
    untouched = torch.ones(M, dtype=torch.bool)
    # rows never touched in any step keep their value and zero state
    # (all rows are likely touched over 3 x 300 draws on 200 rows; check state invariants instead)
    assert torch.isfinite(p_dev.float()).all()
    del untouched


@pytest.mark.parametrize("opt_name", ["adagrad", "adam", "sgdm"])
@pytest.mark.parametrize("case", ["tr_EM_ComplEx0_t_flat_n1", "tr_EM_TransE1_h_pt_n1", "tr_EM_RotatE2_ht_flat_n2",
                                  "tr_EM_aug_t_flat_n4"])
def test_training_step_with_optimizers(dev, opt_name, case):
    from besskge import runtime
    from test_hip_parity import build_model

    c = load_bess_case(case)
    n = c["meta"]["n_shard"]
    keys = ("head", "relation", "tail", "negative", "negative_mask")
    batch = {k: c["batch"][k][0] for k in keys if k in c["batch"]}
    model = build_model(c, dev)
    if opt_name == "adagrad":
        opt, topt = runtime.Adagrad(lr=0.1, eps=1e-10), lambda ps: torch.optim.Adagrad(ps, lr=0.1, eps=1e-10)
    elif opt_name == "adam":
        opt = runtime.Adam(lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8)
        topt = lambda ps: torch.optim.Adam(ps, lr=0.01, betas=(0.9, 0.999), eps=1e-8)  # noqa: E731
    else:
        opt, topt = runtime.SGD(lr=0.05, momentum=0.9), lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9)
    runner = runtime.training_model(model, runtime.Options(device_iterations=1), opt, device=dev)
    res = runner(**batch)
    # oracle: autograd on the dense tables + the torch optimiser, one step from zero state
    t0 = c["table"].clone().requires_grad_(True)
    r0 = c["rel"].clone().requires_grad_(True)
    want = kge.bess_step(c["spec"], c["model_cls"], t0, r0, batch, c["loss"])
    torch.stack(want["loss"]).sum().backward()
    tor = topt([t0, r0])
    tor.step()
    np.testing.assert_allclose(res["loss"].float().cpu().numpy().reshape(n),
                               torch.stack(want["loss"]).detach().numpy(), rtol=1e-4, atol=1e-4)
    # Adam / Adagrad normalise by |g|: entries whose gradient is ~0 amplify rounding noise -> compare where |g| is sane
    got_t = model.score_fn.entity_embedding.detach().float().cpu()
    got_r = model.score_fn.relation_embedding.detach().float().cpu()
    for got, ref, grad, before in ((got_t, t0.detach(), t0.grad, c["table"]), (got_r, r0.detach(), r0.grad, c["rel"])):
        # sign-normalising optimisers turn an analytically cancelling gradient entry
        # (+x - x, exact 0 in dense autograd, ~1e-9 here) into a full +-lr step:
        # compare entries with a solid gradient, and require untouched rows not to move
        solid = grad.abs() > 1e-4
        torch.testing.assert_close(got[solid], ref[solid], rtol=2e-3, atol=5e-5)
        untouched = (grad.reshape(-1, grad.shape[-1]) == 0).all(dim=-1)
        assert torch.equal(got.reshape(-1, grad.shape[-1])[untouched], before.reshape(-1, grad.shape[-1])[untouched])


@pytest.mark.parametrize("kind", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("scorer,p", [("ComplEx", 1), ("TransE", 1), ("RotatE", 2)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_fused_optimizer_step_in_the_segmented_reduction(dev, kind, scorer, p, dtype):
    """bess_neg_pertriple_step_segments (K9 + K10 in one pass, extra contributions merged through
    bess_map_extra_rows, rows only they touch through bess_apply_segments_opt(keep)) over three steps
    equals: gradient rows of the plain segmented reduction + index_add of the extras, then the lazy
    optimiser formulas on the unique touched rows.  One hot row exercises the long-row tier."""
    from besskge import _native as nat

    gen = torch.Generator().manual_seed(3)
    M, d, S, N = 260, 24, 40, 30
    W = 2 * d if scorer in ("ComplEx", "RotatE") else d
    Wr = d if scorer == "RotatE" else W
    table = torch.randn(M, W, generator=gen).to(dtype)
    p_ref, p_dev = table.float(), table.to(dev)
    hp = dict(lr=0.05, momentum=0.9, weight_decay=0.01, eps=1e-8 if kind == "adam" else 1e-10, beta1=0.9, beta2=0.99)
    s_ref = [torch.zeros(M, W), torch.zeros(M, W)]
    s_dev = [torch.zeros(M, W, device=dev), torch.zeros(M, W, device=dev)]
    o = nat.OptDesc()
    o.kind = dict(sgd=nat.OPT_SGD, adagrad=nat.OPT_ADAGRAD, adam=nat.OPT_ADAM)[kind]
    o.lr, o.momentum, o.weight_decay, o.eps, o.beta1, o.beta2 = hp["lr"], hp["momentum"], hp["weight_decay"], hp["eps"], hp["beta1"], hp["beta2"]
    desc = nat.make_desc(dict(TransE=0, RotatE=1, DistMult=2, ComplEx=3)[scorer], p, p_dev, Wr)
    for step in range(1, 4):
        q = torch.randn(S, W, generator=gen).to(dev)
        go = (torch.randn(S, N, generator=gen) * 0.1).to(dev)
        idx = torch.randint(M // 2, (S * N,), generator=gen, dtype=torch.int32)  # negatives: rows < M / 2 only
        idx[torch.rand(S * N, generator=gen) < 0.3] = 9                         # one long row
        xidx = torch.randint(M, (70,), generator=gen, dtype=torch.int32)          # extras: any row, duplicates
        xgrad = torch.randn(70, W, generator=gen)
        seg = nat.SegmentIndex(idx.to(dev), M, width=W)
        assert int(seg.long_segs[0].item()) == 1
        # reference: plain segmented gradient (tested elsewhere) + extras, lazy formulas on the union
        gseg = nat.neg_pertriple_grad_segments(desc, q, p_dev, N, go, seg)
        n = int(seg.n_seg.item())
        total = torch.zeros(M, W, dtype=torch.float64)
        total[seg.seg_rows[:n].cpu().long()] = gseg[:n].cpu().double()
        total.index_add_(0, xidx.long(), xgrad.double())
        rows = torch.unique(torch.cat([idx.long(), xidx.long()]))
        p_ref = lazy_reference(kind, p_ref, rows, total[rows].float(), s_ref, hp, step)
        if dtype == torch.float16:
            p_ref = p_ref.half().float()
        # device: fused step
        xseg = nat.SegmentIndex(xidx.to(dev), M)
        xsum = nat.segment_sum_rows(xgrad.to(dev), xseg)
        xmap, keep = nat.map_extra_rows(seg, xseg)
        nx = int(xseg.n_seg.item())
        orphan = torch.tensor([int(r) not in set(seg.seg_rows[:n].cpu().tolist()) for r in xseg.seg_rows[:nx].cpu()])
        assert torch.equal(keep[:nx].cpu().bool(), orphan) and bool(orphan.any()) and not bool(orphan.all())
        o.step = step
        s2 = s_dev[1] if kind == "adam" else None
        before = p_dev.clone()
        nat.neg_pertriple_step_segments(desc, q, p_dev, N, go, seg, o, s_dev[0], s2, xmap, xsum)
        nat.apply_segments_opt(o, p_dev, xseg, xsum, s_dev[0], s2, keep=keep)
        tol = dict(rtol=1e-4, atol=2e-5) if dtype == torch.float32 else dict(rtol=4e-3, atol=4e-3)
        torch.testing.assert_close(p_dev.float().cpu(), p_ref, **tol)
        if kind != "sgd" or hp["momentum"]:
            torch.testing.assert_close(s_dev[0].cpu(), s_ref[0], rtol=1e-3, atol=1e-5)
        untouched = torch.ones(M, dtype=torch.bool)
        untouched[rows] = False
        assert torch.equal(p_dev.cpu()[untouched], before.cpu()[untouched])  # lazy: untouched rows do not move


@pytest.mark.parametrize("opt_name", ["adam", "sgdm", "adagrad"])
@pytest.mark.parametrize("case", ["tr_EM_TransE1_t_flat_n1", "tr_EM_ComplEx0_h_pt_n1",
                                  # the general coalescing path (no fused optimiser step): two per-triple groups per
                                  # shard, the affine family, BoxE - their unique-row lists are handed on without
                                  # reading the row count back (bess_pad_segments), so they record too
                                  "tr_EM_RotatE2_ht_pt_n2", "tr_EM_PairRE1_h_pt_n1", "tr_EM_BoxE1_ht_pt_n2",
                                  "tr_SM_TransE1_t_pt_n2"])
def test_graph_replay_with_stateful_optimizers(dev, opt_name, case):
    """Options.use_graphs with Adagrad / SGD-momentum / Adam: four recorded-and-replayed steps move the
    tables like four eager steps (Adam's bias correction follows the device-side step count)."""
    from besskge import runtime
    from test_hip_parity import build_model

    c = load_bess_case(case)
    n = c["meta"]["n_shard"]
    keys = ("head", "relation", "tail", "negative", "negative_mask")
    batch = {k: c["batch"][k].flatten(end_dim=1)[:n] for k in keys if k in c["batch"]}

    def make():
        if opt_name == "adam":
            return runtime.Adam(lr=0.01, weight_decay=0.01)
        if opt_name == "sgdm":
            return runtime.SGD(lr=0.05, momentum=0.9)
        return runtime.Adagrad(lr=0.1)

    out = []
    for graphs in (False, True):
        model = build_model(c, dev)
        runner = runtime.training_model(model, runtime.Options(device_iterations=1, use_graphs=graphs), make(),
                                        device=dev)
        losses = [runner(**batch)["loss"].float().cpu().clone() for _ in range(4)]
        out.append((torch.stack(losses), model.score_fn.entity_embedding.detach().float().cpu().clone(),
                    model.score_fn.relation_embedding.detach().float().cpu().clone()))
    (l0, e0, r0), (l1, e1, r1) = out
    assert float(l0[3].sum()) != float(l0[0].sum())  # the steps do change the model
    torch.testing.assert_close(l1, l0, rtol=1e-4, atol=1e-4)
    if n > 1 and opt_name in ("adam", "adagrad"):
        # exchanged rows come back through fp32 atomics: where a gradient entry cancels to ~0 its sign - and with a
        # sign-normalising optimiser a whole +-lr step - depends on their order (tests/test_accumulation.py)
        for a, b in ((e1, e0), (r1, r0)):
            off = (a - b).abs()
            assert float((off > 1e-4).float().mean()) < 0.02 and float(off.max()) <= 4 * 0.1 * 1.01
    else:
        torch.testing.assert_close(e1, e0, rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(r1, r0, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("opt_name", ["adam", "sgdm"])
def test_graph_capture_of_a_second_signature_keeps_training_state(dev, opt_name):
    """A new input signature in the middle of a run (a last, shorter batch) is captured with
    warm-up steps that must leave no trace: tables, accumulated optimiser state and Adam's step
    count afterwards are those of the eager run."""
    from besskge import runtime
    from test_hip_parity import build_model

    c = load_bess_case("tr_EM_TransE1_t_flat_n1")
    keys = ("head", "relation", "tail", "negative")
    full = {k: c["batch"][k].flatten(end_dim=1)[:1] for k in keys}
    short = dict(full, negative=full["negative"][..., : full["negative"].shape[-1] // 2].contiguous())
    assert short["negative"].shape != full["negative"].shape

    def make():
        return runtime.Adam(lr=0.01, weight_decay=0.01) if opt_name == "adam" else runtime.SGD(lr=0.05, momentum=0.9)

    out = []
    for graphs in (False, True):
        model = build_model(c, dev)
        runner = runtime.training_model(model, runtime.Options(device_iterations=1, use_graphs=graphs), make(),
                                        device=dev)
        losses = [runner(**b)["loss"].float().cpu().clone() for b in (full, full, full, short, short, full)]
        out.append((torch.stack(losses), model.score_fn.entity_embedding.detach().float().cpu().clone(),
                    model.score_fn.relation_embedding.detach().float().cpu().clone()))
    (l0, e0, r0), (l1, e1, r1) = out
    torch.testing.assert_close(l1, l0, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(e1, e0, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(r1, r0, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_coalesced_update_of_several_lists(dev, kind, dtype):
    """bess_coalesced_update: three (rows, gradient) lists with duplicates inside and across the lists are
    summed per unique row in fp32 and the row is updated ONCE - for an f16 table the result is
    fp16(row - lr * sum) exactly (a per-contribution atomic would round several times); `sum_only`
    returns the per-row sums; `keep` masks rows out."""
    from besskge import _native as nat

    torch.manual_seed(3)
    M, W = 300, 96
    table = torch.randn(M, W).to(dtype)
    lists = []
    for n in (70, 1, 130):
        idx = torch.randint(0, 40, (n,), dtype=torch.int32)  # heavy duplication
        lists.append((idx, torch.randn(n, W)))
    idx_all = torch.cat([i for i, _ in lists])
    seg = nat.SegmentIndex(idx_all.to(dev), M)
    grads = [g.to(dev) for _, g in lists]
    sums = nat.coalesced_update(None, table.to(dev), seg, grads, sum_only=True)
    n_seg = int(seg.n_seg.item())
    rows = seg.seg_rows[:n_seg].cpu().long()
    want_sum = torch.zeros(M, W, dtype=torch.float64)
    want_sum.index_add_(0, idx_all.long(), torch.cat([g for _, g in lists]).double())
    assert torch.equal(rows, torch.unique(idx_all.long()))
    torch.testing.assert_close(sums[:n_seg].cpu().double(), want_sum[rows], rtol=1e-5, atol=1e-5)

    o = nat.OptDesc()
    hp = dict(lr=0.05, momentum=0.0, weight_decay=0.0, beta1=0.9, beta2=0.999, eps=1e-8)
    o.kind, o.step, o.lr = (nat.OPT_SGD if kind == "sgd" else nat.OPT_ADAM), 1, hp["lr"]
    o.beta1, o.beta2, o.eps = hp["beta1"], hp["beta2"], hp["eps"]
    t_dev = table.to(dev).clone()
    s1 = torch.zeros(M, W, device=dev) if kind == "adam" else None
    s2 = torch.zeros(M, W, device=dev) if kind == "adam" else None
    keep = torch.ones(seg.max_seg, dtype=torch.int32, device=dev)
    keep[0] = 0  # the smallest touched row stays as it is
    nat.coalesced_update(o, t_dev, seg, grads, s1, s2, keep=keep)
    state = [torch.zeros(M, W), torch.zeros(M, W)]
    ref = lazy_reference(kind, table.float(), rows[1:], sums[:n_seg].cpu()[1:], state, hp, 1)
    got = t_dev.cpu()
    if dtype == torch.float16:
        assert torch.equal(got, ref.half())  # rounded once, from the fp32 result
    else:
        torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)
    assert torch.equal(got[rows[0]], table[rows[0]])
    with pytest.raises(ValueError, match="references"):
        nat.coalesced_update(o, t_dev, seg, grads[:2], s1, s2)


@pytest.mark.parametrize("case", ["tr_EM_ComplEx0_h_pt_n1", "tr_EM_TransE1_t_flat_n1"])
@pytest.mark.parametrize("graphs", [False, True])
def test_paged_optimizer_state(dev, case, graphs):
    """`Adam(state_rows=...)`: moment tables of `state_rows` rows instead of the shard's size, a row gets its
    pair of state rows when it is first stepped (bess_assign_state_rows).  With room for every touched row
    the trajectory is that of full-size state; an exhausted pool is reported and does not fault."""
    from besskge import runtime
    from test_hip_parity import build_model

    c = load_bess_case(case)
    keys = ("head", "relation", "tail", "negative", "negative_mask")
    bps = c["meta"]["bps"]
    batches = [{k: c["batch"][k].flatten(end_dim=1)[i: i + 1] for k in keys if k in c["batch"]} for i in range(bps)]
    out = []
    for rows in (None, 116):  # the golden shard has 120 rows, ~110 of them touched over the four steps
        model = build_model(c, dev)
        opt = runtime.Adam(lr=0.01, weight_decay=0.01, state_rows=rows)
        runner = runtime.training_model(model, runtime.Options(device_iterations=1, use_graphs=graphs), opt, device=dev)
        for i in range(4):
            runner(**batches[i % bps])
        out.append(model.score_fn.entity_embedding.detach().float().cpu().clone())
        if rows is not None:
            (used, cap), = model.optimizer_state_rows_used().values()
            assert cap == 116 and 0 < used <= 116
            st = next(v for v in model._optimizer_state.values() if "slot_map" in v)
            assert all(tuple(t.shape) == (116, model.entity_embedding_size) for t in st["s"])
    torch.testing.assert_close(out[1], out[0], rtol=1e-5, atol=1e-6)
    if not graphs:
        model = build_model(c, dev)
        runner = runtime.training_model(model, optimizer=runtime.Adam(lr=0.01, state_rows=8), device=dev)
        for i in range(3):
            runner(**batches[i % bps])
        (used, cap), = model.optimizer_state_rows_used().values()
        assert cap == 8 and used > 8  # exhausted: reported, rows beyond the pool were stepped from zero state
        assert bool(torch.isfinite(model.score_fn.entity_embedding.detach().float()).all())


@pytest.mark.parametrize("opt_name", ["adamw", "sgdm_wd", "adagrad"])
@pytest.mark.parametrize("case", ["tr_EM_ComplEx0_h_pt_n1", "tr_EM_TransE1_t_flat_n1", "tr_EM_RotatE2_ht_flat_n2"])
def test_dense_optimizer_follows_torch_optim_step_for_step(dev, opt_name, case):
    """`dense=True`: the optimisers of the notebooks as they are (`poptorch.optim.AdamW` on dense gradients, reference
    `notebooks/1_biokg_training_inference.ipynb:525-531`): EVERY row is stepped in every update - decoupled weight
    decay on all rows, moments of untouched rows decay.  Three steps (both micro-batches of the fixture, then the
    first again) against oracle autograd + torch.optim on the dense tables: the trajectories agree row for row,
    untouched rows included (the row-lazy default leaves those where they were)."""
    from besskge import runtime
    from test_hip_parity import build_model

    c = load_bess_case(case)
    keys = ("head", "relation", "tail", "negative", "negative_mask")
    batches = [{k: c["batch"][k][it] for k in keys if k in c["batch"]} for it in (0, 1, 0)]
    if opt_name == "adamw":
        opt = runtime.Adam(lr=0.01, weight_decay=0.1, dense=True)
        topt = lambda ps: torch.optim.AdamW(ps, lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)  # noqa: E731
    elif opt_name == "sgdm_wd":
        opt = runtime.SGD(lr=0.05, momentum=0.9, weight_decay=0.01, dense=True)
        topt = lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, weight_decay=0.01)  # noqa: E731
    else:
        opt = runtime.Adagrad(lr=0.1, eps=1e-10, dense=True)
        topt = lambda ps: torch.optim.Adagrad(ps, lr=0.1, eps=1e-10)  # noqa: E731
    model = build_model(c, dev)
    runner = runtime.training_model(model, runtime.Options(device_iterations=1), opt, device=dev)
    t0 = c["table"].clone().requires_grad_(True)
    r0 = c["rel"].clone().requires_grad_(True)
    tor = topt([t0, r0])
    solid_t = torch.ones_like(t0, dtype=torch.bool)
    solid_r = torch.ones_like(r0, dtype=torch.bool)
    for b in batches:
        runner(**b)
        tor.zero_grad()
        want = kge.bess_step(c["spec"], c["model_cls"], t0, r0, b, c["loss"])
        torch.stack(want["loss"]).sum().backward()
        solid_t &= (t0.grad.abs() > 1e-4) | (t0.grad == 0)
        solid_r &= (r0.grad.abs() > 1e-4) | (r0.grad == 0)
        tor.step()
    got_t = model.score_fn.entity_embedding.detach().float().cpu()
    got_r = model.score_fn.relation_embedding.detach().float().cpu()
    # (Adam / Adagrad divide by |g|: an entry whose gradient cancels to ~0 - exact 0 in dense autograd, 1e-9 from the
    # kernels' sums - takes a +-lr step; compared where the gradient of EVERY step was solid, and on every untouched row)
    moved = (t0.detach() - c["table"]).abs() > 0
    assert bool(moved.reshape(-1, moved.shape[-1]).any(-1).all()) or opt_name == "adagrad"  # weight decay moves every row
    for got, ref, solid in ((got_t, t0.detach(), solid_t), (got_r, r0.detach(), solid_r)):
        if opt_name == "sgdm_wd":  # (linear in the gradient: every entry is compared)
            solid = torch.ones_like(solid)
        assert float(solid.float().mean()) > 0.5
        torch.testing.assert_close(got[solid], ref[solid], rtol=2e-3, atol=1e-4)
    # the accumulator is left zero
    for scratch in model._direct_acc.values():
        assert float(scratch.acc.abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["graph", "plan"])
def test_dense_optimizer_replays_from_a_graph_and_a_plan(dev, mode):
    """`dense=True` under `Options.use_graphs` / `use_plans`: four AdamW steps (per-triple negatives of the own shard:
    fused forward + `bess_pertriple_tail` + K9 into the dense accumulator + one pass over the whole shard) replayed
    equal the eager steps to the order of the fp32 additions."""
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import ComplEx
    from besskge.sharding import Sharding

    S_, K_, M = 192, 64, 3000
    sharding = Sharding.create(M, 1, seed=0)
    rng = np.random.default_rng(0)
    batches = []
    for _ in range(4):
        b = dict(head=rng.integers(M, size=(1, 1, S_)), relation=rng.integers(9, size=(1, 1, S_)),
                 tail=rng.integers(M, size=(1, 1, S_)), negative=rng.integers(M, size=(1, 1, S_, K_)))
        batches.append({k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in b.items()})
    res = {}
    for how in ("eager", mode):
        torch.manual_seed(1)
        fn = ComplEx(False, sharding, 9, 32, device=dev)
        ns = RandomShardedNegativeSampler(K_, sharding, 0, "t", local_sampling=False, flat_negative_format=False)
        model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn,
                                       loss_fn=LogSigmoidLoss(margin=4.0, negative_adversarial_sampling=True))
        opt = runtime.Adam(lr=0.01, weight_decay=0.1, dense=True)
        runner = runtime.training_model(model, runtime.Options(use_graphs=how == "graph", use_plans=how == "plan"), opt,
                                        device=dev)
        losses = [float(runner(**b)["loss"]) for b in batches]
        torch.cuda.synchronize()
        res[how] = (losses, model.score_fn.entity_embedding.detach().float().clone(),
                    model.score_fn.relation_embedding.detach().float().clone())
    np.testing.assert_allclose(res[mode][0], res["eager"][0], rtol=1e-5)
    # (Adam divides by |g|: an entry whose gradient cancels to ~0 takes a +-lr step whose sign follows the order of the
    # additions - a handful of entries at most)
    for i in (1, 2):
        off = (res[mode][i] - res["eager"][i]).abs()
        assert float((off > 1e-4).float().mean()) < 0.005, float(off.max())
