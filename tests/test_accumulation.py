"""Gradient accumulation in the runner (`Options.gradient_accumulation`, PopTorch's
`Training.gradientAccumulation`; reference `notebooks/1_biokg_training_inference.ipynb:408-417,470-477`,
`2_yago_topk_prediction.ipynb:240-280`).

k micro-batches with accumulation = ONE optimiser step on the sum of their losses, every micro-batch
differentiated against the same tables.  The expected tables come from the oracle (autograd on the dense
tables, `torch.optim` step from zero state, where dense and lazy semantics coincide); micro-batches are the
reference-generated ones of the golden cases."""

import numpy as np
import pytest
import torch

from oracle import kge

from test_oracle import load_bess_case, step_batch

KEYS = ("head", "relation", "tail", "negative", "negative_mask")


# ----------------------------------------------------------------------------- host logic (no GPU)
def test_options_spelling_and_row_check():
    from besskge import runtime

    o = runtime.Options()
    assert o.deviceIterations(8) is o
    assert o.Training.gradientAccumulation(6) is o
    assert (o.device_iterations, o.gradient_accumulation, o.batches_per_call) == (8, 6, 48)
    o.replication_factor = 4  # the notebooks set it; plain attribute here
    o._popart.setPatterns(dict(RemoveAllReducePattern=True))  # accepted, nothing to do
    o.Training.accumulationAndReplicationReductionType("Mean")
    assert o.accumulation_reduction == "mean"
    assert o.output_mode is None and o.outputMode("All").output_mode == "all"
    with pytest.raises(ValueError):
        o.Training.gradientAccumulation(0)
    with pytest.raises(ValueError):
        o.Training.accumulationAndReplicationReductionType("max")


# ----------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _optimizers(name):
    from besskge import runtime

    if name == "sgd":
        return runtime.SGD(lr=0.05), lambda ps: torch.optim.SGD(ps, lr=0.05)
    if name == "sgdm":
        return runtime.SGD(lr=0.05, momentum=0.9), lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9)
    if name == "adamw":
        # (weight decay 0: with it the dense torch.optim.AdamW also moves rows no gradient touched - the lazy
        # semantics of the row-sparse optimisers, DESIGN.md section 3, differ there by design)
        return (runtime.Adam(lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0),
                lambda ps: torch.optim.AdamW(ps, lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0))
    if name == "adagrad":
        return runtime.Adagrad(lr=0.1, eps=1e-10), lambda ps: torch.optim.Adagrad(ps, lr=0.1, eps=1e-10)
    raise KeyError(name)


def _oracle_step(c, order, topt, dtype):
    """Tables after ONE torch.optim step on the summed loss of the micro-batches `order`."""
    table, rel = c["table"], c["rel"]
    if dtype == torch.float16:
        table, rel = table.half().float(), rel.half().float()
    t0 = table.clone().requires_grad_(True)
    r0 = rel.clone().requires_grad_(True)
    losses = []
    total = 0.0
    for it in order:
        res = kge.bess_step(c["spec"], c["model_cls"], t0, r0, step_batch(c["batch"], it), c["loss"])
        losses.append(torch.stack(res["loss"]).detach())
        total = total + torch.stack(res["loss"]).sum()
    total.backward()
    opt = topt([t0, r0])
    opt.step()
    return t0, r0, table, rel, losses


def _compare(model, t0, r0, table, rel, dtype, normalising):
    got_t = model.score_fn.entity_embedding.detach().float().cpu()
    got_r = model.score_fn.relation_embedding.detach().float().cpu()
    for got, ref, grad, before in ((got_t, t0.detach(), t0.grad, table), (got_r, r0.detach(), r0.grad, rel)):
        W = grad.shape[-1]
        if dtype == torch.float16:
            # the shard is stored in fp16: one rounding of the updated row (round-once update)
            tol = dict(rtol=2e-3, atol=2e-3 if normalising else 1e-3)
        else:
            tol = dict(rtol=2e-3, atol=5e-5) if normalising else dict(rtol=1e-4, atol=2e-5)
        # sign-normalising optimisers turn an analytically cancelling gradient entry into a full step:
        # compare entries with a solid gradient there (as tests/test_optimizers.py does)
        solid = grad.abs() > 1e-4 if normalising else torch.ones_like(grad, dtype=torch.bool)
        torch.testing.assert_close(got[solid], ref[solid], **tol)
        untouched = (grad.reshape(-1, W) == 0).all(dim=-1)
        assert torch.equal(got.reshape(-1, W)[untouched], before.reshape(-1, W)[untouched])


CASES = [
    "tr_EM_TransE1_t_flat_n1",    # shared negatives, one shard
    "tr_EM_ComplEx0_h_pt_n1",     # per-triple negatives of the own shard: fused forward + segmented K9
    "tr_EM_RotatE2_ht_pt_n2",     # two per-triple groups per replica, exchange
    "tr_EM_DistMult0_ht_flat_n2",
    "tr_SM_TransE1_t_pt_n2",      # ScoreMoving with per-shard softmax partials
    "tr_SM_ht_flat_n2",
    "tr_EM_aug_t_flat_n4",        # the wikikg2 recipe's shape: flat negatives + augmentation
    "tr_EM_loc_aug_ht_flat_n4",
    "tr_EM_PairRE1_h_pt_n1",      # affine family
    "tr_EM_BoxE1_ht_pt_n2",
    "tr_EM_ConvE0_t_flat_n2",     # dense parameters accumulate too
]


@pytest.mark.gpu
@pytest.mark.parametrize("opt_name", ["sgd", "adamw", "sgdm"])
@pytest.mark.parametrize("case", CASES)
def test_accumulated_step_equals_one_step_on_the_summed_loss(dev, case, opt_name):
    from besskge import runtime
    from test_hip_parity import build_model

    c = load_bess_case(case)
    n = c["meta"]["n_shard"]
    order = [0, 1, 0]  # k = 3 micro-batches (the goldens hold two; one is used twice)
    opt, topt = _optimizers(opt_name)
    model = build_model(c, dev)
    if c["net"] is not None:
        model.train()
    options = runtime.Options(device_iterations=1).outputMode("all")
    options.Training.gradientAccumulation(len(order))
    runner = runtime.training_model(model, options, opt, device=dev)
    batch = {k: torch.stack([c["batch"][k][it] for it in order]).flatten(end_dim=1) for k in KEYS if k in c["batch"]}
    res = runner(**batch)
    t0, r0, table, rel, losses = _oracle_step(c, order, topt, torch.float32)
    # every micro-batch's own loss comes back (OutputMode.All), computed from the same tables
    np.testing.assert_allclose(res["loss"].float().cpu().numpy().reshape(len(order), n),
                               torch.stack(losses).numpy(), rtol=1e-4, atol=1e-4)
    if c["net"] is None:
        _compare(model, t0, r0, table, rel, torch.float32, opt_name != "sgd")


@pytest.mark.gpu
@pytest.mark.parametrize("opt_name", ["sgd", "adamw"])
@pytest.mark.parametrize("case", ["tr_EM_TransE1_t_flat_n1", "tr_EM_ComplEx0_h_pt_n1", "tr_EM_aug_t_flat_n4",
                                  "tr_SM_TransE1_t_pt_n2"])
def test_accumulation_on_fp16_tables(dev, case, opt_name):
    from besskge import runtime
    from test_hip_parity import build_model

    c = load_bess_case(case)
    order = [0, 1]
    opt, topt = _optimizers(opt_name)
    model = build_model(c, dev)
    model.score_fn.fp32_math = True  # (the oracle is run on the fp16 tables' values in fp32)
    options = runtime.Options(device_iterations=1, gradient_accumulation=2)
    runner = runtime.training_model(model, options, opt, device=dev, dtype=torch.float16)
    batch = {k: torch.stack([c["batch"][k][it] for it in order]).flatten(end_dim=1) for k in KEYS if k in c["batch"]}
    runner(**batch)
    t0, r0, table, rel, _ = _oracle_step(c, order, topt, torch.float16)
    _compare(model, t0, r0, table, rel, torch.float16, opt_name != "sgd")


@pytest.mark.gpu
@pytest.mark.parametrize("opt_name", ["sgd", "adamw"])
@pytest.mark.parametrize("case", ["tr_EM_TransE1_t_flat_n1", "tr_EM_ComplEx0_h_pt_n1", "tr_EM_aug_t_flat_n4"])
def test_accumulation_under_graph_replay(dev, case, opt_name):
    """device_iterations x gradient_accumulation micro-batches recorded as one hipGraph: two weight updates of
    two micro-batches each, replayed twice, equal the eager run."""
    from besskge import runtime
    from test_hip_parity import build_model

    c = load_bess_case(case)
    order = [0, 1, 1, 0]
    batch = {k: torch.stack([c["batch"][k][it] for it in order]).flatten(end_dim=1) for k in KEYS if k in c["batch"]}
    tables = []
    for use_graphs in (False, True):
        opt, _ = _optimizers(opt_name)
        model = build_model(c, dev)
        options = runtime.Options(device_iterations=2, gradient_accumulation=2, use_graphs=use_graphs,
                                  output_mode="all")
        runner = runtime.training_model(model, options, opt, device=dev)
        losses = [runner(**batch)["loss"].float().cpu() for _ in range(2)]
        assert losses[0].numel() == 4 * c["meta"]["n_shard"]
        tables.append((model.score_fn.entity_embedding.detach().float().cpu(),
                       model.score_fn.relation_embedding.detach().float().cpu(), losses))
    for a, b in zip(tables[0][:2], tables[1][:2]):
        if opt_name == "sgd":
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
        else:
            # Adam divides by |g|: where a gradient entry cancels to ~0, the order of the fp32 atomics that
            # sum the returned rows decides the sign of a full +-lr step.  Few entries, bounded by 4 steps of lr
            off = (a - b).abs()
            assert float((off > 1e-5).float().mean()) < 0.01 and float(off.max()) <= 4 * 0.01 * 1.01
    for a, b in zip(tables[0][2], tables[1][2]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_mean_reduction_is_sum_with_a_scaled_rate(dev):
    from besskge import runtime
    from test_hip_parity import build_model

    c = load_bess_case("tr_EM_ComplEx0_h_pt_n1")
    order = [0, 1]
    batch = {k: torch.stack([c["batch"][k][it] for it in order]).flatten(end_dim=1) for k in KEYS if k in c["batch"]}
    out = []
    # (the last row: PopTorch's spelling of the entry point brings PopTorch's default - Mean - with it)
    for reduction, lr, make in (("sum", 0.025, runtime.training_model), ("mean", 0.05, runtime.training_model),
                                (None, 0.025, runtime.training_model), (None, 0.05, runtime.trainingModel)):
        model = build_model(c, dev)
        options = runtime.Options(device_iterations=1, gradient_accumulation=2, accumulation_reduction=reduction)
        runner = make(model, options, runtime.SGD(lr=lr), device=dev)
        res = runner(**batch)
        out.append((model.score_fn.entity_embedding.detach().cpu(), model.score_fn.relation_embedding.detach().cpu(),
                    res["loss"].cpu()))
    for other in out[1:]:
        for a, b in zip(out[0], other):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


def test_reduction_defaults_follow_the_entry_point():
    """`accumulationAndReplicationReductionType` is one PopTorch setting for accumulated micro-batches and replicas;
    unset, `training_model` sums (the fixtures' convention) and the PopTorch-spelled `trainingModel` averages
    (PopTorch's documented default); an optimiser without its own `replica_reduction` follows the runner."""
    import dataclasses

    from besskge import runtime

    def resolved(options, default, opt):
        r = runtime.Runner.__new__(runtime.Runner)  # (the constructor places shards on a GPU: host logic only here)
        r.options, r.default_reduction = options, default
        if getattr(opt, "replica_reduction", "") is None:
            opt = dataclasses.replace(opt, replica_reduction=r.reduction)
        return r.reduction, opt.replica_reduction

    assert resolved(runtime.Options(), "sum", runtime.SGD()) == ("sum", "sum")
    assert resolved(runtime.Options(), "mean", runtime.Adam()) == ("mean", "mean")
    assert resolved(runtime.Options(accumulation_reduction="sum"), "mean", runtime.Adagrad()) == ("sum", "sum")
    assert resolved(runtime.Options(), "mean", runtime.SGD(replica_reduction="sum")) == ("mean", "sum")
    o = runtime.Options()
    o.Training.accumulationAndReplicationReductionType("Sum")
    assert resolved(o, "mean", runtime.SGD()) == ("sum", "sum")
    import inspect

    assert inspect.signature(runtime.trainingModel).parameters.keys() == inspect.signature(runtime.training_model).parameters.keys()


@pytest.mark.gpu
def test_training_runner_returns_the_last_micro_batch_by_default(dev):
    """PopTorch's OutputMode.Final for training models (the notebooks' loops read the last batch's loss)."""
    from besskge import runtime
    from test_hip_parity import build_model

    c = load_bess_case("tr_EM_TransE1_t_flat_n1")
    batch = {k: c["batch"][k].flatten(end_dim=1) for k in KEYS if k in c["batch"]}
    got = {}
    for mode in (None, "all"):
        model = build_model(c, dev)
        runner = runtime.training_model(model, runtime.Options(device_iterations=1, gradient_accumulation=2,
                                                               output_mode=mode), runtime.SGD(lr=0.01), device=dev)
        got[mode] = runner(**batch)["loss"].cpu()
    assert got[None].numel() == 1 and got["all"].numel() == 2
    torch.testing.assert_close(got[None], got["all"][-1:])


@pytest.mark.gpu
def test_wrong_row_count_names_the_accumulation_factor(dev):
    from besskge import runtime
    from test_hip_parity import build_model

    c = load_bess_case("tr_EM_TransE1_t_flat_n1")
    model = build_model(c, dev)
    runner = runtime.training_model(model, runtime.Options(device_iterations=1, gradient_accumulation=3),
                                    runtime.SGD(lr=0.1), device=dev)
    batch = {k: c["batch"][k].flatten(end_dim=1) for k in KEYS if k in c["batch"]}  # 2 micro-batches, 3 expected
    with pytest.raises(ValueError, match="gradient_accumulation"):
        runner(**batch)
