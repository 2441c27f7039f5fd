"""Per-shard streaming checkpoints (besskge/checkpoint.py): SURVEY.md §5 "a 1 TB table needs per-shard
streaming save"; the reference's own round trip (`entity_initializer=<tensor>`, reference
embedding.py:135-163) is covered by tests/test_embedding.py."""

import numpy as np
import pytest
import torch


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.int32])
@pytest.mark.parametrize("shape,chunk", [((1000, 24), 1 << 10), ((7, 3), 1 << 20), ((0, 8), 64), ((33,), 16)])
def test_rows_round_trip_in_chunks(tmp_path, dtype, shape, chunk):
    from besskge import checkpoint

    g = torch.Generator().manual_seed(0)
    t = (torch.randn(shape, generator=g) * 100).to(dtype)
    f = tmp_path / "rows.npy"
    checkpoint.save_rows(t, f, chunk_bytes=chunk)
    assert np.array_equal(np.load(f), t.numpy())  # a plain .npy file
    back = torch.empty_like(t)
    checkpoint.load_rows(f, back, chunk_bytes=chunk)
    assert torch.equal(back, t)
    with pytest.raises(ValueError):
        checkpoint.load_rows(f, torch.empty((shape[0] + 1,) + tuple(shape[1:]), dtype=dtype))
    assert not list(tmp_path.glob("*.tmp*"))


def _model(dev, n_shard=2, dtype=torch.float32):
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import ComplEx
    from besskge.sharding import Sharding

    torch.manual_seed(0)
    sharding = Sharding.create(900, n_shard, seed=1)
    fn = ComplEx(False, sharding, 7, 16, device=dev, dtype=dtype)
    ns = RandomShardedNegativeSampler(6, sharding, 3, "h", local_sampling=False, flat_negative_format=False)
    return EmbeddingMovingBessKGE(ns, fn, LogSigmoidLoss(margin=2.0, negative_adversarial_sampling=True)), sharding


def _batch(sharding, n, S, K, seed):
    rng = np.random.default_rng(seed)
    M = int(sharding.shard_counts.min())
    b = dict(head=rng.integers(M, size=(n, n, S)), relation=rng.integers(7, size=(n, n, S)),
             tail=rng.integers(M, size=(n, n, S)), negative=rng.integers(M, size=(n, n, n * S, K)))
    return {k: torch.from_numpy(v.astype(np.int32)) for k, v in b.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("opt_name", ["sgd", "adam", "adam_paged"])
def test_training_resumes_from_a_checkpoint(tmp_path, opt_name):
    """k steps, save, k more steps == the same steps after loading the checkpoint into a fresh model."""
    from besskge import checkpoint, runtime

    dev = torch.device("cuda", 0)

    def optimizer():
        if opt_name == "sgd":
            return runtime.SGD(lr=0.05, momentum=0.9)
        return runtime.Adam(lr=0.01, weight_decay=0.01, state_rows=300 if opt_name == "adam_paged" else None)

    model, sharding = _model(dev)
    runner = runtime.training_model(model, optimizer=optimizer(), device=dev)
    batches = [_batch(sharding, 2, 16, 6, s) for s in range(4)]
    for b in batches[:2]:
        runner(**b)
    checkpoint.save_checkpoint(model, tmp_path / "ckpt", chunk_bytes=4096)
    for b in batches[2:]:
        want = runner(**b)
    want_ent = model.score_fn.entity_embedding.detach().clone()
    want_rel = model.score_fn.relation_embedding.detach().clone()

    fresh, _ = _model(dev)
    with torch.no_grad():
        fresh.score_fn.entity_embedding.add_(1.0)  # anything but the saved rows
    runner2 = runtime.training_model(fresh, optimizer=optimizer(), device=dev)
    checkpoint.load_checkpoint(fresh, tmp_path / "ckpt", chunk_bytes=4096)
    for b in batches[2:]:
        got = runner2(**b)
    # (the relation gradient is summed with fp32 atomics: the last bits depend on their order, run to run;
    # a lost momentum / Adam state or step count would be off by orders of magnitude more)
    tol = dict(rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(got["loss"], want["loss"], **tol)
    torch.testing.assert_close(fresh.score_fn.entity_embedding.detach(), want_ent, **tol)
    torch.testing.assert_close(fresh.score_fn.relation_embedding.detach(), want_rel, **tol)
    files = sorted(p.name for p in (tmp_path / "ckpt").iterdir())
    assert "entity_shard0.npy" in files and "entity_shard1.npy" in files and "relation.npy" in files
    if opt_name == "adam_paged":
        assert "entity_shard0.slots.npy" in files
        assert np.load(tmp_path / "ckpt" / "entity_shard0.state0.npy").shape[0] == 300


@pytest.mark.gpu
@pytest.mark.parametrize("warm", [True, False])
def test_rollback_in_a_live_process_under_graph_replay(tmp_path, warm):
    """A Runner that replays hipGraphs keeps training correctly after `load_checkpoint` into its live model:
    state of the same layout is read into the tensors the graph was recorded on (warm), and a load that has to
    create or replace state tensors (the graph was recorded before the model had Adam state of that layout)
    makes the Runner record again instead of replaying stale addresses."""
    from besskge import checkpoint, runtime

    dev = torch.device("cuda", 0)
    batches = None
    results = {}
    for use_graphs in (False, True):
        model, sharding = _model(dev)
        batches = batches or [_batch(sharding, 2, 16, 6, s) for s in range(5)]
        runner = runtime.training_model(model, runtime.Options(use_graphs=use_graphs),
                                        runtime.Adam(lr=0.01), device=dev)
        for b in batches[:2]:
            runner(**b)
        ckpt = tmp_path / f"ckpt{int(use_graphs)}"
        checkpoint.save_checkpoint(model, ckpt, chunk_bytes=4096)
        for b in batches[2:4]:
            runner(**b)
        if not warm:
            # a different live layout: paged state with its own (smaller) moment tables
            model2, _ = _model(dev)
            runner = runtime.training_model(model2, runtime.Options(use_graphs=use_graphs),
                                            runtime.Adam(lr=0.01), device=dev)
            runner.optimizer = runtime.Adam(lr=0.01, state_rows=300)
            runner(**batches[4])  # records a graph on paged state tensors
            runner.optimizer = runtime.Adam(lr=0.01)
            model = model2
        checkpoint.load_checkpoint(model, ckpt, chunk_bytes=4096)  # back to the state after two steps
        for b in batches[2:4]:
            out = runner(**b)
        results[use_graphs] = (model.score_fn.entity_embedding.detach().clone(),
                               model.score_fn.relation_embedding.detach().clone(), out["loss"].clone())
        # the step count went back with the tables (Adam's bias correction restarts from step 2, not 4)
        st = model._optimizer_state[model._local_table(0).data_ptr()]
        assert checkpoint._step_of(st) == 4
    # Adam normalises by |g|: entries whose gradient cancels to ~0 take a full +-lr step whose sign depends on
    # the order of fp32 atomics (tests/test_accumulation.py) - a stale-address replay would leave the tables
    # where the rollback put them (off by whole steps everywhere), which this still catches
    for a, b in zip(results[False][:2], results[True][:2]):
        off = (a - b).abs()
        assert float((off > 1e-5).float().mean()) < 0.01 and float(off.max()) <= 2 * 0.01 * 1.01
    torch.testing.assert_close(results[False][2], results[True][2], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_state_of_another_optimiser_is_not_merged(tmp_path):
    """a momentum-SGD checkpoint (one state tensor) loaded into a model that is live under Adam (two)
    used to be read in place - first moment = momentum buffer, second moment zeroed, the checkpoint's step count
    kept: no bias correction, first updates ~30x too large.  Now the file's optimiser kind is compared and the
    load refuses; a checkpoint WITHOUT optimiser history resets the live state in place."""
    from besskge import checkpoint, runtime

    dev = torch.device("cuda", 0)
    model, sharding = _model(dev)
    batches = [_batch(sharding, 2, 16, 6, s) for s in range(3)]
    fresh_dir = tmp_path / "fresh"
    checkpoint.save_checkpoint(model, fresh_dir, chunk_bytes=4096)  # no optimiser history yet
    runner = runtime.training_model(model, optimizer=runtime.SGD(lr=0.05, momentum=0.9), device=dev)
    runner(**batches[0])
    checkpoint.save_checkpoint(model, tmp_path / "sgdm", chunk_bytes=4096)

    live, _ = _model(dev)
    adam = runtime.training_model(live, optimizer=runtime.Adam(lr=0.01), device=dev)
    adam(**batches[0])
    with pytest.raises(ValueError, match="optimiser kind"):
        checkpoint.load_checkpoint(live, tmp_path / "sgdm", chunk_bytes=4096)
    # into a model without live state the file's state comes in as it is - and stepping it with Adam is refused
    cold, _ = _model(dev)
    cold_runner = runtime.training_model(cold, optimizer=runtime.Adam(lr=0.01), device=dev)
    checkpoint.load_checkpoint(cold, tmp_path / "sgdm", chunk_bytes=4096)
    with pytest.raises(RuntimeError, match="optimiser kind"):
        cold_runner(**batches[1])
    # a history-free checkpoint under live Adam: tables back, moments zero, step count 0 (bias correction restarts)
    st = live._optimizer_state[live._local_table(0).data_ptr()]
    ptrs = [x.data_ptr() for x in st["s"]]
    checkpoint.load_checkpoint(live, fresh_dir, chunk_bytes=4096)
    st = live._optimizer_state[live._local_table(0).data_ptr()]
    assert [x.data_ptr() for x in st["s"]] == ptrs and checkpoint._step_of(st) == 0
    assert all(float(x.abs().max()) == 0.0 for x in st["s"])
    before = live.score_fn.entity_embedding.detach().clone()
    adam(**batches[1])
    step = (live.score_fn.entity_embedding.detach() - before).abs().max()
    assert 0 < float(step) <= 0.01 * 1.01  # one bias-corrected Adam step is at most lr
