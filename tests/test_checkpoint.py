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
