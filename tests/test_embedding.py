"""Table allocation / re-sharding (CPU): golden part against vectors produced by
the reference's `initialize_entity_embedding` / `refactor_embedding_sharding`
(tests/golden/embedding.npz), plus shape / dtype / validation behaviour of
reference `besskge/embedding.py:107-290`."""

import dataclasses

import numpy as np
import pytest
import torch

from besskge.embedding import (
    init_KGE_normal,
    init_KGE_uniform,
    init_uniform_norm,
    init_uniform_rotation,
    init_xavier_norm,
    initialize_entity_embedding,
    initialize_relation_embedding,
    refactor_embedding_sharding,
)
from besskge.sharding import Sharding

from conftest import load_golden


def sharding_from(g, prefix, n_shard):
    return Sharding(
        n_shard=n_shard,
        entity_to_shard=g[prefix + "entity_to_shard"],
        entity_to_idx=g[prefix + "entity_to_idx"],
        shard_and_idx_to_entity=g[prefix + "shard_and_idx_to_entity"],
        shard_counts=g[prefix + "shard_counts"],
        entity_type_counts=None,
        entity_type_offsets=None,
    )


def test_unsharded_table_and_resharding_golden():
    g = load_golden("embedding")
    old = sharding_from(g, "old_", 4)
    new = sharding_from(g, "new_", 3)
    # the goldens' shardings are what Sharding.create gives
    for s, (n, seed) in ((old, (4, 3)), (new, (3, 9))):
        ref = Sharding.create(101, n, seed=seed)
        assert np.array_equal(ref.shard_and_idx_to_entity, s.shard_and_idx_to_entity)
    table = initialize_entity_embedding(old, torch.from_numpy(g["unsharded"]))
    assert isinstance(table, torch.nn.Parameter) and table.dtype == torch.float32
    assert torch.equal(table.data, torch.from_numpy(g["sharded_old"]))
    again = refactor_embedding_sharding(table, old, new)
    assert torch.equal(again.data, torch.from_numpy(g["sharded_new"]))
    # every real entity keeps its row; padding rows copy the last entity
    flat = again.data[torch.from_numpy(new.entity_to_shard), torch.from_numpy(new.entity_to_idx)]
    assert torch.equal(flat, torch.from_numpy(g["unsharded"]))


def test_initializers_and_shapes():
    s = Sharding.create(1000, 4, seed=0)
    torch.manual_seed(0)
    t = initialize_entity_embedding(s, [init_KGE_uniform, init_KGE_normal], [8, 24])
    assert t.shape == (4, 250, 32) and t.dtype == torch.float32 and t.requires_grad
    assert float(t[..., :8].abs().max()) <= 1.0 / 8
    # same random stream as allocating the pieces one after the other
    torch.manual_seed(0)
    a = init_KGE_uniform(torch.empty(4, 250, 8))
    b = init_KGE_normal(torch.empty(4, 250, 24))
    assert torch.equal(t.data, torch.cat([a, b], dim=-1))
    r = initialize_relation_embedding(7, True, [init_uniform_rotation], [16])
    assert r.shape == (14, 16) and float(r.min()) >= 0 and float(r.max()) <= 2 * np.pi + 1e-6
    assert initialize_relation_embedding(7, False, [init_xavier_norm], [16]).shape == (7, 16)
    u = init_uniform_norm(torch.empty(5, 9))
    torch.testing.assert_close(u.norm(dim=-1), torch.ones(5))
    # sharded tensor initializer, selected shards only (extension)
    full = torch.randn(4, 250, 6)
    sub = initialize_entity_embedding(s, full, shards=[2])
    assert torch.equal(sub.data, full[2:3])
    only = initialize_entity_embedding(s, [init_KGE_normal], [6], shards=[1, 3])
    assert only.shape == (2, 250, 6)


def test_validation_errors():
    s = Sharding.create(100, 4, seed=0)
    with pytest.raises(ValueError):
        initialize_entity_embedding(s, torch.zeros(3, 25, 4))  # wrong shard count
    with pytest.raises(ValueError):
        initialize_entity_embedding(s, torch.zeros(99, 4))  # wrong entity count
    with pytest.raises(ValueError):
        initialize_entity_embedding(s, torch.zeros(4))
    with pytest.raises(ValueError):
        initialize_entity_embedding(s, [init_KGE_normal], None)
    with pytest.raises(ValueError):
        initialize_entity_embedding(s, [init_KGE_normal], [2, 3])
    with pytest.raises(AssertionError):
        initialize_entity_embedding(s, torch.zeros(100, 4), [5])
    with pytest.raises(ValueError):
        initialize_relation_embedding(3, False, torch.zeros(3, 2, 2))


def test_scorer_constructors_allocate_like_the_reference():
    from besskge.scoring import ComplEx, DistMult, RotatE, TransE

    s = Sharding.create(103, 4, seed=0)
    for cls, args, W, Wr in ((TransE, (True, 1), 16, 16), (RotatE, (False, 2), 32, 16), (DistMult, (True,), 16, 16),
                             (ComplEx, (False,), 32, 32)):
        fn = cls(*args, s, 5, 16, inverse_relations=(cls is RotatE))
        assert fn.entity_embedding.shape == (4, 26, W)
        assert fn.relation_embedding.shape == (10 if cls is RotatE else 5, Wr)
        assert fn.entity_embedding.dtype == torch.float32 and fn.embedding_size == 16
        d = fn.kernel_desc()
        assert (d.width, d.rel_width) == (W, Wr)
    fn = TransE(True, 1, s, 5, 16)
    new = Sharding.create(103, 2, seed=1)
    before = fn.entity_embedding.data[torch.from_numpy(s.entity_to_shard), torch.from_numpy(s.entity_to_idx)]
    fn.update_sharding(new)
    after = fn.entity_embedding.data[torch.from_numpy(new.entity_to_shard), torch.from_numpy(new.entity_to_idx)]
    assert fn.sharding is new and torch.equal(before, after)
    assert TransE(True, 3, s, 5, 16).kernel_desc().norm_p == 3  # any p >= 1, like the reference's torch.norm
    with pytest.raises(ValueError):
        TransE(True, 0, s, 5, 16)
    from besskge.scoring import PairRE

    assert PairRE(True, 3, s, 5, 16).scoring_norm == 3  # (round 4: any p >= 1, as the reference - scoring.py:540-593)
    with pytest.raises(ValueError):
        PairRE(True, 0, s, 5, 16)


def test_scorer_constructors_place_only_the_hosted_shards():
    """`device=`, `shards=`, `dtype=` of the scorer constructors (extension): a rank allocates just
    its own slice, in its final dtype; defaults stay the reference's (whole table, fp32)."""
    import besskge.embedding as emb
    from besskge.scoring import ComplEx, TransE
    from besskge.sharding import Sharding

    sharding = Sharding.create(1000, 4, seed=0)
    M = sharding.max_entity_per_shard
    torch.manual_seed(0)
    whole = TransE(True, 1, sharding, 7, 16)
    assert whole.entity_embedding.shape == (4, M, 16) and whole.entity_embedding.dtype == torch.float32
    fn = TransE(True, 1, sharding, 7, 16, shards=[2], dtype=torch.float16, device=torch.device("cpu"))
    assert fn.entity_embedding.shape == (1, M, 16) and fn.entity_embedding.dtype == torch.float16
    assert fn.relation_embedding.shape == (7, 16) and fn.relation_embedding.dtype == torch.float16
    assert float(fn.entity_embedding.float().abs().max()) <= 1.0 / 16 + 1e-3  # init_KGE_uniform bound
    # block-wise creation in the narrow dtype: several blocks cover every row
    old = emb._INIT_CHUNK_SCALARS
    emb._INIT_CHUNK_SCALARS = 100 * 32
    try:
        cx = ComplEx(False, sharding, 7, 16, shards=[0, 3], dtype=torch.float16)
    finally:
        emb._INIT_CHUNK_SCALARS = old
    assert cx.entity_embedding.shape == (2, M, 32)
    assert bool((cx.entity_embedding.float().abs().sum(-1) > 0).all())
    with pytest.raises(TypeError, match="unexpected keyword"):
        TransE(True, 1, sharding, 7, 16, bogus=1)
