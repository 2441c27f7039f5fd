"""BASELINE configs[4] at its REAL per-shard size: 62.5 M rows x 512 fp32 = 128 GB on one MI355X.

The shard is created on the device through the public scorer constructor (`device=`, `shards=`:
no host copy), rows are gathered / scored / updated over the whole address range (64-bit row
offsets, TLB-miss regime) and checked at full size against torch arithmetic on the gathered rows
(a 1 GB subset): scores of the dominant kernel, and one sparse SGD training step.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

M_C5, D_C5 = 62_500_000, 512


def test_config5_shard_at_full_size():
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import DistMult
    from besskge.sharding import Sharding

    dev = torch.device("cuda", 0)
    free, total = torch.cuda.mem_get_info(dev)
    if free < 150e9:
        pytest.skip(f"needs 150 GB of free HBM, {free / 1e9:.0f} GB available")
    n_rel, S, K = 1000, 2048, 64
    sharding = Sharding.create(M_C5, 1, seed=7)  # one shard of the 8-way split of 5e8 entities
    torch.manual_seed(0)
    fn = DistMult(False, sharding, n_rel, D_C5, device=dev, shards=[0])
    assert fn.entity_embedding.shape == (1, M_C5, D_C5) and fn.entity_embedding.is_cuda
    assert fn.entity_embedding.numel() * 4 == 128_000_000_000
    # make the rows O(1) so that scores are well away from zero (default init is U(-1/W, 1/W)); in place, blockwise
    tab = fn.entity_embedding.data[0]
    for lo in range(0, M_C5, 4_000_000):
        tab[lo: lo + 4_000_000].mul_(D_C5)
    fn.relation_embedding.data.mul_(D_C5)
    ns = RandomShardedNegativeSampler(K, sharding, 1, "t", local_sampling=False, flat_negative_format=False)
    model = EmbeddingMovingBessKGE(ns, fn, LogSigmoidLoss(6.0, True, 0.5), return_scores=True)
    rng = np.random.default_rng(3)
    # rows spread over the whole shard, the very last row and rows around the 4 GiB / 64 GiB marks included
    marks = np.array([0, M_C5 - 1, (1 << 32) // 2048, (1 << 32) // 2048 + 1, (64 << 30) // 2048, (64 << 30) // 2048 + 1])
    neg = rng.integers(M_C5, size=(1, 1, S, K))
    neg[0, 0, 0, : len(marks)] = marks
    batch = dict(head=rng.integers(M_C5, size=(1, 1, S)), relation=rng.integers(n_rel, size=(1, 1, S)),
                 tail=rng.integers(M_C5, size=(1, 1, S)), negative=neg)
    batch = {k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in batch.items()}
    runner = runtime.inference_model(model, device=dev)
    res = runner(**batch)
    # the same scores from torch on the gathered rows (fp64 on the device)
    h = tab[batch["head"].flatten().long()].double()
    t = tab[batch["tail"].flatten().long()].double()
    r = fn.relation_embedding.data[batch["relation"].flatten().long()].double()
    e = tab[batch["negative"].reshape(S, K).long()].double()  # [S, K, W]: 0.5 GB in fp64 for S = 2048
    want_pos = (h * r * t).sum(-1)
    want_neg = torch.einsum("sw,skw->sk", h * r, e)
    scale = float(want_neg.abs().max())
    torch.testing.assert_close(res["positive_score"].double(), want_pos, rtol=1e-5, atol=2e-6 * scale)
    torch.testing.assert_close(res["negative_score"].double(), want_neg, rtol=1e-5, atol=2e-6 * scale)

    # one sparse SGD step: touched rows move by -lr * (autograd gradient on the gathered rows), nothing else moves
    rows = torch.unique(torch.cat([batch["head"].flatten(), batch["tail"].flatten(), batch["negative"].flatten()]).long())
    before = tab[rows].clone()
    probe = torch.from_numpy(rng.integers(M_C5, size=4096)).to(dev)
    probe = probe[~torch.isin(probe, rows)]
    probe_before = tab[probe].clone()
    sub = before.clone().requires_grad_(True)  # the touched rows as a small table; indices remapped into it
    remap = lambda x: torch.searchsorted(rows, x.long())
    hh, tt = sub[remap(batch["head"].flatten())], sub[remap(batch["tail"].flatten())]
    ee = sub[remap(batch["negative"].reshape(S, K))]
    rel = fn.relation_embedding.data.clone().requires_grad_(True)
    rr = rel[batch["relation"].flatten().long()]
    pos = (hh * rr * tt).sum(-1)
    negs = torch.einsum("sw,skw->sk", hh * rr, ee)
    a = torch.softmax(0.5 * negs, dim=-1).detach()
    loss = -0.5 * (torch.nn.functional.logsigmoid(pos + 6.0) + (a * torch.nn.functional.logsigmoid(-negs - 6.0)).sum(-1)).sum()
    loss.backward()
    lr = 0.25
    trainer = runtime.training_model(model, optimizer=runtime.SGD(lr=lr), device=dev)
    out = trainer(**batch)
    torch.testing.assert_close(out["loss"].reshape(()), loss.detach().float(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(tab[rows], before - lr * sub.grad, rtol=1e-4, atol=2e-5)
    assert torch.equal(tab[probe], probe_before)
    torch.testing.assert_close(fn.relation_embedding.data, (rel - lr * rel.grad).detach(), rtol=1e-4, atol=2e-5)
