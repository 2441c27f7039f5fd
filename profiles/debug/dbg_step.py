import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import numpy as np, torch
from besskge import runtime
from besskge.bess import EmbeddingMovingBessKGE
from besskge.loss import SampledSoftmaxCrossEntropyLoss
from besskge.negative_sampler import RandomShardedNegativeSampler
from besskge.scoring import TransE
from besskge.sharding import Sharding
from oracle import kge

dev = torch.device("cuda", 0)
n_entity, n_rel, d, S, K = 3000, 11, 64, 128, 96
for fused_qt, fp32_math in ((True, False), (False, False), (True, True), (False, True)):
    torch.manual_seed(0)
    sharding = Sharding.create(n_entity, 1, seed=3)
    ent = (torch.randn(1, sharding.max_entity_per_shard, d) * 0.5).half().float()
    rel = (torch.randn(n_rel, d) * 0.5).half().float()
    fn = TransE(True, 1, sharding, n_rel, d, ent, rel)
    fn.supports_fused_query_triple = fused_qt
    fn.fp32_math = fp32_math
    ns = RandomShardedNegativeSampler(K, sharding, 5, "t", local_sampling=False, flat_negative_format=True)
    model = EmbeddingMovingBessKGE(ns, fn, SampledSoftmaxCrossEntropyLoss(n_entity), return_scores=True, augment_negative=True)
    rng = np.random.default_rng(2)
    batch = dict(head=rng.integers(n_entity, size=(1, 1, S)), relation=rng.integers(n_rel, size=(1, 1, S)),
                 tail=rng.integers(n_entity, size=(1, 1, S)), negative=rng.integers(n_entity, size=(1, 1, 1, K)))
    batch = {k: torch.from_numpy(v.astype(np.int32)) for k, v in batch.items()}
    spec = kge.StepSpec("TransE", 1, True, "t", True, augment=True)
    t0, r0 = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
    import contextlib
    with (contextlib.nullcontext() if fp32_math else kge.half_queries()):
        want = kge.bess_step(spec, "EmbeddingMoving", t0, r0, batch, dict(kind="ssce", n_entity=n_entity))
        want["loss"][0].backward()
    lr = 0.05
    runner = runtime.training_model(model, optimizer=runtime.SGD(lr=lr), device=dev, dtype=torch.float16)
    res = runner(**batch)
    got = model.score_fn.entity_embedding.detach().float().cpu()[0]
    want_ent = (ent - lr * t0.grad).half().float()[0]
    ulp = torch.exp2(torch.floor(torch.log2(want_ent.abs().clamp(min=2.0 ** -14))) - 10)
    err = ((got - want_ent).abs() / ulp)
    rows = (err > 1.001).any(-1).nonzero().flatten()
    h, t, ng = set(batch["head"].flatten().tolist()), set(batch["tail"].flatten().tolist()), set(batch["negative"].flatten().tolist())
    print(f"fused_qt={fused_qt} fp32_math={fp32_math}: loss {float(res['loss']):.6f} vs {float(want['loss'][0]):.6f}; rows off: {len(rows)} max {float(err.max()):.1f} ulp")
    for r in rows[:8].tolist():
        e = err[r]
        print("   row", r, "in heads" if r in h else "", "in tails" if r in t else "", "in negs" if r in ng else "",
              "n_bad", int((e > 1.001).sum()), "max", float(e.max()), "grad norm", float(t0.grad[0, r].abs().sum()))
    got_rel = model.score_fn.relation_embedding.detach().float().cpu()
    want_rel = (rel - lr * r0.grad).half().float()
    print("   rel max abs err", float((got_rel - want_rel).abs().max()))
