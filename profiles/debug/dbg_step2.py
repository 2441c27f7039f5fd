import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import numpy as np, torch
from besskge import runtime, _native as nat
import besskge.bess as bess_mod
from besskge.bess import EmbeddingMovingBessKGE
from besskge.loss import SampledSoftmaxCrossEntropyLoss
from besskge.negative_sampler import RandomShardedNegativeSampler
from besskge.scoring import TransE
from besskge.sharding import Sharding

dev = torch.device("cuda", 0)
n_entity, n_rel, d, S, K = 3000, 11, 64, 128, 96
torch.manual_seed(0)
sharding = Sharding.create(n_entity, 1, seed=3)
ent = (torch.randn(1, sharding.max_entity_per_shard, d) * 0.5).half().float()
rel = (torch.randn(n_rel, d) * 0.5).half().float()
fn = TransE(True, 1, sharding, n_rel, d, ent, rel)
ns = RandomShardedNegativeSampler(K, sharding, 5, "t", local_sampling=False, flat_negative_format=True)
model = EmbeddingMovingBessKGE(ns, fn, SampledSoftmaxCrossEntropyLoss(n_entity), return_scores=True, augment_negative=True)
rng = np.random.default_rng(2)
batch = dict(head=rng.integers(n_entity, size=(1, 1, S)), relation=rng.integers(n_rel, size=(1, 1, S)),
             tail=rng.integers(n_entity, size=(1, 1, S)), negative=rng.integers(n_entity, size=(1, 1, 1, K)))
batch = {k: torch.from_numpy(v.astype(np.int32)) for k, v in batch.items()}
rec = {}
orig = nat.neg_score_shared_bwd
def spy(dsc, query, neg, out, d_out):
    dq, dn = orig(dsc, query, neg, out, d_out)
    torch.cuda.synchronize()
    rec.update(q=query.cpu(), idx=neg.idx.cpu(), base=neg.base.float().cpu(), out=out.cpu(), go=d_out.cpu(), dq=dq.cpu(), dn=dn.cpu(),
               flags=int(dsc.reserved[0]), strides=(query.stride(), d_out.stride(), out.stride()))
    return dq, dn
nat.neg_score_shared_bwd = spy
runner = runtime.training_model(model, optimizer=runtime.SGD(lr=0.05), device=dev, dtype=torch.float16)
res = runner(**batch)
q16 = rec["q"].half().double(); E = rec["base"][rec["idx"].long()].double(); c = rec["go"].double()
print("strides", rec["strides"], "flags", rec["flags"], "shapes", tuple(rec["q"].shape), tuple(rec["go"].shape))
sg = torch.sign(q16[:, None, :] - E[None])
dq_w = -(c[:, :, None] * sg).sum(1); dE_w = (c[:, :, None] * sg).sum(0)
print("dq err", float((rec["dq"].double() - dq_w).abs().max()), "dn err", float((rec["dn"].double() - dE_w).abs().max()))
sc_w = -(q16[:, None, :] - E[None]).abs().sum(-1)
so = rec["out"].double(); live = so > -40000
print("score err", float((so - sc_w).abs()[live].max()), "killed", int((~live).sum()), "diag killed", bool((~live).diagonal().all()))
print("c at killed max", float(c[~live].abs().max()), "c max", float(c.max()))
