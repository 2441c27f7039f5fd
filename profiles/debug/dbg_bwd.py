import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import numpy as np, torch
from besskge import _native as nat
from besskge.sharding import Sharding

dev = torch.device("cuda", 0)
torch.manual_seed(0)
n_entity, n_rel, d, S, K = 3000, 11, 64, 128, 96
sharding = Sharding.create(n_entity, 1, seed=3)
ent = (torch.randn(1, sharding.max_entity_per_shard, d) * 0.5).half().float()
rel = (torch.randn(n_rel, d) * 0.5).half().float()
rng = np.random.default_rng(2)
batch = dict(head=rng.integers(n_entity, size=(1, 1, S)), relation=rng.integers(n_rel, size=(1, 1, S)),
             tail=rng.integers(n_entity, size=(1, 1, S)), negative=rng.integers(n_entity, size=(1, 1, 1, K)))
batch = {k: torch.from_numpy(v.astype(np.int32)) for k, v in batch.items()}
T = ent[0]
h = T[batch["head"].flatten().long()]; r = rel[batch["relation"].flatten().long()]
cand_idx = torch.cat([batch["tail"].flatten(), batch["negative"].flatten()]).to(torch.int32)
E = T[cand_idx.long()].double()
q = h + r
q16 = q.half().double()
sc = -(q16[:, None, :] - E[None]).abs().sum(-1)
sc[torch.arange(S), torch.arange(S)] += -50000.0
c = torch.softmax(sc, 1).float()
sg = torch.sign(q16[:, None, :] - E[None])
dq_w = -(c.double()[:, :, None] * sg).sum(1)
dE_w = (c.double()[:, :, None] * sg).sum(0)
dsc = nat.ModelDesc(); dsc.scorer, dsc.norm_p, dsc.dtype, dsc.width, dsc.rel_width = nat.TRANSE, 1, nat.F16, d, d
table = T.half().to(dev)
src = nat.RowSource(table, cand_idx.to(dev))
out = nat.neg_score_shared_fwd(dsc, q.to(dev), src)
print("fwd max err", float((out.cpu().double() - (sc + 0)).abs()[sc > -40000].max()))
dq, dn = nat.neg_score_shared_bwd(dsc, q.to(dev), src, out, c.to(dev))
eq, ee = (dq.cpu().double() - dq_w).abs(), (dn.cpu().double() - dE_w).abs()
print("dq max err", float(eq.max()), "dE max err", float(ee.max()))
bad = (eq > 1e-3).nonzero()
print("bad dq elements", len(bad))
for a, w in bad[:6].tolist():
    d_ = q16[a, w] - E[:, w]
    order = torch.argsort(c[a], descending=True)[:4]
    print(f"  a={a} w={w} got {float(dq[a, w]):.5f} want {float(dq_w[a, w]):.5f}; rowmax c {float(c[a].max()):.4f};"
          f" top c {[round(float(c[a, b]), 4) for b in order]} d at top {[float(d_[b]) for b in order]} q16 {float(q16[a, w])} q {float(q[a, w])!r}")
bad = (ee > 1e-3).nonzero()
print("bad dE elements", len(bad))
for b, w in bad[:6].tolist():
    d_ = q16[:, w] - E[b, w]
    order = torch.argsort(c[:, b], descending=True)[:4]
    print(f"  b={b} w={w} got {float(dn[b, w]):.5f} want {float(dE_w[b, w]):.5f}; colmax {float(c[:, b].max()):.4f};"
          f" top c {[round(float(c[a, b]), 4) for a in order]} d at top {[float(d_[a]) for a in order]}")
