import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO, os.path.join(REPO, "tests")]
import torch
from besskge import _native as nat

dev = torch.device("cuda", 0)
calls = []
of, ob = nat.neg_score_shared_fwd, nat.neg_score_shared_bwd
def sf(dsc, query, neg, pad_ld=False, kill=None):
    out = of(dsc, query, neg, pad_ld, kill)
    torch.cuda.synchronize()
    q16 = query.cpu().half().double(); E = neg.base.float().cpu()[neg.idx.cpu().long()].double()
    want = -(q16[:, None, :] - E[None]).abs().sum(-1)
    o = out.cpu().double(); live = o > -40000
    print(f"fwd call: Q{tuple(query.shape)} N={len(neg)} kill={None if kill is None else kill[:3]} max err live {float((o - want)[live].abs().max()):.3e} n_killed {int((~live).sum())}", flush=True)
    return out
def sb(dsc, query, neg, out, d_out):
    dq, dn = ob(dsc, query, neg, out, d_out)
    torch.cuda.synchronize()
    q16 = query.cpu().half().double(); E = neg.base.float().cpu()[neg.idx.cpu().long()].double(); c = d_out.cpu().double()
    sg = torch.sign(q16[:, None, :] - E[None])
    dq_w = -(c[:, :, None] * sg).sum(1); dE_w = (c[:, :, None] * sg).sum(0)
    eq, ee = (dq.cpu().double() - dq_w).abs(), (dn.cpu().double() - dE_w).abs()
    print(f"bwd call: dq err {float(eq.max()):.3e} ({int((eq > 1e-3).sum())} bad) dn err {float(ee.max()):.3e} ({int((ee > 1e-3).sum())} bad) cmax {float(c.max()):.3f}", flush=True)
    return dq, dn
nat.neg_score_shared_fwd, nat.neg_score_shared_bwd = sf, sb
from test_baseline_configs import run_config
for n in (1, 2, 8):
    print("=== n_shard", n, flush=True)
    try:
        run_config(dev, "TransE", 1, 256, torch.float16, n, 40_000, 535, 200_000, 512, 32, "t", True, True, True, "ssce")
        print("PASS")
    except AssertionError as e:
        print("FAIL", str(e)[:200])
