#!/usr/bin/env python3
"""Stress of the fused training forward (scores + d loss / d query in one pass, csrc/neg_pertriple.hip):
random sizes (so that the split of a query's negatives into work items varies), scorers, losses and
table dtypes against the two-pass path (forward, loss kernel, backward for d_query)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import torch
from besskge import _native as nat
from besskge._native import RowSource
from besskge.loss import LogSigmoidLoss, MarginRankingLoss, SampledSoftmaxCrossEntropyLoss

dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(4)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 120
worst = 0.0
for it in range(n_iter):
    scorer = [nat.TRANSE, nat.ROTATE, nat.DISTMULT, nat.COMPLEX][it % 4]
    d = int(torch.randint(1, 65, (1,), generator=gen)) * (4 if it % 5 else 1)
    W = 2 * d if scorer in (nat.COMPLEX, nat.ROTATE) else d
    Wr = d if scorer == nat.ROTATE else W
    S = int(torch.randint(1, 600, (1,), generator=gen))
    N = int(torch.randint(1, 700, (1,), generator=gen))
    M = int(torch.randint(50, 5000, (1,), generator=gen))
    dtype = torch.float16 if it % 3 == 0 else torch.float32
    loss = [LogSigmoidLoss(6.0, True, 0.5), LogSigmoidLoss(1.0, False), MarginRankingLoss(2.0, True, 1.0),
            SampledSoftmaxCrossEntropyLoss(100000)][it % 4 if it % 7 else (it // 7) % 4]
    table = (0.5 * torch.randn(M, W, generator=gen)).to(dtype).to(dev)
    q = (0.5 * torch.randn(S, W, generator=gen)).to(dev)
    idx = torch.randint(M, (S * N,), generator=gen, dtype=torch.int32).to(dev)
    pos = torch.randn(S, generator=gen).to(dev)
    w = (torch.rand(S, generator=gen) + 0.5).to(dev) if it % 2 else torch.ones(1, device=dev)
    desc = nat.make_desc(scorer, 1 + it % 2, table, Wr)
    ld = loss.kernel_desc(N)
    neg = RowSource(table, idx)
    out, dq = nat.neg_score_pertriple_fwd_dq(desc, ld, q, neg, N, pos, w)
    ref = nat.neg_score_pertriple_fwd(desc, q, neg, N)
    _, _, dn = nat.loss_fwd_bwd(ld, pos, ref, w, True)
    dq_ref, _ = nat.neg_score_pertriple_bwd(desc, q, neg, N, dn, want_d_neg=False)
    scale = float(dq_ref.abs().max()) + 1e-20
    err = float((dq - dq_ref).abs().max()) / scale
    worst = max(worst, err)
    if not torch.equal(out, ref) or err > 2e-4:
        print(f"iteration {it}: scorer {scorer} W={W} S={S} N={N} {dtype}: scores equal {torch.equal(out, ref)}, d_query error {err:.2e}  FAIL")
        sys.exit(1)
print(f"{n_iter} problems: scores bit-identical, worst d_query error / max|d_query|: {worst:.2e}")
