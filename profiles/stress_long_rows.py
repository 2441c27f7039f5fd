#!/usr/bin/env python3
"""Stress of the long-row tier of the segmented K9 (partial sums through atomics, last-arriver
write-out): many random problems with several hot rows, fused SGD and gradient-row mode, against
index_add of the per-reference backward.  A lost or doubly counted slice shows as a large error."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import torch
from besskge import _native as nat
from besskge._native import RowSource

dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(0)
worst = 0.0
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for it in range(n_iter):
    M = int(torch.randint(300, 3000, (1,), generator=gen))
    d = int(torch.randint(2, 40, (1,), generator=gen)) * 4
    S = int(torch.randint(50, 400, (1,), generator=gen))
    N = int(torch.randint(16, 200, (1,), generator=gen))
    scorer = [nat.TRANSE, nat.DISTMULT, nat.COMPLEX, nat.ROTATE][it % 4]
    W = 2 * d if scorer in (nat.COMPLEX, nat.ROTATE) else d
    Wr = d if scorer == nat.ROTATE else W
    table = torch.randn(M, W, generator=gen).to(dev)
    q = torch.randn(S, W, generator=gen).to(dev)
    idx = torch.randint(M, (S * N,), generator=gen, dtype=torch.int32)
    r = torch.rand(S * N, generator=gen)
    n_hot = int(torch.randint(1, 5, (1,), generator=gen))
    for h in range(n_hot):
        idx[(r >= 0.15 * h) & (r < 0.15 * h + 0.12)] = int(torch.randint(M, (1,), generator=gen))
    idx = idx.to(dev)
    go = (torch.randn(S, N, generator=gen) * 0.05).to(dev)
    desc = nat.make_desc(scorer, 1 + it % 2, table, Wr)
    _, dn = nat.neg_score_pertriple_bwd(desc, q, RowSource(table, idx), N, go)
    want = torch.zeros(M, W, dtype=torch.float64, device=dev).index_add_(0, idx.long(), dn.double())
    seg = nat.SegmentIndex(idx, M, width=W)
    g = nat.neg_pertriple_grad_segments(desc, q, table, N, go, seg)
    n = int(seg.n_seg.item())
    rows = seg.seg_rows[:n].long()
    scale = float(want.abs().max()) + 1e-12
    e1 = float((g[:n].double() - want[rows]).abs().max()) / scale
    t2 = table.clone()
    nat.neg_pertriple_grad_segments(desc, q, t2, N, go, seg, fused_sgd_lr=0.25)
    e2 = float((t2.double() - (table.double() - 0.25 * want)).abs().max()) / scale
    worst = max(worst, e1, e2)
    if max(e1, e2) > 1e-4:
        print(f"iteration {it}: M={M} W={W} S={S} N={N} long rows {int(seg.long_segs[0])}: errors {e1:.2e} {e2:.2e}  FAIL")
        sys.exit(1)
print(f"{n_iter} problems, worst error relative to the largest gradient entry: {worst:.2e}")
