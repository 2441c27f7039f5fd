#!/usr/bin/env python3
"""Stress of the shared-negative distance kernels (csrc/neg_shared.hip, csrc/l1_f16.hip): random shapes pick every
tile variant (32 / 64-row tiles forward and backward), every reduction split the planner makes and the one-launch
backward; p = 1 and 2, fp32 and fp16 tables, ragged rows and columns, widths that are not a multiple of 4, ties.
Against float64 torch on the same device (the query is fp16-exact, so the packed-fp16 path needs no special case)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import torch
from besskge import _native as nat
from besskge._native import RowSource

dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(11)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 120
worst = 0.0
for it in range(n_iter):
    p = 1 + it % 2
    big = it % 6 == 0
    S = int(torch.randint(1, 1500 if big else 400, (1,), generator=gen))
    N = int(torch.randint(1, 2500 if big else 600, (1,), generator=gen))
    W = int(torch.randint(1, 130, (1,), generator=gen)) * (4 if it % 5 else 1)
    if it % 7 == 0:
        W = (W + 31) // 32 * 32  # the packed-fp16 forward's widths
    M = int(torch.randint(10, 3000, (1,), generator=gen))
    dtype = torch.float16 if it % 3 == 0 else torch.float32
    table = (torch.randn(M, W, generator=gen) * 0.5).to(dtype).to(dev)
    q = (torch.randn(S, W, generator=gen) * 0.5).half().float()
    idx = torch.randint(M, (N,), generator=gen, dtype=torch.int32).to(dev)
    if it % 4 == 0:  # ties: some query entries equal the candidate they meet
        rows0 = table[idx.long()[torch.arange(S, device=dev) % N]].float().cpu()
        q = torch.where(torch.rand(S, W, generator=gen) < 0.2, rows0, q)
    q = q.to(dev)
    go = (torch.randn(S, N, generator=gen)).to(dev)
    desc = nat.make_desc(nat.TRANSE, p, table, W)
    src = RowSource(table, idx)
    out = nat.neg_score_shared_fwd(desc, q, src)
    dq, dn = nat.neg_score_shared_bwd(desc, q, src, out, go)
    rows = table[idx.long()].double()
    qd = q.double()
    ref = torch.empty(S, N, dtype=torch.float64, device=dev)
    wq = torch.zeros(S, W, dtype=torch.float64, device=dev)
    wn = torch.zeros(N, W, dtype=torch.float64, device=dev)
    step = max(1, (1 << 24) // max(1, N * W))
    for a0 in range(0, S, step):
        diff = qd[a0:a0 + step, None, :] - rows[None, :, :]
        if p == 1:
            ref[a0:a0 + step] = -diff.abs().sum(-1)
            coef = torch.sign(diff)
        else:
            nrm = diff.norm(dim=-1, keepdim=True)
            ref[a0:a0 + step] = -nrm[..., 0]
            coef = torch.where(nrm > 0, diff / nrm.clamp(min=1e-300), torch.zeros_like(diff))
        t = go[a0:a0 + step].double()[:, :, None] * coef
        wq[a0:a0 + step] = -t.sum(1)
        wn += t.sum(0)
    e = [float((out.double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-20),
         float((dq.double() - wq).abs().max()) / (float(wq.abs().max()) + 1e-20),
         float((dn.double() - wn).abs().max()) / (float(wn.abs().max()) + 1e-20)]
    worst = max(worst, *e)
    if max(e) > 3e-5:
        print(f"iteration {it}: p={p} W={W} S={S} N={N} {dtype}: errors {e}  FAIL")
        sys.exit(1)
print(f"{n_iter} problems: worst error / max|exact| over scores, d_query, d_neg: {worst:.2e}")
