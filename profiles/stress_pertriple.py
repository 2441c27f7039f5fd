#!/usr/bin/env python3
"""Stress of the dominant kernel (per-triple negatives, csrc/neg_pertriple.hip) and its backward:
random widths (incl. odd, and rows wider than the 1024 / 2048 scalars a lane group holds), table dtypes, query /
negative counts against torch on the same device."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import torch
from besskge import _native as nat
from besskge._native import RowSource

dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(6)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 150
worst = 0.0
for it in range(n_iter):
    red = it % 3  # 0 dot (DistMult), 1 L1, 2 L2 (TransE)
    W = int(torch.randint(1, 600, (1,), generator=gen))
    if it % 4:
        W = (W + 3) // 4 * 4
    else:
        W = W % 256 + 1  # rows that are not a multiple of 4 scalars are read scalar-wise: up to 256 scalars
    if it % 7 == 3 and red != 2:
        # rows wider than a lane group's registers: scored and back-propagated in column windows
        W = 1024 + 4 * int(torch.randint(1, 500, (1,), generator=gen))
    S = int(torch.randint(1, 300, (1,), generator=gen))
    N = int(torch.randint(1, 500 if W <= 1024 else 60, (1,), generator=gen))
    M = int(torch.randint(10, 3000, (1,), generator=gen))
    dtype = torch.float16 if it % 5 == 0 else torch.float32
    if dtype == torch.float16 and W % 8 and W > 256:
        W = W // 8 * 8  # wide fp16 rows: multiples of 8 scalars (include/besskge_hip.h, K5)
    table = torch.randn(M, W, generator=gen).to(dtype).to(dev)
    q = torch.randn(S, W, generator=gen).to(dev)
    idx = torch.randint(M, (S * N,), generator=gen, dtype=torch.int32).to(dev)
    desc = nat.make_desc(nat.DISTMULT if red == 0 else nat.TRANSE, max(red, 1), table, W)
    out = nat.neg_score_pertriple_fwd(desc, q, RowSource(table, idx), N)
    rows = table[idx.long()].double().view(S, N, W)
    qq = q.double()[:, None, :]
    qq.requires_grad_(True)
    rows.requires_grad_(True)
    if red == 0:
        ref = (qq * rows).sum(-1)
    elif red == 1:
        ref = -(qq - rows).abs().sum(-1)
    else:
        ref = -((qq - rows) ** 2).sum(-1).sqrt()
    go = torch.randn(S, N, generator=gen).to(dev)
    ref.backward(go.double())
    dq, dn = nat.neg_score_pertriple_bwd(desc, q, RowSource(table, idx), N, go)
    ref = ref.detach()
    e = [float((out.double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-20),
         float((dq.double() - qq.grad[:, 0]).abs().max()) / (float(qq.grad.abs().max()) + 1e-20),
         float((dn.double().view(S, N, W) - rows.grad).abs().max()) / (float(rows.grad.abs().max()) + 1e-20)]
    worst = max(worst, *e)
    if max(e) > 2e-5:
        print(f"iteration {it}: red {red} W={W} S={S} N={N} {dtype}: errors {e}  FAIL")
        sys.exit(1)
print(f"{n_iter} problems: worst error / max|exact| over scores, d_query, d_neg: {worst:.2e}")
