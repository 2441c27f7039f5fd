#!/usr/bin/env python3
"""Stress of the split-fp16 products (csrc/gemm_split.hip): random shapes that take the path (both
tile kernels, odd slice counts, ragged edges, fp32 / fp16 tables, forward and backward) against
float64 products."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import torch
from besskge import _native as nat
from besskge._native import RowSource

dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(1)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 60
worst_f = worst_b = 0.0
done = 0
while done < n_iter:
    S = int(torch.randint(130, 5000, (1,), generator=gen))
    N = int(torch.randint(130, 9000, (1,), generator=gen))
    W = int(torch.randint(2, 140, (1,), generator=gen)) * 4
    dtype = torch.float32 if done % 3 else torch.float16
    M = max(N, 4000)
    table = (torch.randn(M, W, generator=gen) * 0.3).to(dtype).to(dev)
    d = nat.make_desc(nat.DISTMULT, 0, table, W)
    if nat.load().bess_neg_score_shared_workspace(ctypes.byref(d), S, N) <= 0:
        continue
    done += 1
    q = (torch.randn(S, W, generator=gen) * 0.3).to(dev)
    idx = torch.randint(M, (N,), generator=gen, dtype=torch.int32).to(dev)
    neg = RowSource(table, idx)
    out = nat.neg_score_shared_fwd(d, q, neg)
    rows = table[idx.long()].double()
    ref = q.double() @ rows.T
    ef = float((out.double() - ref).abs().max()) / float(ref.abs().max())
    worst_f = max(worst_f, ef)
    eb = 0.0
    if nat.load().bess_neg_score_shared_bwd_workspace(ctypes.byref(d), S, N) > 0:
        go = torch.randn(S, N, generator=gen).to(dev)
        dq, dn = nat.neg_score_shared_bwd(d, q, neg, out, go)
        rq, rn = go.double() @ rows, go.double().T @ q.double()
        eb = max(float((dq.double() - rq).abs().max()) / float(rq.abs().max()),
                 float((dn.double() - rn).abs().max()) / float(rn.abs().max()))
        worst_b = max(worst_b, eb)
    if max(ef, eb) > 3e-6:
        print(f"S={S} N={N} W={W} {dtype}: forward {ef:.2e} backward {eb:.2e}  FAIL")
        sys.exit(1)
print(f"{n_iter} shapes, worst error / max|exact|: forward {worst_f:.2e}, backward {worst_b:.2e}")
