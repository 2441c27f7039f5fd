"""Eager launches of the p = 1 shared-negative kernels (k_l1_fwd_pk, k_l1_bwd_both, k_l1_bwd_parts) at three
micro-batch sizes, for `rocprofv3 --pmc` (profiles/pmc_l1_r04.sh): no graphs, a few launches per size."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "bess-kge_amd"))
from besskge import _native as nat
from besskge._native import RowSource
dev = torch.device("cuda:0")
W = 256
table = (torch.randn(100_000, W) * 0.1).half().to(dev)
dsc = nat.make_desc(nat.TRANSE, 1, table, W)
for S, N in ((512, 544), (2048, 2176), (8192, 8448)):
    q = torch.randn(S, W, device=dev)
    neg = RowSource(table, torch.randint(100_000, (N,), dtype=torch.int32, device=dev))
    go = torch.randn(S, N, device=dev)
    buf = nat.shared_bwd_buffer(dsc, S, N, dev)
    for _ in range(4):
        out = nat.neg_score_shared_fwd(dsc, q, neg)
        buf.zero_()
        nat.neg_score_shared_bwd(dsc, q, neg, out, go, prezeroed=buf)
        if nat.shared_bwd_parts_plan(dsc, S, N)[0] > 0:
            nat.neg_score_shared_bwd_parts(dsc, q, neg, go)
    torch.cuda.synchronize()
    print(S, N, "done", flush=True)
