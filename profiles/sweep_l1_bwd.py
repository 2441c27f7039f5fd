"""k_l1_bwd_both (p = 1 shared-negative backward, both products in one launch) from notebook-size to C4-size
micro-batches: kernel time alone - back-to-back launches recorded in a hipGraph, outputs cleared by the caller -
which profiles/microbench.py cannot show below ~1024 queries (its eager calls are launch-bound there)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "bess-kge_amd"))
from besskge import _native as nat
from besskge._native import RowSource
dev = torch.device("cuda:0")
W = 256
table = (torch.randn(100_000, W) * 0.1).half().to(dev)
dsc = nat.make_desc(nat.TRANSE, 1, table, W)

def time_graph(S, N, reps=20):
    q = torch.randn(S, W, device=dev)
    neg = RowSource(table, torch.randint(100_000, (N,), dtype=torch.int32, device=dev))
    out = nat.neg_score_shared_fwd(dsc, q, neg)
    go = torch.randn_like(out)
    buf = nat.shared_bwd_buffer(dsc, S, N, dev)
    buf.zero_()
    nat.neg_score_shared_bwd(dsc, q, neg, out, go, prezeroed=buf)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            nat.neg_score_shared_bwd(dsc, q, neg, out, go, prezeroed=buf)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best

def time_parts(S, N, reps=20):
    """The same two products as partial sums (bess_neg_score_shared_bwd_parts: independent waves, plain stores)."""
    q = torch.randn(S, W, device=dev)
    neg = RowSource(table, torch.randint(100_000, (N,), dtype=torch.int32, device=dev))
    go = torch.randn(S, N, device=dev)
    nat.neg_score_shared_bwd_parts(dsc, q, neg, go)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            nat.neg_score_shared_bwd_parts(dsc, q, neg, go)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best

for S, N in ((256, 288), (512, 544), (512, 768), (1024, 1088), (2048, 2176), (4096, 4352), (8192, 8448)):
    r = 5 if S > 1024 else 20
    parts_us = time_parts(S, N, reps=r) if nat.shared_bwd_parts_plan(dsc, S, N)[0] > 0 else float("nan")  # (capped at 24 MB of slabs)
    print(f"S={S:5d} N={N:5d}: atomic sums {time_graph(S, N, reps=r):8.1f} us   partial sums {parts_us:8.1f} us "
          f"(parts {nat.shared_bwd_parts_plan(dsc, S, N)})", flush=True)
