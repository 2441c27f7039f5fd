#!/usr/bin/env python3
"""Segmented K9 under skew: a share of all per-triple negatives points at ONE row (what padded
candidate lists of TripleBasedShardedNegativeSampler do).  Times bess_neg_pertriple_grad_segments alone."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import torch
from besskge import _native as nat
from besskge._native import RowSource

dev = torch.device("cuda", 0)
M, W, S, N = 93_773, 512, 4096, 256
torch.manual_seed(0)
table = torch.randn(M, W, device=dev) * 0.1
q = torch.randn(S, W, device=dev)
d = nat.make_desc(nat.COMPLEX, 0, table, W)
go = torch.randn(S, N, device=dev) * 1e-3
for share in (0.0, 0.01, 0.1, 0.3):
    idx = torch.randint(M, (S * N,), dtype=torch.int32, device=dev)
    hot = torch.rand(S * N, device=dev) < share
    idx[hot] = 7
    seg = nat.SegmentIndex(idx, M)
    def run():
        nat.neg_pertriple_grad_segments(d, q, table, N, go, seg, fused_sgd_lr=1e-6)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    b.record()
    torch.cuda.synchronize()
    print(f"share of references on one row {share:5.2f}: longest segment {int(hot.sum()):8d} refs, "
          f"{a.elapsed_time(b) / 5 * 1e3:10.1f} us per pass", flush=True)
