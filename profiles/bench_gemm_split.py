#!/usr/bin/env python3
"""Split-fp16 MFMA forward GEMM (csrc/gemm_split.hip) vs the exact fp32 MFMA kernel and
vs a float64 product: accuracy and rate on the shared-negative / all-entities shapes.

    python profiles/bench_gemm_split.py            # split kernel (default dispatch)
    python profiles/bench_gemm_split.py fp32   # exact fp32 MFMA kernels (descriptor flag BESS_FLAG_FP32_MATH)
"""

import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "bess-kge_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from besskge import _native as nat  # noqa: E402
from besskge._native import RowSource  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def run(name, dtype, M, W, S, N, use_idx=True, scale=0.1, check=True, bwd=False):
    g = torch.Generator(device="cpu").manual_seed(0)
    table = (torch.randn(M, W, generator=g) * scale).to(dtype).to(dev)
    d = nat.make_desc(nat.DISTMULT, 0, table, W)
    if "fp32" in sys.argv[1:]:
        d.reserved[0] = nat.FLAG_FP32_MATH
    q = (torch.randn(S, W, generator=g) * scale).to(dev)
    idx = torch.randint(M, (N,), dtype=torch.int32, device=dev) if use_idx else None
    neg = RowSource(table, idx) if use_idx else RowSource(table[:N], None)
    out = nat.neg_score_shared_fwd(d, q, neg)
    err = ""
    if check:
        rows = table[idx.long()] if use_idx else table[:N]
        sq = min(S, 512)
        ref = q[:sq].double() @ rows.double().T
        f32 = q[:sq] @ rows.float().T
        den = ref.abs().max().item()
        err = (f" max|err|/max|ref| split {((out[:sq].double() - ref).abs().max().item() / den):.2e}"
               f"  torch-f32 {((f32.double() - ref).abs().max().item() / den):.2e}")
    t = timeit(lambda: nat.neg_score_shared_fwd(d, q, neg))
    flops = 2.0 * S * N * W
    print(f"{name:36s} S={S:6d} N={N:7d} W={W:4d} {str(dtype)[6:]:7s} {t*1e3:9.1f} us {flops/t/1e9:7.1f} TFLOP/s{err}",
          flush=True)
    if bwd:
        go = torch.randn(S, N, generator=g).to(dev)
        dq, dn = nat.neg_score_shared_bwd(d, q, neg, out, go)
        rows = (table[idx.long()] if use_idx else table[:N]).double()
        sq = min(S, 512)
        rq = go[:sq].double() @ rows
        rn = go.double().T[:sq] @ q.double()
        e1 = (dq[:sq].double() - rq).abs().max().item() / rq.abs().max().item()
        e2 = (dn[:sq].double() - rn).abs().max().item() / rn.abs().max().item()
        t = timeit(lambda: nat.neg_score_shared_bwd(d, q, neg, out, go))
        print(f"{'  backward (d_query, d_neg)':36s} {'':34s} {t*1e3:9.1f} us {2*flops/t/1e9:7.1f} TFLOP/s"
              f" max|err|/max|ref| d_query {e1:.2e} d_neg {e2:.2e}", flush=True)


if __name__ == "__main__":
    print("fp32 kernels only:", "fp32" in sys.argv[1:])
    run("C2 ComplEx shared 4096x4096", torch.float32, 93_773, 512, 4096, 4096, bwd=True)
    run("C5 DistMult shared 8192x4096", torch.float32, 1_000_000, 512, 8192, 4096, bwd=True)
    run("ragged 4099 x 5001, W=500", torch.float32, 20_000, 500, 4099, 5001, bwd=True)
    run("tiny values (1e-4)", torch.float32, 20_000, 256, 2048, 4096, scale=1e-4)
    run("large values (100)", torch.float32, 20_000, 256, 2048, 4096, scale=100.0)
    run("fp16 table 4096x4096 W=256", torch.float16, 312_576, 256, 4096, 4096, bwd=True)
    run("YAGO3-10 all entities d=128", torch.float32, 123_182, 256, 5000, 123_182, use_idx=False)
    run("wikikg2-like all entities W=512", torch.float32, 1_000_000, 512, 4096, 1_000_000, use_idx=False, check=False)
