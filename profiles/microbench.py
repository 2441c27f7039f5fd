#!/usr/bin/env python3
"""Per-kernel micro-benchmarks (GPU box): times each hot entry point of the C ABI
on the BASELINE shapes and prints achieved GB/s / TFLOP/s.  Not the headline
bench (that is bench.py); used to steer kernel work.

    python profiles/microbench.py [filter]
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
SC = dict(TransE=nat.TRANSE, RotatE=nat.ROTATE, DistMult=nat.DISTMULT, ComplEx=nat.COMPLEX)


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


def desc(scorer, p, table, W, Wr):
    return nat.make_desc(SC[scorer], p, table, Wr)


def run(name, scorer, p, dtype, M, W, Wr, S, N, shared, fp32_math=False):
    g = torch.Generator(device="cpu").manual_seed(0)
    table = (torch.randn(M, W, generator=g) * 0.1).to(dtype).to(dev)
    d = desc(scorer, p, table, W, Wr)
    if fp32_math:
        d.reserved[0] = nat.FLAG_FP32_MATH
    q = torch.randn(S, W, device=dev)
    sz = table.element_size()
    if shared:
        idx = torch.randint(M, (N,), dtype=torch.int32, device=dev)
        neg = RowSource(table, idx)
        out = nat.neg_score_shared_fwd(d, q, neg)
        go = torch.randn_like(out)
        t_f = timeit(lambda: nat.neg_score_shared_fwd(d, q, neg))
        t_b = timeit(lambda: nat.neg_score_shared_bwd(d, q, neg, out, go))
        ops = (2 if nat.reduce_of(d) == 0 else 3) * S * N * W if hasattr(nat, "reduce_of") else 2 * S * N * W
        flops = 2.0 * S * N * W
        extra = ""
        if scorer in ("TransE", "RotatE"):  # VALU kernels: elements per second (one element = one (q, j, w))
            extra = f" | {S*N*W/t_f/1e9:6.1f} / {2*S*N*W/t_b/1e9:6.1f} T elements/s fwd / bwd (2 products)"
        print(f"{name:34s} shared  S={S:5d} N={N:5d} W={W:4d} {str(dtype)[6:]:7s} fwd {t_f*1e3:8.1f} us "
              f"{flops/t_f/1e9:7.1f} TFLOP/s(2SNW) | bwd {t_b*1e3:8.1f} us {2*flops/t_b/1e9:7.1f} TFLOP/s(4SNW){extra}")
    else:
        idx = torch.randint(M, (S * N,), dtype=torch.int32, device=dev)
        neg = RowSource(table, idx)
        out = nat.neg_score_pertriple_fwd(d, q, neg, N)
        go = torch.randn_like(out)
        t_f = timeit(lambda: nat.neg_score_pertriple_fwd(d, q, neg, N))
        t_b = timeit(lambda: nat.neg_score_pertriple_bwd(d, q, neg, N, go), reps=5)
        bf = S * N * (W * sz + 8) + S * W * 4
        bb = S * N * (W * sz + W * 4 + 8) + 2 * S * W * 4
        print(f"{name:34s} per-tri S={S:5d} N={N:5d} W={W:4d} {str(dtype)[6:]:7s} fwd {t_f*1e3:8.1f} us "
              f"{bf/t_f/1e6:7.0f} GB/s | bwd {t_b*1e3:8.1f} us {bb/t_b/1e6:7.0f} GB/s")


CASES = [
    # name, scorer, p, dtype, M, W, Wr, S, N, shared
    ("C2 ComplEx d256 f32 (L3)", "ComplEx", 0, torch.float32, 93_773, 512, 512, 4096, 256, False),
    ("C2 ComplEx d256 f32 (HBM 8GB)", "ComplEx", 0, torch.float32, 4_000_000, 512, 512, 4096, 256, False),
    ("C5 DistMult d512 f32 (HBM 8GB)", "DistMult", 0, torch.float32, 4_000_000, 512, 512, 8192, 64, False),
    ("C3 RotatE d200 f32 p1", "RotatE", 1, torch.float32, 61_591, 400, 200, 4096, 256, False),
    ("C4 TransE d256 f16 p1 (L3)", "TransE", 1, torch.float16, 312_576, 256, 256, 4096, 256, False),
    ("C4 TransE d256 f16 p1 (HBM 4GB)", "TransE", 1, torch.float16, 8_000_000, 256, 256, 4096, 256, False),
    ("C1 TransE d128 f32 p1", "TransE", 1, torch.float32, 10_000, 128, 128, 4096, 256, False),
    # the per-GPU launch of bench.py --gpus N (ScoreMoving: N*S gathered queries x 256/N local negatives)
    ("scale N=2 ComplEx d256 f32", "ComplEx", 0, torch.float32, 93_773, 512, 512, 8192, 128, False),
    ("scale N=4 ComplEx d256 f32", "ComplEx", 0, torch.float32, 93_773, 512, 512, 16384, 64, False),
    ("scale N=8 ComplEx d256 f32", "ComplEx", 0, torch.float32, 93_773, 512, 512, 32768, 32, False),
    ("C2 ComplEx shared 4096x4096", "ComplEx", 0, torch.float32, 93_773, 512, 512, 4096, 4096, True),
    ("C5 DistMult shared 8192x4096", "DistMult", 0, torch.float32, 1_000_000, 512, 512, 8192, 4096, True),
    ("C4 TransE f16 L1 shared 4096x4096", "TransE", 1, torch.float16, 312_576, 256, 256, 4096, 4096, True),
    ("C4 TransE f16 L1 shared 512x768", "TransE", 1, torch.float16, 312_576, 256, 256, 512, 768, True),
    ("C4 TransE f16 L1 shared 4096x4352", "TransE", 1, torch.float16, 312_576, 256, 256, 4096, 4352, True),
    ("C4 fp32-math L1 shared 4096x4096", "TransE", 1, torch.float16, 312_576, 256, 256, 4096, 4096, True, True),
    ("C4 fp32-math L1 shared 512x768", "TransE", 1, torch.float16, 312_576, 256, 256, 512, 768, True, True),
    ("C3 RotatE L1 shared 4096x4096", "RotatE", 1, torch.float32, 61_591, 400, 200, 4096, 4096, True),
    ("C1 TransE L2 shared 512x64", "TransE", 2, torch.float32, 10_000, 128, 128, 512, 64, True),
]

def run_affine(name, n_part, normalize, p, dtype, M, d, S, N, shared):
    """PairRE / TripleRE (n_part 1) and InterHT / TranS (n_part 2) kernel family."""
    g = torch.Generator(device="cpu").manual_seed(0)
    W = n_part * d
    table = (torch.randn(M, W, generator=g) * 0.1).to(dtype).to(dev)
    dsc = nat.make_desc(nat.AFFINE, p, table, d)
    dsc.reserved[0], dsc.reserved[1] = n_part, int(normalize)
    q = torch.randn(S, (n_part + 1) * d, device=dev)
    sz = table.element_size()
    if shared:
        neg = RowSource(table, torch.randint(M, (N,), dtype=torch.int32, device=dev))
        out = nat.neg_score_shared_fwd(dsc, q, neg)
        go = torch.randn_like(out)
        t_f = timeit(lambda: nat.neg_score_shared_fwd(dsc, q, neg))
        t_b = timeit(lambda: nat.neg_score_shared_bwd(dsc, q, neg, out, go))
        ops_f = (n_part + 1) * S * N * d  # fma per part + |.| accumulate
        print(f"{name:34s} shared  S={S:5d} N={N:5d} d={d:4d} {str(dtype)[6:]:7s} fwd {t_f*1e3:8.1f} us "
              f"{ops_f/t_f/1e9:6.1f} T lane-ops/s | bwd {t_b*1e3:8.1f} us")
    else:
        neg = RowSource(table, torch.randint(M, (S * N,), dtype=torch.int32, device=dev))
        out = nat.neg_score_pertriple_fwd(dsc, q, neg, N)
        go = torch.randn_like(out)
        t_f = timeit(lambda: nat.neg_score_pertriple_fwd(dsc, q, neg, N))
        t_b = timeit(lambda: nat.neg_score_pertriple_bwd(dsc, q, neg, N, go), reps=5)
        bf = S * N * (W * sz + 8) + S * q.shape[1] * 4
        bb = S * N * (W * sz + W * 4 + 8) + 2 * S * q.shape[1] * 4
        print(f"{name:34s} per-tri S={S:5d} N={N:5d} d={d:4d} {str(dtype)[6:]:7s} fwd {t_f*1e3:8.1f} us "
              f"{bf/t_f/1e6:7.0f} GB/s | bwd {t_b*1e3:8.1f} us {bb/t_b/1e6:7.0f} GB/s")


def run_movers():
    """K1 gather and K9/K10 sparse update on config-2-like rows (2 KiB) and config-4-like (512 B)."""
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, M, W, dtype, n in (("f32 2 KiB rows (L3)", 93_773, 512, torch.float32, 1 << 20),
                                 ("f32 2 KiB rows (HBM 8GB)", 4_000_000, 512, torch.float32, 1 << 20),
                                 ("f16 512 B rows (HBM 4GB)", 8_000_000, 256, torch.float16, 1 << 20)):
        table = torch.zeros(M, W, dtype=dtype, device=dev)
        idx = torch.randint(M, (n,), generator=g, dtype=torch.int32).to(dev)
        out = nat.gather_rows(table, idx)
        t_g = timeit(lambda: nat.gather_rows(table, idx, out))
        grad = torch.randn(n, W, device=dev)
        t_s = timeit(lambda: nat.sparse_sgd(table, idx, grad, 1e-3), reps=5)
        sz = table.element_size()
        print(f"movers {name:28s} gather {t_g*1e3:8.1f} us {2*n*W*sz/t_g/1e6:7.0f} GB/s (read+write) | "
              f"atomic sparse SGD {t_s*1e3:8.1f} us {n*W*(4+sz)/t_s/1e6:7.0f} GB/s")


AFFINE_CASES = [
    # name, n_part, normalize, p, dtype, M, d, S, N, shared
    ("PairRE d256 f32 norm (L3)", 1, True, 1, torch.float32, 93_773, 256, 4096, 256, False),
    ("PairRE d512 f32 norm (L3)", 1, True, 1, torch.float32, 93_773, 512, 4096, 256, False),
    ("TranS d256 f32 norm (L3)", 2, True, 1, torch.float32, 93_773, 256, 4096, 256, False),
    ("TranS d256 f32 norm (HBM 8GB)", 2, True, 1, torch.float32, 4_000_000, 256, 4096, 256, False),
    ("InterHT d128 f16 (L3)", 2, True, 2, torch.float16, 312_576, 128, 4096, 256, False),
    ("PairRE d256 shared 4096x4096", 1, True, 1, torch.float32, 93_773, 256, 4096, 4096, True),
    ("TranS d256 shared 4096x4096", 2, True, 1, torch.float32, 93_773, 256, 4096, 4096, True),
]

if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    print(torch.cuda.get_device_name(0))
    for c in CASES:
        if flt in c[0]:
            run(*c)
    for c in AFFINE_CASES:
        if flt in c[0]:
            run_affine(*c)
    if flt in "movers":
        run_movers()
