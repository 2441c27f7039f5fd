"""The S-row tail of a per-triple training step at the C2 micro-batch (S = 4096 triples x 256 negatives, ComplEx
W = 512 fp32): `bess_pertriple_tail` (one launch) against the four launches it replaces (combine inside
`bess_neg_score_pertriple_fwd_dq` is not separable: timed as forward with - forward without), each replayed back to
back from a hipGraph.  `python3 profiles/bench_tail.py`"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "bess-kge_amd"))
from besskge import _native as nat
from besskge._native import RowSource
dev = torch.device("cuda:0")


def replay_us(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


def main(S=4096, N=256, W=512, M=93_773, R=51):
    g = torch.Generator().manual_seed(0)
    table = (torch.randn(M, W, generator=g) / W).to(dev)
    rel = (torch.randn(R, W, generator=g) / W).to(dev)
    desc = nat.make_desc(nat.COMPLEX, 0, table, W)
    hi, ti = (torch.randint(M, (S,), generator=g, dtype=torch.int32).to(dev) for _ in range(2))
    ri = torch.randint(R, (S,), generator=g, dtype=torch.int32).to(dev)
    neg = RowSource(table, torch.randint(M, (S * N,), generator=g, dtype=torch.int32).to(dev))
    w = torch.full((1,), 1.0 / S, device=dev)
    ld = nat.LossDesc()
    ld.kind, ld.margin, ld.adversarial, ld.adversarial_scale, ld.loss_scale = nat.LOSS_LOGSIGMOID, 12.0, 1, 1.0, 1.0
    head, tail = RowSource(table, hi), RowSource(table, ti)
    q, pos = nat.query_triple_fwd(desc, nat.CORRUPT_TAIL, head, tail, rel, ri)
    out, dq = nat.neg_score_pertriple_fwd_dq(desc, ld, q, neg, N, pos, w)
    out, parts = nat.neg_score_pertriple_fwd_dq(desc, ld, q, neg, N, pos, w, defer=True)
    drel = torch.zeros(rel.shape, dtype=torch.float32, device=dev)
    nat.loss_fwd_bwd(ld, pos, out, w, True); nat.pertriple_tail(desc, ld, nat.CORRUPT_TAIL, head, tail, rel, ri, parts, pos, out, w, drel)
    torch.cuda.synchronize()
    t_fwd_c = replay_us(lambda: nat.neg_score_pertriple_fwd_dq(desc, ld, q, neg, N, pos, w), reps=4)
    t_fwd = replay_us(lambda: nat.neg_score_pertriple_fwd_dq(desc, ld, q, neg, N, pos, w, defer=True), reps=4)
    t_loss = replay_us(lambda: nat.loss_fwd_bwd(ld, pos, out, w, True))
    dp = torch.randn(S, device=dev)
    t_qtb = replay_us(lambda: nat.query_triple_bwd(desc, nat.CORRUPT_TAIL, head, tail, rel, ri, dp, dq, drel))
    t_tail = replay_us(lambda: nat.pertriple_tail(desc, ld, nat.CORRUPT_TAIL, head, tail, rel, ri, parts, pos, out, w, drel))
    print(f"S={S} N={N} W={W} items={parts[0].shape[1]}: forward {t_fwd:.1f} us (+ combine launch {t_fwd_c - t_fwd:.1f}), "
          f"loss {t_loss:.1f}, query/triple backward {t_qtb:.1f}  |  four launches {t_fwd_c - t_fwd + t_loss + t_qtb:.1f} us, "
          f"bess_pertriple_tail {t_tail:.1f} us", flush=True)


if __name__ == "__main__":
    main()
    main(S=512, N=64, W=256)
