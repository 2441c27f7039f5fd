#!/usr/bin/env python3
"""next-1 measurement: top-k of queries against ALL entities on one MI355X, on the
shapes of the reference's published inference numbers:

  YAGO3-10   ComplEx d=128 (W=256) fp32, 123,182 entities, 5,000 (h, r, ?) queries, k=10
             reference: 0.0227 s on 4 IPUs (27 G scores/s), 0.1207 s on 1 IPU, 0.654 s on CPU
  wikikg2    TransE d=100 fp16 in the notebook -> here the BASELINE C4 width d=256 fp16,
             2,500,604 entities, 16,384 queries of the 429,456, k=10
             reference: 47.1 s for 429,456 queries on 4 IPUs (22.8 G scores/s)

Prints seconds per batch and scores/s (queries x entities / time), and the split
between the scoring kernel and the streaming top-k kernel.
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from besskge import runtime  # noqa: E402
from besskge.bess import TopKQueryBessKGE  # noqa: E402
from besskge.embedding import initialize_entity_embedding  # noqa: E402
from besskge.negative_sampler import PlaceholderNegativeSampler  # noqa: E402
from besskge.scoring import ComplEx, TransE  # noqa: E402
from besskge.sharding import Sharding  # noqa: E402

dev = torch.device("cuda", 0)


def run(name, cls, args, n_entity, n_rel, d, n_query, dtype, k=10):
    sharding = Sharding.create(n_entity, 1, seed=1234)
    W = 2 * d if cls is ComplEx else d
    Wr = W
    torch.manual_seed(0)
    fn = cls.__new__(cls)
    torch.nn.Module.__init__(fn)
    fn.negative_sample_sharing = True
    fn.sharding = sharding
    fn.embedding_size = d
    if cls is TransE:
        fn.scoring_norm = 1
    fn.entity_embedding = torch.nn.Parameter(torch.randn(1, n_entity, W, device=dev).to(dtype), requires_grad=False)
    fn.relation_embedding = torch.nn.Parameter(torch.randn(n_rel, Wr, device=dev).to(dtype), requires_grad=False)
    model = TopKQueryBessKGE(k=k, candidate_sampler=PlaceholderNegativeSampler("t"), score_fn=fn, return_scores=True,
                             window_size=1000)
    model.score_tile_bytes = int(os.environ.get("BESS_TOPK_TILE_MB", "1024")) << 20
    model.prune_scores = os.environ.get("BESS_TOPK_PRUNE", "1") == "1"
    model.first_tile = int(os.environ.get("BESS_TOPK_FIRST_TILE", model.first_tile))
    model.attach(runtime.SingleProcessGroup(1) if hasattr(runtime, "SingleProcessGroup") else None)
    rng = np.random.default_rng(0)
    batch = dict(relation=torch.from_numpy(rng.integers(n_rel, size=(1, n_query)).astype(np.int32)).to(dev),
                 head=torch.from_numpy(rng.integers(n_entity, size=(1, n_query)).astype(np.int32)).to(dev))
    from besskge import _native as nat

    for _ in range(2):
        model.forward_replicas([batch])
    torch.cuda.synchronize()
    reps = 5
    nat.start_kernel_timing(["bess_neg_score_shared_fwd", "bess_topk_update", "bess_gather_rows"])
    t0 = time.perf_counter()
    for _ in range(reps):
        out = model.forward_replicas([batch])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    km = nat.stop_kernel_timing()
    score_ms = sum(km.get("bess_neg_score_shared_fwd", [])) / reps
    topk_ms = sum(km.get("bess_topk_update", [])) / reps
    n_topk = len(km.get("bess_topk_update", [])) // reps
    print(f"{name:28s} {n_query:6d} queries x {n_entity:9,d} entities  W={W} {str(dtype)[6:]:8s}: {dt*1e3:8.2f} ms/batch"
          f"  {n_query*n_entity/dt/1e9:7.1f} G scores/s   (scoring {score_ms:7.2f} ms, top-k {topk_ms:7.2f} ms in {n_topk} launches)")
    if os.environ.get("BESS_TOPK_RANKS", "1") == "1":
        run_ranks(name, fn, sharding, n_entity, n_rel, n_query, batch)
    return out


def run_ranks(name, fn, sharding, n_entity, n_rel, n_query, batch):
    """next-2: full ranks of the same queries (true completion = a random entity), by counting in the scoring
    kernel's epilogue (AllScoresBESS.rank_counts_replicas) against scoring every entity into a matrix +
    bess_ranks_from_scores (what AllScoresPipeline does when scores are asked for too)."""
    from besskge import _native as nat
    from besskge._native import RowSource
    from besskge.bess import AllScoresBESS

    mod = AllScoresBESS(PlaceholderNegativeSampler("t"), fn, window_size=1000)
    mod.attach(runtime.SingleProcessGroup(1))
    rng = np.random.default_rng(1)
    truth = torch.from_numpy(rng.integers(n_entity, size=(1, n_query)).astype(np.int32)).to(dev)
    b = dict(batch, rank_truth=truth)

    def counted():
        return mod.rank_counts_replicas([b])[0]

    def matrix():
        q = mod._gather_queries([batch])[0]
        table = mod._local_table(0)
        desc = fn.kernel_desc()
        rows = sharding.entity_to_idx[truth.reshape(-1).cpu().numpy()]
        rows_t = torch.from_numpy(np.ascontiguousarray(rows)).to(device=dev, dtype=torch.int64)
        ranks = []
        tile = max(64, (1 << 30) // 4 // n_entity)  # 1 GiB score tiles, as the top-k passes
        for q0 in range(0, n_query, tile):
            qq = q[q0:q0 + tile]
            sc = nat.neg_score_shared_fwd(desc, qq, RowSource(table))
            if fn.relation_embedding.dtype == torch.float16:  # (scores leave AllScoresBESS in the model's dtype)
                sc = sc.half().float()
            r = torch.arange(qq.shape[0], device=dev)
            pos = sc[r, rows_t[q0:q0 + tile]].clone()
            sc[r, rows_t[q0:q0 + tile]] = -torch.inf
            ranks.append(nat.ranks_from_scores(pos, sc, 2, False))
        return torch.cat(ranks)

    res = {}
    for label, f in (("counted", counted), ("matrix", matrix)):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = f()
        torch.cuda.synchronize()
        res[label] = ((time.perf_counter() - t0) / 5, out)
    c = res["counted"][1]["counts"].float()
    r_counted = 1 + c[:, 0] + 0.5 * c[:, 1]
    agree = float((r_counted == res["matrix"][1]).float().mean())
    tc, tm = res["counted"][0], res["matrix"][0]
    print(f"{'  ranks, same queries':28s} counted in the epilogue {tc*1e3:8.2f} ms ({n_query*n_entity/tc/1e9:6.1f} G scores/s)"
          f"   score matrix + ranks_from_scores {tm*1e3:8.2f} ms ({n_query*n_entity/tm/1e9:6.1f} G)   ranks equal {agree:.4f}")


if __name__ == "__main__":
    from besskge.collectives import SingleProcessGroup

    runtime.SingleProcessGroup = SingleProcessGroup  # type: ignore[attr-defined]
    print(torch.cuda.get_device_name(0))
    run("YAGO3-10 ComplEx d=128", ComplEx, (), 123_182, 37, 128, 5000, torch.float32)
    run("YAGO3-10 ComplEx d=128 (all q)", ComplEx, (), 123_182, 37, 128, 20000, torch.float32)
    run("wikikg2 TransE d=256 fp16", TransE, (), 2_500_604, 535, 256, 4096, torch.float16)
    run("biokg ComplEx d=256", ComplEx, (), 93_773, 51, 256, 8192, torch.float32)
