#!/usr/bin/env python3
"""End-to-end use of the package, the way the reference's notebooks are written:
shard a knowledge graph, train with BESS on the device (device-side samplers,
fused forward + backward + sparse update), then rank every entity for held-out
queries and report filtered-free MRR / hits@10.

    python examples/train_and_evaluate.py [--n-shard 1] [--steps 400] [--scorer TransE]

The graph is synthetic but learnable: entities and relations get hidden TransE
embeddings and the tail of (h, r) is the entity nearest to h + r, so a trained
TransE / RotatE / PairRE ... model has a signal to find.  Everything runs on one GPU
(`n_shard` replicas step in lock-step: `SingleProcessGroup`); with one process per
GPU pass `group=besskge.collectives.DistributedGroup()` to the runners instead.
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bess-kge_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import besskge  # noqa: E402,F401
from besskge import runtime, scoring  # noqa: E402
from besskge.batch_sampler import RandomShardedBatchSampler, RigidShardedBatchSampler  # noqa: E402
from besskge.bess import EmbeddingMovingBessKGE  # noqa: E402
from besskge.dataset import KGDataset  # noqa: E402
from besskge.device_sampler import DeviceBatchSampler  # noqa: E402
from besskge.loss import LogSigmoidLoss  # noqa: E402
from besskge.metric import Evaluation  # noqa: E402
from besskge.negative_sampler import PlaceholderNegativeSampler, RandomShardedNegativeSampler  # noqa: E402
from besskge.pipeline import AllScoresPipeline  # noqa: E402
from besskge.sharding import PartitionedTripleSet, Sharding  # noqa: E402


def synthetic_graph(n_entity: int, n_rel: int, n_triple: int, seed: int):
    rng = np.random.default_rng(seed)
    ent = rng.normal(size=(n_entity, 8)).astype(np.float32)
    rel = rng.normal(size=(n_rel, 8)).astype(np.float32)
    h = rng.integers(n_entity, size=n_triple)
    r = rng.integers(n_rel, size=n_triple)
    target = ent[h] + rel[r]
    dist = ((target[:, None, :] - ent[None, :, :]) ** 2).sum(-1)
    dist[np.arange(n_triple), h] = np.inf  # no self loops
    t = dist.argmin(-1)
    triples = np.unique(np.stack([h, r, t], axis=1), axis=0)
    rng.shuffle(triples)
    n_test = len(triples) // 10
    return triples[n_test:], triples[:n_test]


def main(argv=None) -> dict:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-entity", type=int, default=2000)
    ap.add_argument("--n-relation", type=int, default=8)
    ap.add_argument("--n-triple", type=int, default=30000)
    ap.add_argument("--n-shard", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--scorer", default="TransE", choices=["TransE", "RotatE", "DistMult", "ComplEx", "PairRE", "TranS"])
    ap.add_argument("--embedding-size", type=int, default=32)
    ap.add_argument("--lr", type=float, default=0.05)
    args = ap.parse_args(argv)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)

    train, test = synthetic_graph(args.n_entity, args.n_relation, args.n_triple, seed=0)
    ds = KGDataset(n_entity=args.n_entity, n_relation_type=args.n_relation, triples={"train": train, "test": test},
                   original_triple_ids={"train": np.arange(len(train)), "test": np.arange(len(test))})
    sharding = Sharding.create(args.n_entity, args.n_shard, seed=0)

    # ---- model (same constructors as the reference)
    d = args.embedding_size
    if args.scorer in ("TransE", "RotatE", "PairRE", "TranS"):
        fn = getattr(scoring, args.scorer)(True, 1, sharding, args.n_relation, d)
    else:
        fn = getattr(scoring, args.scorer)(True, sharding, args.n_relation, d)

    # ---- training: shared ("flat") random negatives, log-sigmoid loss, Adam
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding, partition_mode="ht_shardpair")
    ns = RandomShardedNegativeSampler(n_negative=64, sharding=sharding, seed=1, corruption_scheme="t",
                                      local_sampling=False, flat_negative_format=True)
    bs = RandomShardedBatchSampler(pts, ns, shard_bs=512, batches_per_step=4, seed=2)
    model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn,
                                   loss_fn=LogSigmoidLoss(margin=6.0, negative_adversarial_sampling=True))
    trainer = runtime.training_model(model, runtime.Options(device_iterations=bs.batches_per_step),
                                     runtime.Adam(lr=args.lr), device=dev)
    feed = DeviceBatchSampler(bs, dev)  # the numpy streams of `bs`, continued in HBM
    losses = []
    t0 = time.perf_counter()
    for step in range(args.steps):
        out = trainer(**{k: v.flatten(end_dim=1) for k, v in feed.sample().items()})
        if step % 50 == 0 or step == args.steps - 1:
            losses.append(float(out["loss"].mean()))
            print(f"step {step:4d}  loss {losses[-1]:.4f}", flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step = bs.batches_per_step * args.n_shard * 512 * (1 + 64 * args.n_shard)
    print(f"{args.steps} steps in {dt:.2f} s  ({args.steps * per_step / dt / 1e6:.1f} M triples/s incl. sampling)")

    # ---- evaluation: score every entity as tail of the held-out (h, r, ?) queries
    pts_test = PartitionedTripleSet.create_from_dataset(ds, "test", sharding, partition_mode="h_shard")
    bs_test = RigidShardedBatchSampler(pts_test, PlaceholderNegativeSampler("t"), shard_bs=256, batches_per_step=2,
                                       seed=0, return_triple_idx=True)
    ev = Evaluation(["mrr", "hits@1", "hits@10"], mode="average", reduction="sum")
    pipe = AllScoresPipeline(bs_test, "t", fn, evaluation=ev, return_scores=False, device=dev)
    res = pipe()
    n = len(test)
    metrics = {k: float(v) / n for k, v in res["metrics"].items()}
    print("held-out tail prediction over all", args.n_entity, "entities:",
          "  ".join(f"{k} {v:.3f}" for k, v in metrics.items()))
    return dict(losses=losses, **metrics)


if __name__ == "__main__":
    main()
