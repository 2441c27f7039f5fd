#!/usr/bin/env python3
"""One process per GPU (the deployment mode): every rank holds one shard of the entity table,
samples only its own slice of every micro-batch on its device, and the ranks meet in the BESS
all-to-all / all-gather through torch.distributed (RCCL).

    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_multi_gpu.py
    # rehearsal on one GPU (ranks share the device, collectives staged through the host):
    BESS_BACKEND=gloo torchrun --nproc-per-node 2 --master-addr 127.0.0.1 examples/train_multi_gpu.py

Same synthetic graph and model as train_and_evaluate.py; rank 0 prints the loss and the
held-out hits@10 / MRR (every rank evaluates its share of the queries).
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), os.path.join(REPO, "examples")]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import besskge  # noqa: E402,F401
from besskge import runtime, scoring  # noqa: E402
from besskge.batch_sampler import RandomShardedBatchSampler, RigidShardedBatchSampler  # noqa: E402
from besskge.bess import EmbeddingMovingBessKGE, TopKQueryBessKGE  # noqa: E402
from besskge.collectives import DistributedGroup, NativeGroup  # noqa: E402
from besskge.dataset import KGDataset  # noqa: E402
from besskge.device_sampler import DeviceBatchSampler  # noqa: E402
from besskge.loss import LogSigmoidLoss  # noqa: E402
from besskge.metric import Evaluation  # noqa: E402
from besskge.negative_sampler import PlaceholderNegativeSampler, RandomShardedNegativeSampler  # noqa: E402
from besskge.sharding import PartitionedTripleSet, Sharding  # noqa: E402
from train_and_evaluate import synthetic_graph  # noqa: E402


def main(argv=None) -> dict:
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--n-entity", type=int, default=2000)
    ap.add_argument("--embedding-size", type=int, default=32)
    args = ap.parse_args(argv)
    backend = os.environ.get("BESS_BACKEND", "nccl")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, **(dict(device_id=dev) if backend == "nccl" else {}))
    # nccl: the library's own RCCL entry points on the kernels' stream (bess_comm_*); gloo (several ranks sharing one
    # GPU in a test): c10d with host staging
    group = NativeGroup(dev) if backend == "nccl" else DistributedGroup()

    # every rank builds the same (seeded) graph, sharding and samplers; only its shard goes to its GPU
    train, test = synthetic_graph(args.n_entity, 8, 30000, seed=0)
    ds = KGDataset(n_entity=args.n_entity, n_relation_type=8, triples={"train": train, "test": test},
                   original_triple_ids={"train": np.arange(len(train)), "test": np.arange(len(test))})
    sharding = Sharding.create(args.n_entity, world, seed=0)
    # only this rank's shard is allocated, directly on its GPU; the relation table is replicated (same seed everywhere)
    torch.manual_seed(100 + rank)
    fn = scoring.TransE(True, 1, sharding, 8, args.embedding_size, device=dev, shards=[rank])
    torch.manual_seed(0)
    torch.nn.init.uniform_(fn.relation_embedding.data, -1.0 / args.embedding_size, 1.0 / args.embedding_size)
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding, partition_mode="ht_shardpair")
    ns = RandomShardedNegativeSampler(n_negative=64, sharding=sharding, seed=1, corruption_scheme="t",
                                      local_sampling=False, flat_negative_format=True)
    bs = RandomShardedBatchSampler(pts, ns, shard_bs=512, batches_per_step=4, seed=2)
    model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn,
                                   loss_fn=LogSigmoidLoss(margin=6.0, negative_adversarial_sampling=True))
    trainer = runtime.training_model(model, runtime.Options(device_iterations=bs.batches_per_step),
                                     runtime.Adam(lr=0.05), group=group, device=dev)
    feed = DeviceBatchSampler(bs, dev, shards=[rank])  # this rank's slice of the common random streams
    losses = []
    for step in range(args.steps):
        out = trainer(**{k: v.flatten(end_dim=1) for k, v in feed.sample().items()})
        if step % 50 == 0 or step == args.steps - 1:
            loss = out["loss"].mean().reshape(1).clone()
            (loss,) = group.all_reduce_sum([loss])
            losses.append(float(loss) / world)
            if rank == 0:
                print(f"step {step:4d}  loss {losses[-1]:.4f}", flush=True)

    # evaluation: top-10 completion of (h, r, ?) over all entities, ground truth ranked among them
    pts_test = PartitionedTripleSet.create_from_dataset(ds, "test", sharding, partition_mode="h_shard")
    bs_test = RigidShardedBatchSampler(pts_test, PlaceholderNegativeSampler("t"), shard_bs=256, batches_per_step=1,
                                       seed=0)
    ev = Evaluation(["mrr", "hits@10"], worst_rank_infty=True, reduction="sum")
    topk = TopKQueryBessKGE(k=10, candidate_sampler=bs_test.negative_sampler, score_fn=fn, evaluation=ev)
    tester = runtime.inference_model(topk, runtime.Options(device_iterations=1), group=group, device=dev)
    totals = torch.zeros(3, device=dev)
    for idx in bs_test.get_dataloader_sampler(shuffle=False):
        b = bs_test[idx]
        own = {k: v[:, rank: rank + 1].flatten(end_dim=1) for k, v in b.items() if k in ("head", "relation", "tail", "triple_mask")}
        res = tester(**own)
        names = list(ev.metrics.keys())
        totals[0] += res["metrics"][:, names.index("mrr")].sum()
        totals[1] += res["metrics"][:, names.index("hits@10")].sum()
        totals[2] += own["triple_mask"].sum().to(dev) if "triple_mask" in own else own["relation"].numel()
    (totals,) = group.all_reduce_sum([totals])
    out = dict(losses=losses, mrr=float(totals[0] / totals[2]), **{"hits@10": float(totals[1] / totals[2])})
    if rank == 0:
        print(f"held-out tail prediction, top-10 over all {args.n_entity} entities on {world} shard(s):"
              f" mrr {out['mrr']:.3f}  hits@10 {out['hits@10']:.3f}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
