#!/usr/bin/env python3
"""The training recipe of the reference's first notebook (`notebooks/1_biokg_training_inference.ipynb`),
cell for cell, with `poptorch.X` replaced by `besskge.runtime.X`:

  * 4 shards, entities of one type contiguous inside every shard (`Sharding.create(type_offsets=...)`),
  * `RandomShardedNegativeSampler(n_negative=1, "ht", non-flat)` + negative sample sharing,
  * `RigidShardedBatchSampler(shard_bs=240, batches_per_step=device_iterations * accum_factor)`,
  * `options.deviceIterations(8)`, `options.Training.gradientAccumulation(6)`: a call consumes 48
    micro-batches and makes 8 weight updates, each with the summed gradient of 6 micro-batches,
  * RotatE (p = 1, embedding_size 64), `LogSigmoidLoss(margin=12, adversarial)`, AdamW,
  * validation with `TripleBasedShardedNegativeSampler` + `ScoreMovingBessKGE` + `Evaluation`.

ogbl-biokg itself cannot be downloaded here: the graph is a synthetic typed one of the same
structure (typed entities, candidate lists of the tail's / head's type for validation).

    python examples/biokg_recipe.py [--epochs 3] [--graphs]
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), os.path.join(REPO, "examples")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import besskge  # noqa: E402,F401
from besskge import runtime as poptorch  # noqa: E402  (the only line that differs from the notebook's imports)
from besskge.batch_sampler import RigidShardedBatchSampler  # noqa: E402
from besskge.bess import EmbeddingMovingBessKGE, ScoreMovingBessKGE  # noqa: E402
from besskge.dataset import KGDataset  # noqa: E402
from besskge.loss import LogSigmoidLoss  # noqa: E402
from besskge.metric import Evaluation  # noqa: E402
from besskge.negative_sampler import RandomShardedNegativeSampler, TripleBasedShardedNegativeSampler  # noqa: E402
from besskge.scoring import RotatE  # noqa: E402
from besskge.sharding import PartitionedTripleSet, Sharding  # noqa: E402


def typed_graph(n_per_type, n_rel, n_triple, n_cand, seed):
    """Entities clustered by type; relation r links type r % T to type (r + 1) % T; the tail of (h, r) is
    the entity of the target type nearest to a hidden `h + r` (learnable).  Validation triples come with
    `n_cand` candidate heads / tails of the right type, like ogbl-biokg's."""
    rng = np.random.default_rng(seed)
    T = len(n_per_type)
    offsets = np.concatenate([[0], np.cumsum(n_per_type)])
    n_entity = int(offsets[-1])
    ent = rng.normal(size=(n_entity, 8)).astype(np.float32)
    rel = rng.normal(size=(n_rel, 8)).astype(np.float32)
    r = rng.integers(n_rel, size=n_triple)
    th, tt = r % T, (r + 1) % T
    h = offsets[th] + rng.integers(1 << 30, size=n_triple) % np.asarray(n_per_type)[th]
    t = np.empty(n_triple, dtype=np.int64)
    for ty in range(T):
        sel = np.nonzero(tt == ty)[0]
        pool = ent[offsets[ty]: offsets[ty + 1]]
        dist = (((ent[h[sel]] + rel[r[sel]])[:, None, :] - pool[None, :, :]) ** 2).sum(-1)
        t[sel] = offsets[ty] + dist.argmin(-1)
    triples = np.unique(np.stack([h, r, t], axis=1), axis=0).astype(np.int32)
    rng.shuffle(triples)
    n_valid = len(triples) // 10
    parts = {"train": triples[n_valid:], "valid": triples[:n_valid]}
    v = parts["valid"]
    vh, vt = v[:, 1] % T, (v[:, 1] + 1) % T
    neg_heads = (offsets[vh][:, None] + rng.integers(1 << 30, size=(n_valid, n_cand)) % np.asarray(n_per_type)[vh][:, None])
    neg_tails = (offsets[vt][:, None] + rng.integers(1 << 30, size=(n_valid, n_cand)) % np.asarray(n_per_type)[vt][:, None])
    return KGDataset(
        n_entity=n_entity, n_relation_type=n_rel, triples=parts,
        original_triple_ids={k: np.arange(len(x)) for k, x in parts.items()},
        type_offsets={f"type{i}": int(offsets[i]) for i in range(T)},
        neg_heads={"valid": neg_heads.astype(np.int32)}, neg_tails={"valid": neg_tails.astype(np.int32)})


def main(argv=None) -> dict:
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--lr", type=float, default=0.01)  # (the notebook: 0.001 over 25 epochs of 4.8 M triples)
    ap.add_argument("--graphs", action="store_true", help="replay every call from one recorded hipGraph")
    args = ap.parse_args(argv)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    biokg = typed_graph([900, 700, 500, 300], n_rel=12, n_triple=60000, n_cand=100, seed=0)

    seed, n_shard = 1234, 4
    sharding = Sharding.create(n_entity=biokg.n_entity, n_shard=n_shard, seed=seed,
                               type_offsets=np.fromiter(biokg.type_offsets.values(), dtype=np.int32))
    train_triples = PartitionedTripleSet.create_from_dataset(dataset=biokg, part="train", sharding=sharding,
                                                             partition_mode="ht_shardpair")
    neg_sampler = RandomShardedNegativeSampler(n_negative=1, sharding=sharding, seed=seed, corruption_scheme="ht",
                                               local_sampling=False, flat_negative_format=False)
    device_iterations, accum_factor, shard_bs = 8, 6, 240
    batch_sampler = RigidShardedBatchSampler(partitioned_triple_set=train_triples, negative_sampler=neg_sampler,
                                             shard_bs=shard_bs, batches_per_step=device_iterations * accum_factor,
                                             seed=seed)

    options = poptorch.Options(use_graphs=args.graphs)
    options.replication_factor = sharding.n_shard
    options.deviceIterations(device_iterations)
    options.Training.gradientAccumulation(accum_factor)
    options._popart.setPatterns(dict(RemoveAllReducePattern=True))
    train_dl = batch_sampler.get_dataloader(options=options, shuffle=True, num_workers=0)

    logsigmoid_loss_fn = LogSigmoidLoss(margin=12.0, negative_adversarial_sampling=True)
    rotate_score_fn = RotatE(negative_sample_sharing=True, scoring_norm=1, sharding=sharding,
                             n_relation_type=biokg.n_relation_type, embedding_size=64)
    model = EmbeddingMovingBessKGE(negative_sampler=neg_sampler, score_fn=rotate_score_fn, loss_fn=logsigmoid_loss_fn)
    opt = poptorch.Adam(lr=args.lr, weight_decay=0.01)  # = AdamW (decoupled decay), row-sparse
    poptorch_model = poptorch.trainingModel(model, options=options, optimizer=opt)

    training_loss = []
    updates = 0
    t0 = time.time()
    for ep in range(args.epochs):
        ep_loss = []
        for batch in train_dl:
            triple_mask = batch.pop("triple_mask")
            batch.pop("triple_idx", None)
            res = poptorch_model(**{k: v.flatten(end_dim=1) for k, v in batch.items()})
            # res["loss"]: the summed loss of the last micro-batch, one value per shard (OutputMode.Final)
            ep_loss.append(float(torch.sum(res["loss"])) / triple_mask[-1].numel())
            updates += device_iterations
        training_loss.append(float(np.mean(ep_loss)))
        print(f"Epoch {ep + 1} loss: {training_loss[-1]:.6f}  ({updates} weight updates of {accum_factor} "
              f"micro-batches, {time.time() - t0:.1f} s)", flush=True)

    # ---- validation against each triple's own candidate lists (notebook cells "valid_triples" ... "evaluator")
    valid_triples = PartitionedTripleSet.create_from_dataset(dataset=biokg, part="valid", sharding=sharding,
                                                             partition_mode="ht_shardpair")
    ns_valid = TripleBasedShardedNegativeSampler(negative_heads=valid_triples.neg_heads,
                                                 negative_tails=valid_triples.neg_tails, sharding=sharding,
                                                 corruption_scheme="ht", seed=seed)
    bs_valid = RigidShardedBatchSampler(partitioned_triple_set=valid_triples, negative_sampler=ns_valid,
                                        shard_bs=shard_bs, batches_per_step=10, seed=seed, duplicate_batch=True)
    rotate_score_fn.negative_sample_sharing = False
    val_options = poptorch.Options()
    val_options.deviceIterations(bs_valid.batches_per_step)
    val_options.outputMode("All")
    valid_dl = bs_valid.get_dataloader(options=val_options, shuffle=False)
    evaluator = Evaluation(["mrr", "hits@1", "hits@5", "hits@10"], reduction="sum")
    model_inf = ScoreMovingBessKGE(negative_sampler=ns_valid, score_fn=rotate_score_fn, evaluation=evaluator)
    poptorch_model_inf = poptorch.inferenceModel(model_inf, options=val_options)
    totals, n_val = 0.0, 0
    for batch in valid_dl:
        batch.pop("triple_idx", None)
        res = poptorch_model_inf(**{k: v.flatten(end_dim=1) for k, v in batch.items()})
        totals = totals + res["metrics"].sum(dim=0).cpu()
        n_val += int(batch["triple_mask"].sum())
    metrics = {k: float(totals[i]) / n_val for i, k in enumerate(evaluator.metrics.keys())}
    print("validation (100 typed candidates per side):", "  ".join(f"{k} {v:.3f}" for k, v in metrics.items()))
    return dict(losses=training_loss, **metrics)


if __name__ == "__main__":
    main()
