#!/usr/bin/env python3
"""BASELINE configs[3] regime on one GPU (one shard of the 8-way wikikg2 setup): TransE d=256,
fp16 tables, 312,576 entities per shard, flat (shared) negatives, `augment_negative`, sampled-softmax
cross entropy - the wikikg2 notebook's training setup (3_wikikg2...ipynb:251-256), swept over the
micro-batch size S and the negatives per shard K as SURVEY 8d prescribes.  Scoring step and full
training step (forward + backward + sparse SGD) through runtime.Runner, with hipGraph replay."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import numpy as np
import torch
import besskge  # noqa: F401
from besskge import runtime
from besskge.bess import EmbeddingMovingBessKGE
from besskge.embedding import init_KGE_uniform
from besskge.loss import SampledSoftmaxCrossEntropyLoss
from besskge.negative_sampler import RandomShardedNegativeSampler
from besskge.scoring import TransE
from besskge.sharding import Sharding

dev = torch.device("cuda", 0)
N_ENT, N_REL, D, ITERS = 312_576, 535, 256, 8
sharding = Sharding.create(N_ENT, 1, seed=0)
for S, K in ((512, 32), (512, 256), (4096, 256), (4096, 2048), (16384, 2048), (65536, 2048)):
    ns = RandomShardedNegativeSampler(K, sharding, 0, "t", local_sampling=False, flat_negative_format=True)
    rng = np.random.default_rng(0)
    batch = dict(head=rng.integers(N_ENT, size=(ITERS, 1, S)), relation=rng.integers(N_REL, size=(ITERS, 1, S)),
                 tail=rng.integers(N_ENT, size=(ITERS, 1, S)), negative=rng.integers(N_ENT, size=(ITERS, 1, 1, K)))
    batch = {k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in batch.items()}
    for train in (False, True, "adamw"):  # "adamw": the notebook's optimiser (row-sparse AdamW here)
        if train == "adamw" and S > 4096:
            continue
        for graphs in ((False, True) if S <= 4096 else (False,)):
            torch.manual_seed(0)
            fn = TransE(True, 1, sharding, N_REL, D, [init_KGE_uniform], [init_KGE_uniform]).half()
            model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn, augment_negative=True,
                                           loss_fn=SampledSoftmaxCrossEntropyLoss(n_entity=2_500_604))
            opts = runtime.Options(device_iterations=ITERS, use_graphs=graphs, pipeline_streams=1)
            optim = runtime.Adam(lr=1e-3, weight_decay=1e-2) if train == "adamw" else runtime.SGD(lr=1e-3)
            runner = runtime.training_model(model, opts, optim, device=dev) if train else \
                runtime.inference_model(model, opts, device=dev)
            for _ in range(2):
                runner(**batch)
            torch.cuda.synchronize()
            R = 3
            t0 = time.perf_counter()
            for _ in range(R):
                runner(**batch)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (R * ITERS)
            n_neg = K + S  # augmentation adds the S tails of the micro-batch
            what = "adamw" if train == "adamw" else ("train" if train else "score")
            print(f"S={S:6d} K={K:5d} {what} graphs={'on ' if graphs else 'off'}: "
                  f"{1e6*dt:9.1f} us/micro-batch  {S*(1+n_neg)/dt/1e9:8.2f} G triples/s", flush=True)
