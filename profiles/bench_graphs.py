#!/usr/bin/env python3
"""Launch-bound regime: BASELINE configs[0] shape (10k entities, TransE d=128,
n_shard=1) with small micro-batches, through runtime.Runner with and without
hipGraph replay (Options.use_graphs)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import numpy as np
import torch
import besskge  # noqa: F401
from besskge import runtime
from besskge.bess import EmbeddingMovingBessKGE
from besskge.embedding import init_KGE_uniform
from besskge.loss import LogSigmoidLoss
from besskge.negative_sampler import RandomShardedNegativeSampler
from besskge.scoring import TransE
from besskge.sharding import Sharding

dev = torch.device("cuda", 0)
N_ENT, N_REL, D = 10_000, 20, 128
ITERS = 64
for S, K, flat in ((512, 32, True), (512, 32, False), (128, 16, True)):
    sharding = Sharding.create(N_ENT, 1, seed=0)
    ns = RandomShardedNegativeSampler(K, sharding, 0, "t", local_sampling=False, flat_negative_format=flat)
    rng = np.random.default_rng(0)
    neg_shape = (ITERS, 1, 1, K) if flat else (ITERS, 1, S, K)
    batch = dict(head=rng.integers(N_ENT, size=(ITERS, 1, S)), relation=rng.integers(N_REL, size=(ITERS, 1, S)),
                 tail=rng.integers(N_ENT, size=(ITERS, 1, S)), negative=rng.integers(N_ENT, size=neg_shape))
    batch = {k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in batch.items()}
    for train in (False, True):
        for graphs in (False, True):
            torch.manual_seed(0)
            fn = TransE(flat, 1, sharding, N_REL, D, [init_KGE_uniform], [init_KGE_uniform])
            model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn,
                                           loss_fn=LogSigmoidLoss(margin=6.0, negative_adversarial_sampling=False))
            opts = runtime.Options(device_iterations=ITERS, use_graphs=graphs, pipeline_streams=1)
            runner = runtime.training_model(model, opts, runtime.SGD(lr=1e-3), device=dev) if train else \
                runtime.inference_model(model, opts, device=dev)
            for _ in range(3):
                runner(**batch)
            torch.cuda.synchronize()
            R = 5
            t0 = time.perf_counter()
            for _ in range(R):
                runner(**batch)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (R * ITERS)
            n_neg = K
            print(f"S={S} K={K} {'flat' if flat else 'per-triple'} {'train' if train else 'score'} "
                  f"graphs={'on ' if graphs else 'off'}: {1e6*dt:8.1f} us/micro-batch  "
                  f"{S*(1+n_neg)/dt/1e6:8.1f} M triples/s", flush=True)
