#!/usr/bin/env python3
"""Where the HOST time of an eager C4 training step (S = 512, K = 32: the notebook's micro-batch) goes:
cProfile over 200 steps, top entries by own time.   python profiles/host_profile.py [c4s|c2]"""
import cProfile
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import numpy as np
import torch

import bench
from besskge import runtime
from besskge.bess import EmbeddingMovingBessKGE
from besskge.collectives import SingleProcessGroup
from besskge.loss import SampledSoftmaxCrossEntropyLoss
from besskge.negative_sampler import RandomShardedNegativeSampler
from besskge.scoring import TransE
from besskge.sharding import Sharding

dev = torch.device("cuda", 0)
what = sys.argv[1] if len(sys.argv) > 1 else "c4s"
if what == "c2":
    model, sharding, k_pair = bench.build_c2(bench.N_ENTITY_C2, 1, 0, dev, SingleProcessGroup(1), False)
    batches = bench.make_batches_c2(1, 0, sharding, k_pair, pool=4, dev=dev)

    def step(i):
        model.train_step_replicas([batches[i % 4]], 1e-3)
else:
    S_, K_ = 512, 32
    sharding = Sharding.create(bench.C4_ROWS_PER_SHARD, 1, seed=0)
    fn = TransE(True, 1, sharding, bench.C4_N_REL, bench.C4_D, device=dev, shards=[0], dtype=torch.float16)
    ns = RandomShardedNegativeSampler(K_, sharding, 0, "t", local_sampling=False, flat_negative_format=True)
    model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn, augment_negative=True,
                                   loss_fn=SampledSoftmaxCrossEntropyLoss(n_entity=bench.C4_N_ENTITY))
    rng = np.random.default_rng(0)
    M = bench.C4_ROWS_PER_SHARD
    batch = dict(head=rng.integers(M, size=(1, 1, S_)), relation=rng.integers(bench.C4_N_REL, size=(1, 1, S_)),
                 tail=rng.integers(M, size=(1, 1, S_)), negative=rng.integers(M, size=(1, 1, 1, K_)))
    batch = {k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in batch.items()}
    runner = runtime.training_model(model, runtime.Options(device_iterations=1), runtime.SGD(lr=1e-3), device=dev)

    def step(i):
        runner(**batch)

for i in range(20):
    step(i)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for i in range(N):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{what}: host issue {1e3 * (t1 - t0) / N:.3f} ms/step, until GPU done {1e3 * (t2 - t0) / N:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
