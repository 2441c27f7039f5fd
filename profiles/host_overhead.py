#!/usr/bin/env python3
"""How much of a C2 step is host (Python / launch) time?  Runs bench's model, measures (a) host time to *issue* K
steps, (b) time until the GPU is done, for the scoring step and the training step with SGD and with AdamW."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import torch
import bench
from besskge import runtime
from besskge.collectives import SingleProcessGroup

dev = torch.device("cuda", 0)
model, sharding, k_pair = bench.build_c2(bench.N_ENTITY_C2, 1, 0, dev, SingleProcessGroup(1), False)
batches = bench.make_batches_c2(1, 0, sharding, k_pair, pool=4, dev=dev)
for mode, opt in (("score", None), ("train sgd", 1e-3), ("train adamw", runtime.Adam(lr=1e-3, weight_decay=1e-2))):
    def step(i):
        b = batches[i % 4]
        if opt is None:
            with torch.no_grad():
                model.forward_replicas([b])
        else:
            model.train_step_replicas([b], opt)
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode}: host issue {1e3*(t1-t0)/K:.3f} ms/step, until GPU done {1e3*(t2-t0)/K:.3f} ms/step", flush=True)
