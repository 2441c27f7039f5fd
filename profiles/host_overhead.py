#!/usr/bin/env python3
"""How much of a step is host (Python / launch) time?  Runs bench's model in both
modes, measures (a) host time to *issue* K steps, (b) time until the GPU is done."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import torch
import bench

dev = torch.device("cuda", 0)
for mode in ("score", "train"):
    model, sharding, k_pair = bench.build(1, 0, dev, mode)
    batches = bench.make_batches(1, 0, sharding, k_pair, pool=4, dev=dev)
    def step(i):
        b = batches[i % 4]
        if mode == "train":
            model.train_step_replicas([b], 1e-3)
        else:
            with torch.no_grad():
                model.forward_replicas([b])
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
    print(f"{mode}: host issue {1e3*(t1-t0)/K:.3f} ms/step, until GPU done {1e3*(t2-t0)/K:.3f} ms/step")
