#!/usr/bin/env python3
"""Training step of the bench workload (BASELINE config 2: ComplEx d=256, S=4096 x 256 per-triple
negatives, one shard) under each row-sparse optimiser of besskge.runtime."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "bess-kge_amd")]
sys.argv = ["bench.py"]
import torch
import bench
from besskge import runtime

dev = torch.device("cuda", 0)
bench.N_ENTITY_PER_SHARD = 93_773
for name, make in (("SGD (fused into the reduction)", lambda: 1e-3),
                   ("SGD + momentum", lambda: runtime.SGD(lr=1e-3, momentum=0.9)),
                   ("Adagrad", lambda: runtime.Adagrad(lr=1e-2)),
                   ("AdamW", lambda: runtime.Adam(lr=1e-3, weight_decay=1e-2))):
    model, sharding, k_pair = bench.build(1, 0, dev, "train", False)
    batches = bench.make_batches(1, 0, sharding, k_pair, pool=8, dev=dev)
    opt = make()
    for i in range(5):
        model.train_step_replicas([batches[i % 8]], opt)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for i in range(n):
        model.train_step_replicas([batches[i % 8]], opt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:34s} {1e3 * dt:7.3f} ms/step  {bench.S * (1 + bench.K_TOTAL) / dt / 1e9:6.2f} G triples/s", flush=True)
