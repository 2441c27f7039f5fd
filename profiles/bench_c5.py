#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: ONE shard at its real size (62.5 M rows x 512 fp32 = 128 GB, created on the
device through DistMult(..., device=, shards=)), per-triple negatives S = 8192 x K = 64 spread over the whole
shard (HBM + TLB-miss regime): dominant kernel TB/s, scoring step, training step (SGD)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import numpy as np
import torch
import besskge  # noqa: F401
from besskge import _native as nat
from besskge.bess import EmbeddingMovingBessKGE
from besskge.collectives import SingleProcessGroup
from besskge.loss import LogSigmoidLoss
from besskge.negative_sampler import RandomShardedNegativeSampler
from besskge.scoring import DistMult
from besskge.sharding import Sharding

dev = torch.device("cuda", 0)
M, D, N_REL, S, K = 62_500_000, 512, 1000, 8192, 64
t0 = time.perf_counter()
sharding = Sharding.create(M, 1, seed=7)
print(f"Sharding.create({M:,}, 1): {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
fn = DistMult(False, sharding, N_REL, D, device=dev, shards=[0])
torch.cuda.synchronize()
print(f"shard on the device: {fn.entity_embedding.numel() * 4 / 1e9:.0f} GB in {time.perf_counter() - t0:.1f} s; "
      f"HBM in use {torch.cuda.memory_allocated() / 1e9:.0f} GB", flush=True)
for p in (fn.entity_embedding, fn.relation_embedding):
    p.requires_grad_(False)
ns = RandomShardedNegativeSampler(K, sharding, 1, "t", local_sampling=False, flat_negative_format=False)
model = EmbeddingMovingBessKGE(ns, fn, LogSigmoidLoss(12.0, True))
model.attach(SingleProcessGroup(1), {0: 0})
rng = np.random.default_rng(0)
batches = []
for _ in range(4):
    b = dict(head=rng.integers(M, size=(1, 1, S)), relation=rng.integers(N_REL, size=(1, 1, S)),
             tail=rng.integers(M, size=(1, 1, S)), negative=rng.integers(M, size=(1, 1, S, K)))
    batches.append({k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in b.items()})
algo = S * K * (D * 4 + 8) + S * D * 4
for mode in ("score", "train"):
    def step(i):
        if mode == "score":
            with torch.no_grad():
                model.forward_replicas([batches[i % 4]])
        else:
            model.train_step_replicas([batches[i % 4]], 1e-3)
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    nat.start_kernel_timing(["bess_neg_score_pertriple_fwd", "bess_neg_score_pertriple_fwd_dq"])
    t0 = time.perf_counter()
    for i in range(30):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    ms = nat.stop_kernel_timing()
    k = [v for v in ms.values() if v][0]
    print(f"{mode}: {1e3 * dt:.3f} ms/step = {S * (1 + K) / dt / 1e9:.2f} G triples/s; dominant kernel {np.mean(k):.3f} ms = "
          f"{algo / np.mean(k) / 1e6:.0f} GB/s algorithmic ({algo / np.mean(k) / 1e6 / 8000:.2f} of the 8 TB/s spec)", flush=True)
