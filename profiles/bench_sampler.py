#!/usr/bin/env python3
"""Index feed of the BASELINE config-2 training loop (ComplEx d=256, 93,773
entities, S=4096 x 256 per-triple negatives): the reference's host path (numpy
samplers + H2D copy; DataLoader workers) against the device sampler that
continues the same PCG64 streams in HBM, and the end-to-end training rate each
one sustains."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import numpy as np
import torch
import besskge  # noqa: F401
from besskge import runtime
from besskge.batch_sampler import RandomShardedBatchSampler
from besskge.bess import EmbeddingMovingBessKGE
from besskge.dataset import KGDataset
from besskge.device_sampler import DeviceBatchSampler
from besskge.embedding import init_KGE_normal
from besskge.loss import LogSigmoidLoss
from besskge.negative_sampler import RandomShardedNegativeSampler
from besskge.scoring import ComplEx
from besskge.sharding import PartitionedTripleSet, Sharding

dev = torch.device("cuda", 0)
N_ENT, N_REL, N_TRIPLE, D, S, K, BPS = 93_773, 51, 4_762_678, 256, 4096, 256, 4
rng = np.random.default_rng(0)
triples = np.stack([rng.integers(N_ENT, size=N_TRIPLE), rng.integers(N_REL, size=N_TRIPLE),
                    rng.integers(N_ENT, size=N_TRIPLE)], axis=1)
ds = KGDataset(n_entity=N_ENT, n_relation_type=N_REL, triples={"train": triples},
               original_triple_ids={"train": np.arange(N_TRIPLE)})
sharding = Sharding.create(N_ENT, 1, seed=1234)
pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding, partition_mode="ht_shardpair")


def make_bs():
    ns = RandomShardedNegativeSampler(K, sharding, 1234, "t", local_sampling=False, flat_negative_format=False)
    return RandomShardedBatchSampler(pts, ns, S, BPS, seed=1234)


per_call = BPS * S * (1 + K)
# ---- sampling alone
host = make_bs()
for _ in range(2):
    host[[0]]
t0 = time.perf_counter()
R = 5
for _ in range(R):
    b = host[[0]]
t_host = (time.perf_counter() - t0) / R
t0 = time.perf_counter()
for _ in range(R):
    b = host[[0]]
    b = {k: v.to(dev, non_blocking=True) for k, v in b.items()}
torch.cuda.synchronize()
t_host_h2d = (time.perf_counter() - t0) / R
dbs = DeviceBatchSampler(make_bs(), dev)
for _ in range(3):
    dbs.sample()
torch.cuda.synchronize()
R = 50
t0 = time.perf_counter()
for _ in range(R):
    dbs.sample()
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / R
print(f"sampling one call = {BPS} micro-batches of {S} x (1+{K}) indices")
print(f"  host numpy                : {1e3*t_host:8.2f} ms/call  ({1e3*t_host/BPS:.2f} ms per micro-batch)")
print(f"  host numpy + H2D          : {1e3*t_host_h2d:8.2f} ms/call")
print(f"  device (bit-exact stream) : {1e3*t_dev:8.3f} ms/call  ({1e3*t_dev/BPS:.3f} ms per micro-batch, "
      f"{BPS*S*K*4/t_dev/1e9:.0f} GB/s of indices)", flush=True)

# ---- end-to-end training rate
torch.manual_seed(0)
fn = ComplEx(False, sharding, N_REL, D, [init_KGE_normal], [init_KGE_normal])
model = EmbeddingMovingBessKGE(negative_sampler=host.negative_sampler, score_fn=fn,
                               loss_fn=LogSigmoidLoss(margin=12.0, negative_adversarial_sampling=True))
runner = runtime.training_model(model, runtime.Options(device_iterations=BPS), runtime.SGD(lr=1e-3), device=dev)


def run(feed, calls):
    it = iter(feed)
    for _ in range(2):
        runner(**{k: v.flatten(end_dim=1) for k, v in next(it).items()})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        runner(**{k: v.flatten(end_dim=1) for k, v in next(it).items()})
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / calls


def device_feed():
    while True:
        yield dbs.sample()


def loader_feed(workers):
    bs = make_bs()
    while True:
        for b in bs.get_dataloader(shuffle=True, num_workers=workers, persistent_workers=workers > 0):
            yield b


print("end-to-end training (forward + backward + sparse SGD), per call of 4 micro-batches:")
for name, feed, calls in (("device sampler", device_feed(), 40), ("DataLoader, 0 workers", loader_feed(0), 6),
                          ("DataLoader, 5 workers (as the notebooks)", loader_feed(5), 12),
                          ("DataLoader, 12 workers", loader_feed(12), 20)):
    dt = run(feed, calls)
    print(f"  {name:42s}: {1e3*dt:8.2f} ms/call  {per_call/dt/1e9:6.3f} G triples/s", flush=True)
