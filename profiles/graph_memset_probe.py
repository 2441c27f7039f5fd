"""What does a memset node of a recorded hipGraph do on replay?  (VERDICT r3, weak #3.)

Round 3 replaced every `hipMemsetAsync` of the library by a fill kernel after Adam moments of 1e20 turned up from
the second replay of a recorded step (`use_graphs`, n_shard = 2, exchanged per-triple negatives) - but captured
steps still hold memset nodes that come from torch (`zeros` / `zero_()`), and they pass.  This probe settles it
with three experiments, each printed as one JSON line:

 A  micro: graph = [memset buf <- 0 | kernel buf += 1 | copy out <- buf] over pool memory; between replays the
    buffer is filled with garbage.  `out == 1` on every replay <=> memset nodes re-run, in order.
    Variants: torch `.zero_()` (what node does torch record?), sizes from 4 B to 64 MiB, unaligned tails,
    and the same inside a two-branch graph (side stream joined back).
 B  the failing scenario on the PROBE build of the library (`make PROBE_MEMSET=1`: fills through hipMemsetAsync,
    the round-2 form) next to the product build: 4 recorded AdamW steps, n_shard = 2, per-triple negatives;
    max |m| of the Adam state after every step, node-type counts of the recorded step.
 C  the memset nodes of B's graph one by one: destination, bytes, value (hipGraphMemsetNodeGetParams through the
    DOT dump) - is any of them shorter / elsewhere than the `hipMemsetAsync` call that made it?

    python profiles/graph_memset_probe.py [product|probe]     (B runs in the mode given; A always)
"""
import json
import os
import pathlib
import sys

REPO = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "bess-kge_amd"), str(REPO / "tests"), str(REPO)]
import torch  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "product"
from besskge import _native as nat  # noqa: E402

if mode == "probe":
    # (`import besskge` has loaded the product library: drop it and load the probe build in its place)
    nat.library_path = lambda: REPO / "profiles" / "probe_lib" / "libbesskge_hip_memset.so"
    nat._lib = None
    nat.load()
from besskge import runtime  # noqa: E402
from test_checkpoint import _batch, _model  # noqa: E402

dev = torch.device("cuda", 0)
out_dir = REPO / "gpurun_out"
out_dir.mkdir(exist_ok=True)


def micro(n_words: int, how: str, branch: bool) -> dict:
    """graph: clear(buf); buf += 1; out = buf.  Garbage goes into buf between replays."""
    buf = torch.full((n_words,), 7.0, device=dev)
    out = torch.empty_like(buf)
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        if branch:  # the clear runs on a forked stream and is joined back, as the index build of a step does
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                buf.zero_() if how == "torch_zero" else buf.fill_(0.0)
            cur.wait_stream(side)
        else:
            buf.zero_() if how == "torch_zero" else buf.fill_(0.0)
        buf.add_(1.0)
        out.copy_(buf)
    counts = nat.graph_node_counts(g)
    bad = []
    for k in range(6):
        buf.fill_(1000.0 + k)  # what an earlier replay / another user of the pool left there
        g.replay()
        torch.cuda.synchronize()
        if not bool((out == 1.0).all()):
            bad.append((k, float(out.min()), float(out.max())))
    return dict(exp="A", words=n_words, how=how, branch=branch, nodes=counts, wrong_replays=bad)


for how in ("torch_zero", "torch_fill"):
    for n in (1, 3, 1000, 4097, 1 << 20, (1 << 24) + 5):
        for branch in (False, True):
            print(json.dumps(micro(n, how, branch)), flush=True)

# ---- B: the round-3 failure, on the build asked for
model, sharding = _model(dev, n_shard=2)
batches = [_batch(sharding, 2, 16, 6, s) for s in range(5)]
x = [torch.empty(1 << 20, device=dev) for _ in range(64)]  # a warm allocator pool, as in the failing run
del x
runner = runtime.training_model(model, runtime.Options(use_graphs=True, keep_graph=True), runtime.Adam(lr=0.01), device=dev)
m_max = []
for i in range(4):
    runner(**batches[i])
    torch.cuda.synchronize()
    st = model._optimizer_state[model._local_table(0).data_ptr()]
    m_max.append(float(st["s"][0].abs().max()))
(counts,) = runner.graph_node_counts().values()
print(json.dumps(dict(exp="B", build=mode, adam_m_max_per_step=m_max, nodes=counts,
                      verdict="stale" if max(m_max) > 1e3 else "clean")), flush=True)
# ---- B2: the round-3 code path itself.  B no longer reaches a library clear (ComplEx rows that arrived through the
# exchange are scored by the fused forward since round 3; the backward then gets d_query = NULL).  TransE p = 1
# with 64 negatives per shard pair does: bess_neg_score_pertriple_bwd clears d_query (two work items per query add
# to it with atomics) - through hipMemsetAsync in the probe build.
from besskge.bess import EmbeddingMovingBessKGE as _EM  # noqa: E402
from besskge.loss import LogSigmoidLoss as _LS  # noqa: E402
from besskge.negative_sampler import RandomShardedNegativeSampler as _NS  # noqa: E402
from besskge.scoring import TransE as _TransE  # noqa: E402

torch.manual_seed(0)
fn2 = _TransE(False, 1, sharding, 7, 32, device=dev)
ns2 = _NS(64, sharding, 3, "h", local_sampling=False, flat_negative_format=False)
model2 = _EM(ns2, fn2, _LS(margin=2.0, negative_adversarial_sampling=True))
batches2 = [_batch(sharding, 2, 16, 64, s) for s in range(6)]
runner2 = runtime.training_model(model2, runtime.Options(use_graphs=True, keep_graph=True), runtime.Adam(lr=0.01), device=dev)
m2, t2 = [], []
for i in range(6):
    runner2(**batches2[i])
    torch.cuda.synchronize()
    st = model2._optimizer_state[model2._local_table(0).data_ptr()]
    m2.append(float(st["s"][0].abs().max()))
    t2.append(float(model2.score_fn.entity_embedding.detach().abs().max()))
(counts2,) = runner2.graph_node_counts().values()
# the same six steps without a graph, from the same start: the reference trajectory
torch.manual_seed(0)
fn3 = _TransE(False, 1, sharding, 7, 32, device=dev)
model3 = _EM(_NS(64, sharding, 3, "h", local_sampling=False, flat_negative_format=False), fn3,
             _LS(margin=2.0, negative_adversarial_sampling=True))
runner3 = runtime.training_model(model3, runtime.Options(), runtime.Adam(lr=0.01), device=dev)
for i in range(6):
    runner3(**batches2[i])
torch.cuda.synchronize()
off = (model3.score_fn.entity_embedding.detach() - model2.score_fn.entity_embedding.detach()).abs()
print(json.dumps(dict(exp="B2", build=mode, adam_m_max_per_step=m2, table_max_per_step=t2, nodes=counts2,
                      replayed_vs_eager_max_diff=float(off.max()), rows_differing=int((off.max(-1).values > 1e-4).sum()),
                      verdict="stale" if (max(m2) > 1e3 or float(off.max()) > 0.05) else "clean")), flush=True)

# ---- C: the DOT dump names every memset node with its parameters
(entry,) = runner._graphs.values()
dot = out_dir / f"graph_memset_probe_{mode}.dot"
try:
    entry[0].enable_debug_mode()
except Exception:
    pass
try:
    entry[0].debug_dump(str(dot))
    txt = dot.read_text() if dot.exists() else ""
    lines = [ln.strip() for ln in txt.splitlines() if "MEMSET" in ln.upper()]
    print(json.dumps(dict(exp="C", build=mode, memset_nodes=len(lines), sample=lines[:12])), flush=True)
except Exception as e:  # noqa: BLE001
    print(json.dumps(dict(exp="C", build=mode, error=f"{type(e).__name__}: {e}"[:300])), flush=True)

# ---- D: what do the recorded BASELINE steps hold?  (r03 trace of the C2 training step: five
# `__amd_rocclr_fillBufferAligned` per step - ROCclr's memset kernel - on the index-build stream)
import numpy as np  # noqa: E402

from besskge.bess import EmbeddingMovingBessKGE  # noqa: E402
from besskge.loss import LogSigmoidLoss, SampledSoftmaxCrossEntropyLoss  # noqa: E402
from besskge.negative_sampler import RandomShardedNegativeSampler  # noqa: E402
from besskge.scoring import ComplEx, TransE  # noqa: E402
from besskge.sharding import Sharding  # noqa: E402


def recorded_step(name, fn, flat, S, K, loss, augment, opt):
    sharding = fn.sharding
    ns = RandomShardedNegativeSampler(K, sharding, 0, "t", local_sampling=False, flat_negative_format=flat)
    model = EmbeddingMovingBessKGE(ns, fn, loss, augment_negative=augment)
    runner = runtime.training_model(model, runtime.Options(use_graphs=True, keep_graph=True), opt, device=dev)
    rng = np.random.default_rng(0)
    M = int(sharding.shard_counts[0])
    b = dict(head=rng.integers(M, size=(1, 1, S)), relation=rng.integers(fn.relation_embedding.shape[0], size=(1, 1, S)),
             tail=rng.integers(M, size=(1, 1, S)), negative=rng.integers(M, size=(1, 1, 1 if flat else S, K)))
    b = {k: torch.from_numpy(v.astype(np.int32)) for k, v in b.items()}
    for _ in range(3):
        runner(**b)
    torch.cuda.synchronize()
    (counts,) = runner.graph_node_counts().values()
    (entry,) = runner._graphs.values()
    dot = out_dir / f"graph_{name}_{mode}.dot"
    sample = []
    try:
        entry[0].debug_dump(str(dot))
        sample = [ln.strip()[:200] for ln in dot.read_text().splitlines() if "MEMSET" in ln.upper()][:8]
    except Exception as e:  # noqa: BLE001
        sample = [f"{type(e).__name__}: {e}"[:200]]
    print(json.dumps(dict(exp="D", build=mode, step=name, nodes=counts, memset_sample=sample)), flush=True)


sh = Sharding.create(93_773, 1, seed=0)
torch.manual_seed(0)
recorded_step("c2_sgd", ComplEx(False, sh, 51, 256, device=dev, shards=[0]), False, 4096, 256,
              LogSigmoidLoss(margin=12.0, negative_adversarial_sampling=True), False, runtime.SGD(lr=1e-3))
recorded_step("c2_adamw", ComplEx(False, sh, 51, 256, device=dev, shards=[0]), False, 4096, 256,
              LogSigmoidLoss(margin=12.0, negative_adversarial_sampling=True), False, runtime.Adam(lr=1e-3, weight_decay=1e-2))
sh4 = Sharding.create(312_576, 1, seed=0)
recorded_step("c4_512x32", TransE(True, 1, sh4, 535, 256, device=dev, shards=[0], dtype=torch.float16), True, 512, 32,
              SampledSoftmaxCrossEntropyLoss(n_entity=2_500_604), True, runtime.SGD(lr=1e-3))
recorded_step("c4_4096x256", TransE(True, 1, sh4, 535, 256, device=dev, shards=[0], dtype=torch.float16), True, 4096, 256,
              SampledSoftmaxCrossEntropyLoss(n_entity=2_500_604), True, runtime.SGD(lr=1e-3))
