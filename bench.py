#!/usr/bin/env python3
"""Benchmark of the BESS hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode score|train] [--comm auto|native|c10d]
                    [--workload c2|c4] [--c4-point S,K] [--c4-graph]

Metric (BASELINE.json): positive+negative triples scored per second, and the
achieved GB/s of the dominant (gather + score) kernel against the roofline.

Headline workload (BASELINE.json configs[1], "C2"): ogbl-biokg-shaped ComplEx,
embedding_size 256 (W = Wr = 512, fp32, 2 KiB rows), 93,773 entities per shard,
51 relations; one step = one micro-batch of S = 4096 positive triples per GPU,
each scored against its own K = 256 negative tails (per-triple negatives: the
HBM-bound regime, one gathered row per scored triple), with the log-sigmoid
loss: K1 gather (fused), K2-K6 scoring, K8 loss.  `--mode train` adds backward
+ sparse SGD (K9/K10).  N = 1: `EmbeddingMovingBessKGE`, n_shard = 1.  N > 1
(one process per GPU): weak scaling with `ScoreMovingBessKGE` (queries
all-gathered, scores returned by all-to-all: the scheme the reference recommends
for per-triple negatives, docs/source/bess.rst:75-86), K = 256 / N per shard pair.

Extra objects on the same JSON line:
  roofline      dominant kernel of the headline leg.  The C2 table is 192 MB and
                lives in the 256 MiB Infinity Cache: bound = "infinity-cache".
  roofline_hbm  (N = 1) the same launch on an HBM-resident shard (4 M rows, 8.2 GB),
                timed in this run with its own HIP events: the honest HBM figure.
  c4            BASELINE.json configs[3] / north_star's scaling workload, at every N:
                ogbl-wikikg2-shaped TransE d=256 **fp16**, 312,576 rows per shard,
                n_shard = N, flat (shared) negatives, `augment_negative`,
                sampled-softmax cross entropy, `EmbeddingMovingBessKGE`, full
                **training** step (notebooks/3_wikikg2_fp16.ipynb:251-256,344,385-392),
                swept over the micro-batch size S and negatives per shard pair K.
  train_step    (N = 1) the headline workload as a full training step (SGD, AdamW,
                eager and hipGraph replay).
  cpu_baseline  (N = 1) the oracle (CPU restatement of the reference's torch path) on
                the same S = 4096 x 256 micro-batch, on this box's host cores.
  xgmi          (N > 1) the library's all-to-all (`bess_alltoall`: one grouped send/recv per peer)
                timed at the per-peer message sizes of the c4 sweep, GB/s per GPU against 7 x 153 GB/s.

`--workload c4` makes north_star's scaling workload the line itself: `value`, `ms_per_step`, `dtype: "f16"` and
`config.workload` are the C4 training step at `--c4-point` (default 4096,256), `roofline` its VALU figure - the line
to take a 1 -> 8 curve on.  The default (`c2`) keeps BASELINE configs[1] as the headline at every N.

Synthetic indices (uniform), default-initialised tables (the scorers' own
constructors, allocating only this rank's shard, on the device); the index
tensors of a pool of distinct micro-batches are resident in HBM before the
timed region.
"""

import argparse
import json
import os
import sys
import threading
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "bess-kge_amd"), REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_ENTITY_C2 = 93_773
N_ENTITY_HBM = 4_000_000  # rows of the HBM-resident variant (8.2 GB of 2 KiB rows)
N_REL = 51
D = 256  # complex embedding size -> W = 512
S = 4096
K_TOTAL = 256
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0  # same guide: float4 copy / random-row gathers from HBM
VALU_PEAK_TLOPS = 78.6     # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, one op per lane per cycle

C4_ROWS_PER_SHARD = 312_576  # ceil(2,500,604 / 8)
C4_N_ENTITY = 2_500_604
C4_N_REL = 535
C4_D = 256
C4_GRAPH_DEADLINE_S = 180  # wall-clock bound on the recorded-collectives variant at N > 1 (see main)
XGMI_LINKS, XGMI_LINK_GBS = 7, 153.0  # per GPU: 7 point-to-point links of ~153 GB/s (MI355X_MICROARCH.md)
EXIT_ABANDONED = 3  # a leg hung and was abandoned: the line says so and the process does NOT report success
# (S per GPU, K per shard pair): SURVEY 8d's sweep - K in {32, 256, 2048}, S from the notebook's 512 to 65,536
C4_SWEEP = ((512, 32), (4096, 256), (4096, 2048), (16384, 256), (65536, 256))


# --------------------------------------------------------------------------- #
def make_group(comm: str, world: int, rank: int, dev: torch.device, distributed: bool, backend: str):
    """(group, name of the collective backend in use)."""
    from besskge.collectives import DistributedGroup, NativeGroup, SingleProcessGroup

    if not distributed:
        return SingleProcessGroup(1), "none"
    if comm == "c10d" or backend != "nccl":
        return DistributedGroup(), f"c10d/{backend}"
    # native: the library's own RCCL entry points on the kernels' stream.  A short self-check of the
    # three collectives against their definition decides - on all ranks together - whether to use them.
    ok, why, group = 1, "", None
    try:
        group = NativeGroup(dev)
        x = (1000.0 * rank + torch.arange(world, dtype=torch.float32, device=dev))[:, None].repeat(1, 64).contiguous()
        (y,) = group.all_to_all([x])
        want = (1000.0 * torch.arange(world, dtype=torch.float32, device=dev) + rank)[:, None].repeat(1, 64)
        (g,) = group.all_gather([torch.full((8,), float(rank), device=dev)])
        (r,) = group.all_reduce_sum([torch.full((8,), float(rank + 1), device=dev)])
        torch.cuda.synchronize()
        if not torch.equal(y, want):
            ok, why = 0, "all_to_all routing"
        elif not torch.equal(g, torch.arange(world, dtype=torch.float32, device=dev)[:, None].repeat(1, 8)):
            ok, why = 0, "all_gather order"
        elif not torch.equal(r, torch.full((8,), world * (world + 1) / 2.0, device=dev)):
            ok, why = 0, "all_reduce sum"
    except Exception as e:  # noqa: BLE001 - any failure means "use c10d"
        ok, why = 0, f"{type(e).__name__}: {e}"
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return group, "native (bess_comm_* / RCCL on the kernels' stream)"
    if comm == "native":
        raise SystemExit(f"--comm native: self-check failed on rank {rank}: {why or 'another rank failed'}")
    if rank == 0:
        print(f"bench: native collectives unavailable ({why or 'another rank failed'}); using c10d", file=sys.stderr)
    return DistributedGroup(), f"c10d/{backend} (native self-check failed)"


def build_c2(n_rows: int, n_shard: int, rank: int, dev: torch.device, group, distributed: bool):
    """The headline model through the public constructors: only this rank's shard, on the device."""
    from besskge.bess import EmbeddingMovingBessKGE, ScoreMovingBessKGE
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import ComplEx
    from besskge.sharding import Sharding

    sharding = Sharding.create(n_rows * n_shard, n_shard, seed=1234)
    torch.manual_seed(rank)
    fn = ComplEx(False, sharding, N_REL, D, device=dev, shards=[rank])
    torch.manual_seed(1)  # the relation table is replicated: same values on every rank
    torch.nn.init.normal_(fn.relation_embedding.data, std=1.0 / (2 * D))
    k_pair = K_TOTAL // n_shard
    ns = RandomShardedNegativeSampler(k_pair, sharding, 1234, "t", local_sampling=False, flat_negative_format=False)
    loss = LogSigmoidLoss(margin=12.0, negative_adversarial_sampling=True)
    cls = ScoreMovingBessKGE if distributed else EmbeddingMovingBessKGE
    model = cls(negative_sampler=ns, score_fn=fn, loss_fn=loss)
    for p in (fn.entity_embedding, fn.relation_embedding):
        p.requires_grad_(False)
    model.attach(group, {rank: 0})
    return model, sharding, k_pair


def make_batches_c2(n_shard: int, rank: int, sharding, k_pair: int, pool: int, dev: torch.device):
    """Index tensors of `pool` micro-batches for this rank, resident on the device.
    Layout of one replica's inputs (reference bess.py:142-156)."""
    rng = np.random.default_rng(1000 + rank)
    ppp = S // n_shard
    counts = sharding.shard_counts
    out = []
    for _ in range(pool):
        b = dict(
            head=rng.integers(counts[rank], size=(1, n_shard, ppp)),
            relation=rng.integers(N_REL, size=(1, n_shard, ppp)),
            tail=rng.integers(counts[rank], size=(1, n_shard, ppp)),
            negative=rng.integers(counts[rank], size=(1, n_shard, S, k_pair)),
        )
        out.append({k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in b.items()})
    return out


def usable_cores() -> int:
    """Host cores this process may really use (affinity mask and cgroup quota,
    not the machine's core count)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(budget_s: float = 12.0):
    """The oracle (CPU restatement of the reference's torch path) on this box's host cores, on the
    SAME micro-batch shape as the GPU run: S = 4096 triples x 256 per-triple negatives, ComplEx
    d=256, table of 93,773 rows.  Forward (gather + score + loss) and forward + autograd backward
    (dense zero-filled table gradient + index_put: the reference's CPU training path) separately;
    median over the passes that fit the time budget (at least 3 / 2)."""
    from oracle import kge

    cores = usable_cores()
    torch.set_num_threads(cores)
    gen = torch.Generator().manual_seed(0)
    W = 2 * D
    table = (torch.randn(1, N_ENTITY_C2, W, generator=gen) / W)
    rel = torch.randn(N_REL, W, generator=gen) / W
    spec = kge.StepSpec("ComplEx", 0, False, "t", False)
    rng = np.random.default_rng(0)
    batch = dict(
        head=torch.from_numpy(rng.integers(N_ENTITY_C2, size=(1, 1, S))),
        relation=torch.from_numpy(rng.integers(N_REL, size=(1, 1, S))),
        tail=torch.from_numpy(rng.integers(N_ENTITY_C2, size=(1, 1, S))),
        negative=torch.from_numpy(rng.integers(N_ENTITY_C2, size=(1, 1, S, K_TOTAL))),
    )
    loss = dict(kind="logsigmoid", margin=12.0, adversarial=True, adversarial_scale=1.0)
    scored = S * (1 + K_TOTAL)

    def timed(fn, min_passes: int, budget: float):
        fn()  # warm-up
        ts, t_all = [], time.perf_counter()
        while len(ts) < min_passes or (time.perf_counter() - t_all < budget and len(ts) < 20):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return ts

    def fwd():
        with torch.no_grad():
            kge.bess_step(spec, "EmbeddingMoving", table, rel, batch, loss)

    tg = table.clone().requires_grad_(True)
    rg = rel.clone().requires_grad_(True)

    def fwd_bwd():
        tg.grad = rg.grad = None
        res = kge.bess_step(spec, "EmbeddingMoving", tg, rg, batch, loss)
        torch.stack(res["loss"]).sum().backward()

    tf = timed(fwd, 3, budget_s)
    tb = timed(fwd_bwd, 2, budget_s)
    model = ""
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return dict(
        value=scored / float(np.median(tf)),
        unit="triples/s",
        cores=cores,
        kind="port",
        cpu_model=model,
        sample=f"median of {len(tf)} passes over one full micro-batch ({S} triples x {K_TOTAL} per-triple negatives: "
               f"gather+score+loss), torch CPU fp32, {cores} threads, {sum(tf):.1f} s",
        train_value=scored / float(np.median(tb)),
        train_sample=f"median of {len(tb)} passes forward + autograd backward of the same micro-batch, {sum(tb):.1f} s",
    )


# --------------------------------------------------------------------------- #
def kernel_roofline(kernel_ms, world: int, k_pair: int, mode: str):
    """Algorithmic bytes of the dominant kernel (K5 forward) per launch and its measured rate."""
    W, sz = 2 * D, 4
    rows = S * world * k_pair  # rows gathered per launch on this GPU
    nq = S * world
    algo_bytes = rows * (W * sz + 4 + 4) + nq * W * 4
    fwd = kernel_ms.get("bess_neg_score_pertriple_fwd", []) or kernel_ms.get("bess_neg_score_pertriple_fwd_dq", [])
    avg_ms = float(np.mean(fwd)) if fwd else float("nan")
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if fwd else float("nan")
    return dict(
        kernel="k_neg_pertriple_fwd (bess_neg_score_pertriple_fwd)" if mode == "score" else
               "k_neg_pertriple_fwd<FUSE> (bess_neg_score_pertriple_fwd_dq; its partials are combined by "
               "bess_pertriple_tail)",
        achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
        algorithmic_bytes_per_launch=algo_bytes, avg_launch_ms=avg_ms, launches_timed=len(fwd))


def hbm_leg(dev: torch.device, steps: int, warmup: int):
    """The headline launch on an HBM-resident shard: 4 M rows x 2 KiB = 8.2 GB (32x the Infinity
    Cache), same S x K, n_shard = 1; the kernel is timed with HIP events on its stream."""
    from besskge import _native as nat
    from besskge.collectives import SingleProcessGroup

    model, sharding, k_pair = build_c2(N_ENTITY_HBM, 1, 0, dev, SingleProcessGroup(1), False)
    batches = make_batches_c2(1, 0, sharding, k_pair, pool=4, dev=dev)
    with torch.no_grad():
        for i in range(warmup):
            model.forward_replicas([batches[i % len(batches)]])
        torch.cuda.synchronize()
        nat.start_kernel_timing(["bess_neg_score_pertriple_fwd"])
        t0 = time.perf_counter()
        for i in range(steps):
            model.forward_replicas([batches[i % len(batches)]])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out = kernel_roofline(nat.stop_kernel_timing(), 1, k_pair, "score")
    out["bound"] = "hbm"
    out["frac_of_achievable"] = out["achieved"] / HBM_ACHIEVABLE_GBS
    out["achievable_note"] = f"{HBM_ACHIEVABLE_GBS:.0f} GB/s = what a plain copy / random-row gather reaches (MI355X_MICROARCH.md)"
    out["traffic"] = None
    out["workload"] = (f"same launch as the headline on a shard of {N_ENTITY_HBM:,} rows x 2 KiB = "
                       f"{N_ENTITY_HBM * 2048 / 1e9:.1f} GB (HBM-resident)")
    out["ms_per_step"] = 1e3 * dt / steps
    out["value"] = S * (1 + K_TOTAL) * steps / dt
    del model
    torch.cuda.empty_cache()
    return out


def c4_leg(world: int, rank: int, dev: torch.device, group, distributed: bool, comm_name: str, steps: int,
           state: dict, variants=("eager", "graph"), sweep=None, warmup_calls: int = 2):
    """north_star's scaling workload: one shard of the 8-way wikikg2 setup per GPU (weak scaling:
    312,576 rows per shard whatever N), TransE d=256 fp16, flat negatives, augmentation,
    sampled-softmax CE, EmbeddingMoving, full training step (forward + backward + C8 + sparse SGD)."""
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.collectives import DistributedGroup
    from besskge.loss import SampledSoftmaxCrossEntropyLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import TransE
    from besskge.sharding import Sharding

    iters = 8
    n = world
    sharding = Sharding.create(C4_ROWS_PER_SHARD * n, n, seed=0)
    points = state.setdefault("points", {})
    for name in variants:
        graphs = name == "graph"
        plans = name == "plan"  # the step as a recorded list of library calls replayed from C (Options.use_plans)
        if (graphs or plans) and isinstance(group, DistributedGroup):
            continue
        for S_, K_ in (sweep or C4_SWEEP):
            ppp = S_ // n
            if ppp * n != S_ or S_ > state.get("max_s", 1 << 30):
                continue
            if (graphs or plans) and S_ > 4096:
                continue  # steps of 10+ ms: nothing for a replay to save
            rng = np.random.default_rng(100 + rank)
            M = int(sharding.shard_counts[rank])
            batch = dict(head=rng.integers(M, size=(iters, n, ppp)), relation=rng.integers(C4_N_REL, size=(iters, n, ppp)),
                         tail=rng.integers(M, size=(iters, n, ppp)), negative=rng.integers(M, size=(iters, n, 1, K_)))
            batch = {k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in batch.items()}
            n_neg = n * K_ + S_  # augmentation adds the S tails of the micro-batch
            scored = world * S_ * (1 + n_neg)
            point = points.setdefault((S_, K_), dict(shard_bs=S_, negatives_per_shard_pair=K_, negatives_per_triple=n_neg))
            try:
                torch.manual_seed(rank)
                fn = TransE(True, 1, sharding, C4_N_REL, C4_D, device=dev, shards=[rank], dtype=torch.float16)
                torch.manual_seed(1)
                torch.nn.init.uniform_(fn.relation_embedding.data, -1.0 / C4_D, 1.0 / C4_D)
                ns = RandomShardedNegativeSampler(K_, sharding, 0, "t", local_sampling=False, flat_negative_format=True)
                model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn, augment_negative=True,
                                               loss_fn=SampledSoftmaxCrossEntropyLoss(n_entity=C4_N_ENTITY))
                opts = runtime.Options(device_iterations=iters, use_graphs=graphs, use_plans=plans, pipeline_streams=1)
                runner = runtime.training_model(model, opts, runtime.SGD(lr=1e-3), group=group, device=dev)
                if graphs or plans:
                    # inputs resident in HBM where the recorded step reads them (what a device-side sampler
                    # writing into Runner.static_inputs() gives): a call is one graph launch / plan run, no input copies
                    static = runner.static_inputs(**batch)
                    for k_, v_ in batch.items():
                        static[k_].copy_(v_)
                    batch = static
                for _ in range(max(2, warmup_calls)):
                    runner(**batch)
                group.barrier()
                torch.cuda.synchronize()
                reps = max(1, -(-steps // iters))
                t0 = time.perf_counter()
                for _ in range(reps):
                    runner(**batch)
                group.barrier()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / (reps * iters)
                if distributed:
                    t = torch.tensor([dt], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt = float(t.item())
                point[f"{name}_ms_per_step"] = 1e3 * dt
                point[f"{name}_value"] = scored / dt
                point[f"{name}_steps_timed"] = reps * iters
            except Exception as e:  # noqa: BLE001 - a failing variant must not lose the line
                point[f"{name}_error"] = f"{type(e).__name__}: {e}"[:300]
            finally:
                runner = model = fn = None
                torch.cuda.empty_cache()
            best = min((point[k] for k in ("eager_ms_per_step", "plan_ms_per_step", "graph_ms_per_step") if k in point), default=None)
            if best is not None:
                point["ms_per_step"] = best
                point["value"] = scored / (best * 1e-3)
                # VALU work of the L1 distance matrix per GPU and step: forward |q - e| accumulate and the two
                # backward products, S x N x W elements each (csrc/neg_shared.hip states the ops per element)
                elems = S_ * n_neg * C4_D
                # lane-instructions per element: forward 1 (a packed max + a packed dot per 2 elements); backward 4
                # where both products come from one evaluation of sgn(q - e) (k_l1_bwd_both: sub, sign, two
                # multiply-adds; 256 <= S <= 8192, N >= 256, N % 32 == 0, S % 8 == 0), else 3 per product (sub, sign, multiply-add);
                # whole step / step time -> fraction of the issue peak
                both = 256 <= S_ <= 8192 and n_neg >= 256 and n_neg % 32 == 0 and S_ % 8 == 0
                per_elem = 5 if both else 7
                lane_ops = per_elem * elems / (best * 1e-3) / 1e12
                point["valu"] = dict(bound="valu", elements_per_product=elems,
                                     achieved=3 * elems / (best * 1e-3) / 1e12, unit="T element-updates/s (3 products/step)",
                                     lane_ops_per_element=per_elem, achieved_lane_ops=lane_ops,
                                     peak=VALU_PEAK_TLOPS, peak_unit="T lane-ops/s", frac=lane_ops / VALU_PEAK_TLOPS)
    return c4_summary(state, world, comm_name)


def c4_summary(state: dict, world: int, comm_name: str) -> dict:
    points = list(state.get("points", {}).values())
    n = world
    return dict(
        what="BASELINE configs[3] / north_star scaling workload: ogbl-wikikg2-shaped TransE d=256 fp16, "
             f"{C4_ROWS_PER_SHARD:,} rows per shard, n_shard={n}, flat negatives K per shard pair, augment_negative, "
             "sampled-softmax CE, EmbeddingMovingBessKGE, training step (fwd + bwd + C8 + sparse SGD); "
             "value = whole-job positive+negative triples scored/s",
        unit="triples/s", dtype="f16", scaling="weak", n_gpus=world, collectives=comm_name, sweep=points)


def xgmi_leg(group, world: int, rank: int, dev: torch.device) -> dict:
    """the all-to-all of the exchange step by itself.  Per-peer bytes of the c4 sweep: 49 kB (the notebook's
    S = 512, K = 32), 344 kB, 2.75 MB (S = 4096, K = 256), 32 MB.  Each GPU sends and receives (n - 1) such blocks
    over its point-to-point links; rate = bytes sent per GPU / time, max over ranks."""
    sizes = (49_152, 344_064, 2_752_512, 33_554_432)
    rows = []
    for b in sizes:
        x = torch.empty((world, b // 4), dtype=torch.float32, device=dev).normal_()
        for _ in range(3):
            group.all_to_all([x])
        group.barrier()
        torch.cuda.synchronize()
        reps = 20 if b < (1 << 22) else 8
        t0 = time.perf_counter()
        for _ in range(reps):
            group.all_to_all([x])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        sent = (world - 1) * b
        rows.append(dict(bytes_per_peer=b, us=1e6 * dt, gbs_per_gpu=sent / dt / 1e9,
                         frac_of_links=sent / dt / 1e9 / (min(world - 1, XGMI_LINKS) * XGMI_LINK_GBS)))
        del x
    return dict(what="all-to-all of [n, bytes_per_peer] through the group in use, back-to-back calls (launch + transfer); "
                     "gbs_per_gpu = (n - 1) * bytes_per_peer / time, one direction",
                peak_per_gpu_gbs=XGMI_LINKS * XGMI_LINK_GBS, links_used=min(world - 1, XGMI_LINKS), sizes=rows)


def newest_pmc_traffic():
    """(bytes per launch of the dominant kernel, source file) from the newest profiles/r*/pmc_traffic.json:
    rocprofv3 --pmc counters of a separate run of this command (they cannot be read inside the timed run)."""
    import glob

    cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9]*", "pmc_traffic.json")), reverse=True)
    cands.append(os.path.join(REPO, "profiles", "pmc_traffic.json"))
    for pmc in cands:
        if os.path.exists(pmc):
            try:
                return json.load(open(pmc)).get("neg_score_pertriple_fwd_bytes_per_launch"), os.path.relpath(pmc, REPO)
            except Exception:  # noqa: BLE001
                pass
    return None, None


def cpu_baseline_c4(S_: int, K_: int, budget_s: float = 8.0):
    """The oracle on one C4 micro-batch (TransE d=256 p=1, shared negatives + augmentation, sampled-softmax CE) on
    this box's host cores; tables hold fp16 values, arithmetic in fp32 (the oracle's restatement of `model.half()`)."""
    from oracle import kge

    cores = usable_cores()
    torch.set_num_threads(cores)
    gen = torch.Generator().manual_seed(0)
    M = 50_000  # rows actually touched are << this; the CPU cost is the S x N x W distance matrix, not the table
    table = (torch.rand(1, M, C4_D, generator=gen) * 2 - 1).div(C4_D).half().float()
    rel = (torch.rand(C4_N_REL, C4_D, generator=gen) * 2 - 1).div(C4_D).half().float()
    spec = kge.StepSpec("TransE", 1, True, "t", True, augment=True)
    rng = np.random.default_rng(0)
    batch = dict(head=torch.from_numpy(rng.integers(M, size=(1, 1, S_))),
                 relation=torch.from_numpy(rng.integers(C4_N_REL, size=(1, 1, S_))),
                 tail=torch.from_numpy(rng.integers(M, size=(1, 1, S_))),
                 negative=torch.from_numpy(rng.integers(M, size=(1, 1, 1, K_))))
    loss = dict(kind="ssce", n_entity=C4_N_ENTITY)
    tg, rg = table.clone().requires_grad_(True), rel.clone().requires_grad_(True)

    def fwd_bwd():
        tg.grad = rg.grad = None
        with kge.half_queries():
            res = kge.bess_step(spec, "EmbeddingMoving", tg, rg, batch, loss)
        torch.stack(res["loss"]).sum().backward()

    fwd_bwd()
    ts, t_all = [], time.perf_counter()
    while len(ts) < 2 or (time.perf_counter() - t_all < budget_s and len(ts) < 20):
        t0 = time.perf_counter()
        fwd_bwd()
        ts.append(time.perf_counter() - t0)
    scored = S_ * (1 + K_ + S_)
    return dict(value=scored / float(np.median(ts)), unit="triples/s", cores=cores, kind="port",
                sample=f"median of {len(ts)} passes forward + autograd backward over one micro-batch (S = {S_}, "
                       f"K = {K_}, {K_ + S_} shared negatives), torch CPU fp32 on fp16 table values, {cores} threads, "
                       f"{sum(ts):.1f} s")


def train_leg(model, batches, steps: int):
    """The headline workload as a full training step, eager and under hipGraph replay."""
    from besskge import runtime as _rt

    def eager(optimizer, n, repeats=3):
        """Best of `repeats` timings of n steps each (the eager step issues ~0.3 ms of host work per 0.6 ms step: a
        busy host shows up in a single timing; `eager_repeats_ms` keeps all of them)."""
        for i in range(3):
            model.train_step_replicas([batches[i % len(batches)]], optimizer)
        times = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n):
                model.train_step_replicas([batches[i % len(batches)]], optimizer)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t1) / n)
        eager.last = [round(1e3 * t, 4) for t in times]
        return min(times)

    tsteps = max(5, min(steps, 30))
    dt = eager(1e-3, tsteps)
    out = {
        "what": "same workload, full training step: gather+score+loss+backward+segmented scatter+sparse SGD",
        "value": S * (1 + K_TOTAL) / dt,
        "unit": "triples/s",
        "ms_per_step": 1e3 * dt,
        "steps": tsteps,
        "timing": "best of 3 runs of `steps` steps each",
        "eager_repeats_ms": eager.last,
    }
    # the notebooks train with AdamW: same step with the row-sparse AdamW of besskge.runtime
    out["adamw_ms_per_step"] = 1e3 * eager(_rt.Adam(lr=1e-3, weight_decay=1e-2), tsteps)
    # hipGraph replay of the SGD step and of the AdamW step (one launch per 8 steps)
    for key, opt in (("graph_ms_per_step", _rt.SGD(lr=1e-3)), ("adamw_graph_ms_per_step", _rt.Adam(lr=1e-3, weight_decay=1e-2))):
        try:
            iters = 8
            stacked = {k: torch.cat([batches[i % len(batches)][k] for i in range(iters)], dim=0) for k in batches[0]}
            runner = _rt.Runner(model, _rt.Options(device_iterations=iters, use_graphs=True), model.replica_group,
                                stacked["head"].device, opt)
            for _ in range(2):
                runner(**stacked)
            torch.cuda.synchronize()
            reps = max(1, tsteps // iters)
            t1 = time.perf_counter()
            for _ in range(reps):
                runner(**stacked)
            torch.cuda.synchronize()
            out[key] = 1e3 * (time.perf_counter() - t1) / (reps * iters)
            runner.reset_graphs()
            del runner
        except Exception as e:  # noqa: BLE001
            out[key.replace("ms_per_step", "error")] = f"{type(e).__name__}: {e}"[:300]
    return out


# --------------------------------------------------------------------------- #
def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=["score", "train"], default="score")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the scoring steps alternate on (default: 1 at N=1 so that the per-launch "
                         "HIP-event timing of the dominant kernel is not blurred by overlap, 2 at N>1 to overlap "
                         "the exchange with scoring)")
    ap.add_argument("--comm", choices=["auto", "native", "c10d"], default="auto",
                    help="collectives at N > 1: the library's own RCCL entry points on the kernels' stream "
                         "(native), torch.distributed (c10d), or native when its self-check passes (auto)")
    ap.add_argument("--workload", choices=["c2", "c4"], default="c2",
                    help="c2 (default): BASELINE configs[1] is the line; c4: north_star's scaling workload "
                         "(BASELINE configs[3], wikikg2-shaped TransE fp16 training step) is the line")
    ap.add_argument("--c4-point", default="4096,256", help="S,K of the --workload c4 headline (positives per GPU, "
                                                           "negatives per shard pair)")
    ap.add_argument("--c4-graph", action="store_true",
                    help="N > 1: also run the c4 sweep with the collectives recorded into a hipGraph.  Opt-in there: "
                         "recorded RCCL send/recv has only been rehearsed on one-rank communicators; it runs last, under "
                         "a deadline, and a hang ends the process with a non-zero code (never a silent success)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="headline leg only (no roofline_hbm / c4 / train_step)")
    ap.add_argument("--c4-max-s", type=int, default=65536, help="largest micro-batch of the c4 sweep (positives per GPU)")
    ap.add_argument("--entities-per-shard", type=int, default=N_ENTITY_C2,
                    help="rows per shard of the headline leg (default: the ogbl-biokg count)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # BESS_BENCH_BACKEND=gloo lets several ranks share one GPU (host-staged
    # collectives) to rehearse the N > 1 code path on a 1-GPU box; the real
    # multi-GPU run uses RCCL.
    backend = os.environ.get("BESS_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    # BESS_BENCH_REHEARSE_DIST=1 at N = 1: the multi-GPU code path (ScoreMoving, one process per GPU,
    # pipelined begin / finish, RCCL collectives) on a one-rank process group - what can be checked
    # of it on a 1-GPU box
    distributed = world > 1 or os.environ.get("BESS_BENCH_REHEARSE_DIST", "0") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = {} if world > 1 else dict(rank=0, world_size=1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(backend, **kw)

    from besskge import _native as nat

    group, comm_name = make_group(args.comm, world, rank, dev, distributed, backend)
    if args.workload == "c4":
        return main_c4(args, world, rank, dev, group, distributed, comm_name)
    model, sharding, k_pair = build_c2(args.entities_per_shard, world, rank, dev, group, distributed)
    batches = make_batches_c2(world, rank, sharding, k_pair, pool=8, dev=dev)
    lr = 1e-3

    # scoring micro-batches are independent (read-only tables): issue them
    # round-robin on two HIP streams so that the exchange / small kernels of
    # step i+1 overlap the gather+score kernel of step i.  Training steps
    # depend on each other (table updates) and stay on one stream.
    if args.streams <= 0:
        # (native collectives: ONE stream - two collectives of the one RCCL communicator in flight on different
        # streams could start in different orders on different ranks; c10d serialises on its own stream)
        args.streams = 2 if (distributed and not comm_name.startswith("native")) else 1
    main_stream = torch.cuda.current_stream(dev)
    streams = [main_stream] if args.mode == "train" or args.streams == 1 else \
        [torch.cuda.Stream(device=dev) for _ in range(args.streams)]

    def step(i: int) -> None:
        b = batches[i % len(batches)]
        with torch.cuda.stream(streams[i % len(streams)]):
            if args.mode == "train":
                model.train_step_replicas([b], lr)
            else:
                with torch.no_grad():
                    model.forward_replicas([b])

    def run_steps(first: int, count: int) -> None:
        """`count` scoring / training steps.  With more than one GPU the forward is software
        pipelined: the row gathers and all-gathers of micro-batch i + 1 are issued before the
        scoring of micro-batch i, so that they run under its scoring kernel (on the other stream)."""
        if not distributed or args.mode == "train":
            for i in range(first, first + count):
                step(i)
            return
        with torch.no_grad():
            with torch.cuda.stream(streams[first % len(streams)]):
                ctx = model.forward_begin([batches[first % len(batches)]])
            for i in range(first, first + count):
                nxt = None
                if i + 1 < first + count:
                    with torch.cuda.stream(streams[(i + 1) % len(streams)]):
                        nxt = model.forward_begin([batches[(i + 1) % len(batches)]])
                with torch.cuda.stream(streams[i % len(streams)]):
                    model.forward_finish(ctx)
                ctx = nxt

    def fence() -> None:
        if distributed:
            # drain this rank's own streams before the process group's barrier kernel is queued: the barrier runs
            # on torch's communicator, the steps on the library's - two communicators are never in flight together
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(0, args.warmup)
    fence()
    # only the dominant kernel is bracketed with HIP events: timing events are
    # barriers on the stream and cost ~0.9 ms/step when put around every kernel
    # of a training step (the other kernels' durations are in profiles/)
    # (in train mode the same pass also accumulates d_query: entry point ..._fwd_dq)
    timed = ["bess_neg_score_pertriple_fwd", "bess_neg_score_pertriple_fwd_dq"]
    nat.start_kernel_timing(timed)
    t0 = time.perf_counter()
    run_steps(0, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = nat.stop_kernel_timing()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    extra_legs = not args.no_extra_legs
    train_extra = hbm = None
    if not distributed and args.mode == "score" and extra_legs:
        train_extra = train_leg(model, batches, args.steps)
    elif distributed and args.mode == "score" and extra_legs:
        # the headline workload as a full training step in its multi-GPU form (ScoreMovingBessKGE: queries
        # all-gathered, scores and score gradients by all-to-all, fused forward with per-shard partials)
        tsteps = max(5, min(args.steps, 30))
        try:
            for i in range(3):
                model.train_step_replicas([batches[i % len(batches)]], 1e-3)
            fence()
            t1 = time.perf_counter()
            for i in range(tsteps):
                model.train_step_replicas([batches[i % len(batches)]], 1e-3)
            fence()
            dt = (time.perf_counter() - t1) / tsteps
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            train_extra = {
                "what": "same workload, full training step, one process per GPU (ScoreMovingBessKGE): gather + "
                        "all-gather + score (fused forward) + all-to-all + loss + backward + segmented scatter + sparse SGD",
                "value": world * S * (1 + K_TOTAL) / dt, "unit": "triples/s", "ms_per_step": 1e3 * dt, "steps": tsteps,
            }
        except Exception as e:  # noqa: BLE001 - an extra leg must not lose the line
            train_extra = {"error": f"{type(e).__name__}: {e}"[:300]}
    del model
    torch.cuda.empty_cache()
    if not distributed and extra_legs and args.entities_per_shard == N_ENTITY_C2:
        hbm = hbm_leg(dev, max(10, min(args.steps, 50)), 5)

    n_neg = K_TOTAL  # negatives per positive, over all shards
    scored_per_step = world * S * (1 + n_neg)
    value = scored_per_step * args.steps / elapsed

    roof = kernel_roofline(kernel_ms, world, k_pair, args.mode)
    table_mb = args.entities_per_shard * 2 * D * 4 / 1e6
    in_cache = table_mb < 256 * 1.048576
    roof["bound"] = "infinity-cache" if in_cache else "hbm"
    roof["note"] = (f"the {table_mb:.0f} MB shard sits in the 256 MiB Infinity Cache: this rate is a cache figure priced "
                    "against the 8 TB/s HBM spec; the HBM-resident figure of the same launch is `roofline_hbm`"
                    if in_cache else "HBM-resident shard")
    traffic, src = newest_pmc_traffic()
    roof["traffic"] = traffic
    roof["traffic_source"] = (f"{src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of a separate run of this command "
                              "(counters cannot be read inside the timed run)") if src else None

    line = {
        "metric": "positive+negative triples scored/sec",
        "value": value,
        "unit": "triples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"ogbl-biokg-shaped ComplEx d=256 fp32 (W=512), {args.entities_per_shard:,} entities per shard, "
                        f"n_shard={world}, S=4096 positives x 256 per-triple negatives per GPU per step, "
                        f"{'ScoreMoving' if distributed else 'EmbeddingMoving'}, mode={args.mode} "
                        "(gather+score+loss" + ("+backward+sparse SGD)" if args.mode == "train" else ")"),
            "n_shard": world,
            "shard_bs": S,
            "negatives_per_triple": n_neg,
            "mode": args.mode,
            "collectives": comm_name,
        },
        "roofline": roof,
    }
    line["kernel_avg_ms"] = {k: float(np.mean(v)) for k, v in kernel_ms.items() if v}
    if hbm is not None:
        line["roofline_hbm"] = hbm
    if train_extra is not None:
        line["train_step"] = train_extra
    if rank == 0 and not distributed and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline()
    if extra_legs:
        c4_steps = max(16, min(args.steps, 48))
        state: dict = {"max_s": args.c4_max_s}
        line["c4"] = c4_leg(world, rank, dev, group, distributed, comm_name, c4_steps, state, variants=("eager", "plan"))
        if world > 1:
            try:
                line["xgmi"] = xgmi_leg(group, world, rank, dev)
            except Exception as e:  # noqa: BLE001 - an extra leg must not lose the line
                line["xgmi"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 or args.c4_graph:
            run_graph_variant(line, "c4", world, rank, dev, group, distributed, comm_name, c4_steps, state)
        else:
            line["c4"]["graph_note"] = "hipGraph variant not run at N > 1 (opt-in: --c4-graph)"
    finish(line, rank, distributed, group)


def run_graph_variant(line: dict, key: str, world: int, rank: int, dev, group, distributed: bool, comm_name: str,
                      steps: int, state: dict, sweep=None) -> None:
    """The hipGraph variant of the c4 sweep.  At N > 1 it records RCCL send/recv into the graph; that has only been
    rehearsed on one-rank communicators (1-GPU boxes), so it runs last and under a deadline.  If it does not come
    back, rank 0 prints the line it has with `"graph_abandoned": true` and every rank leaves with a NON-ZERO code:
    a stuck leg is never reported as a success."""
    deadline = None
    if world > 1:
        def give_up() -> None:
            if rank == 0:
                line[key] = c4_summary(state, world, comm_name)
                line[key]["graph_note"] = f"hipGraph variant abandoned after {C4_GRAPH_DEADLINE_S} s"
                line["graph_abandoned"] = True
                print(json.dumps(line), flush=True)
            os._exit(EXIT_ABANDONED)

        deadline = threading.Timer(C4_GRAPH_DEADLINE_S, give_up)
        deadline.daemon = True
        deadline.start()
    line[key] = c4_leg(world, rank, dev, group, distributed, comm_name, steps, state, variants=("graph",), sweep=sweep)
    if deadline is not None:
        deadline.cancel()


def finish(line: dict, rank: int, distributed: bool, group) -> None:
    if rank == 0:
        print(json.dumps(line), flush=True)
    if distributed:
        # the line is out.  Teardown in the one order that cannot hang: the recorded steps (hipGraphs holding the
        # communicator's send / recv nodes) go first, then the communicator, then the process group
        # (NativeGroup.close; round 3 needed a watchdog here because ncclCommDestroy waited for a live graph)
        torch.cuda.synchronize()
        dist.barrier()
        if hasattr(group, "close"):
            group.close()
        dist.destroy_process_group()


def main_c4(args, world: int, rank: int, dev, group, distributed: bool, comm_name: str) -> None:
    """`--workload c4`: the C4 EmbeddingMoving training step IS the line (north_star: 1 -> 8 scaling on
    ogbl-wikikg2 TransE d=256).  One step = one micro-batch of S positives per GPU against n * K + S shared
    negatives: K1 gather + C1 exchange, packed-fp16 L1 scoring, sampled-softmax CE, both backward products,
    C8 return, round-once sparse SGD on the fp16 shard, C9 + relation update."""
    S_, K_ = (int(x) for x in args.c4_point.split(","))
    if S_ % world:
        raise SystemExit(f"--c4-point: S = {S_} is not a multiple of n_shard = {world}")
    state: dict = {"max_s": 1 << 30}
    sweep = ((S_, K_),)
    c4_leg(world, rank, dev, group, distributed, comm_name, args.steps, state, variants=("eager", "plan"), sweep=sweep,
           warmup_calls=-(-args.warmup // 8))
    line: dict = {}
    if world == 1 or args.c4_graph:
        run_graph_variant(line, "c4", world, rank, dev, group, distributed, comm_name, args.steps, state, sweep=sweep)
    point = state["points"][(S_, K_)]
    if "ms_per_step" not in point:
        raise SystemExit(f"c4 leg failed: {point}")
    which = min(("eager", "plan", "graph"), key=lambda v: point.get(f"{v}_ms_per_step", 1e30))
    n_neg = point["negatives_per_triple"]
    valu = dict(point["valu"])
    valu.update(kernel="k_l1_fwd_pk + k_neg_shared_bwd (both products), whole step", peak=VALU_PEAK_TLOPS,
                achieved=valu["achieved_lane_ops"], unit="T lane-ops/s", traffic=None,
                note="the L1 distance matrix has no matrix-core form: `lane_ops_per_element` VALU lane-instructions per "
                     "(query, candidate, column) over forward + two backward products (5 when both products share one "
                     "evaluation of sgn(q - e), else 7); achieved = that count / whole step time")
    out = {
        "metric": "positive+negative triples scored/sec",
        "value": point["value"], "unit": "triples/s", "n_gpus": world,
        "steps": point[f"{which}_steps_timed"], "warmup": max(2, -(-args.warmup // 8)) * 8,
        "ms_per_step": point["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {
            "workload": f"ogbl-wikikg2-shaped TransE d=256 fp16 (BASELINE configs[3]), {C4_ROWS_PER_SHARD:,} rows per "
                        f"shard, n_shard={world}, S={S_} positives per GPU x {n_neg} shared negatives (K={K_} per shard "
                        "pair + augmentation), EmbeddingMoving, training step (fwd + bwd + C8 + sparse SGD), "
                        f"{which} launch",
            "n_shard": world, "shard_bs": S_, "negatives_per_triple": n_neg, "mode": "train", "collectives": comm_name,
            "launch": which,
        },
        "roofline": valu,
        "c4": c4_summary(state, world, comm_name),
    }
    if world > 1:
        try:
            out["xgmi"] = xgmi_leg(group, world, rank, dev)
        except Exception as e:  # noqa: BLE001
            out["xgmi"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0 and world == 1 and not distributed and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c4(min(S_, 1024), K_)
    finish(out, rank, distributed, group)


if __name__ == "__main__":
    main()
