#!/usr/bin/env python3
"""Benchmark of the BESS hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode score|train]

Metric (BASELINE.json): positive+negative triples scored per second, and the
achieved HBM GB/s of the dominant (gather + score) kernel against the roofline.

Workload at N = 1 (BASELINE.json configs[1], "C2"): ogbl-biokg-shaped ComplEx,
embedding_size 256 (W = Wr = 512, fp32, 2 KiB rows), 93,773 entities,
51 relations, n_shard = 1; one step = one micro-batch of S = 4096 positive
triples, each scored against its own K = 256 negative tails (per-triple
negatives: the HBM-bound regime, one gathered row per scored triple) through
`EmbeddingMovingBessKGE`, with the log-sigmoid loss: K1 gather (fused), K2-K6
scoring, K8 loss.  `--mode train` adds backward + sparse SGD (K9/K10).
Synthetic indices (uniform), default-initialised tables; the index tensors of a
pool of distinct micro-batches are resident in HBM before the timed region.

N > 1 (one process per GPU, torch.distributed / RCCL): weak scaling - every GPU
holds a 93,773-row shard and scores S = 4096 positives against 256 negatives
spread over the N shards (K = 256 / N per shard pair) with
`ScoreMovingBessKGE` (queries all-gathered, scores returned by all-to-all: the
scheme the reference recommends for per-triple negatives, docs/source/bess.rst).
"""

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "bess-kge_amd"), REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_ENTITY_PER_SHARD = 93_773  # overridden by --entities-per-shard (out-of-cache variant)
N_REL = 51
D = 256  # complex embedding size -> W = 512
S = 4096
K_TOTAL = 256
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def build(n_shard: int, rank: int, dev: torch.device, mode: str, distributed: bool):
    import besskge  # noqa: F401
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE, ScoreMovingBessKGE
    from besskge.collectives import DistributedGroup, SingleProcessGroup
    from besskge.embedding import init_KGE_normal, initialize_entity_embedding
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import ComplEx
    from besskge.sharding import Sharding

    n_entity = N_ENTITY_PER_SHARD * n_shard
    sharding = Sharding.create(n_entity, n_shard, seed=1234)
    torch.manual_seed(rank)
    # only this rank's slice is allocated, directly on the device
    table = initialize_entity_embedding(sharding, [init_KGE_normal], [2 * D], device=dev, shards=[rank])
    placeholder = torch.zeros(n_shard, sharding.max_entity_per_shard, 0)
    fn = ComplEx.__new__(ComplEx)
    torch.nn.Module.__init__(fn)
    fn.negative_sample_sharing = False
    fn.sharding = sharding
    fn.embedding_size = D
    fn.entity_embedding = table
    torch.manual_seed(1)
    fn.relation_embedding = torch.nn.Parameter(init_KGE_normal(torch.empty(N_REL, 2 * D, device=dev)))
    del placeholder
    k_pair = K_TOTAL // n_shard
    ns = RandomShardedNegativeSampler(k_pair, sharding, 1234, "t", local_sampling=False, flat_negative_format=False)
    loss = LogSigmoidLoss(margin=12.0, negative_adversarial_sampling=True)
    cls = ScoreMovingBessKGE if distributed else EmbeddingMovingBessKGE
    model = cls(negative_sampler=ns, score_fn=fn, loss_fn=loss)
    group = DistributedGroup() if distributed else SingleProcessGroup(1)
    model.entity_embedding = fn.entity_embedding
    for p in (fn.entity_embedding, fn.relation_embedding):
        p.requires_grad_(False)
    model.attach(group, {rank: 0})
    return model, sharding, k_pair


def make_batches(n_shard: int, rank: int, sharding, k_pair: int, pool: int, dev: torch.device):
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


def cpu_baseline(seconds: float = 12.0):
    """The oracle (CPU restatement of the reference's torch path) on this box's
    host cores, on a bounded sample of the same workload: S_cpu triples x
    K_TOTAL per-triple negatives, ComplEx d=256, table of 93,773 rows."""
    from oracle import kge

    cores = usable_cores()
    torch.set_num_threads(cores)
    gen = torch.Generator().manual_seed(0)
    W = 2 * D
    n_ent = 93_773
    table = (torch.randn(1, n_ent, W, generator=gen) / W)
    rel = torch.randn(N_REL, W, generator=gen) / W
    s_cpu = 256
    spec = kge.StepSpec("ComplEx", 0, False, "t", False)
    rng = np.random.default_rng(0)
    batch = dict(
        head=torch.from_numpy(rng.integers(n_ent, size=(1, 1, s_cpu))),
        relation=torch.from_numpy(rng.integers(N_REL, size=(1, 1, s_cpu))),
        tail=torch.from_numpy(rng.integers(n_ent, size=(1, 1, s_cpu))),
        negative=torch.from_numpy(rng.integers(n_ent, size=(1, 1, s_cpu, K_TOTAL))),
    )
    loss = dict(kind="logsigmoid", margin=12.0, adversarial=True, adversarial_scale=1.0)
    with torch.no_grad():
        kge.bess_step(spec, "EmbeddingMoving", table, rel, batch, loss)  # warm-up
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < seconds:
            kge.bess_step(spec, "EmbeddingMoving", table, rel, batch, loss)
            reps += 1
        dt = time.perf_counter() - t0
    # forward + backward (torch autograd: dense zero-filled table gradient + index_put, the
    # reference's CPU training path), a few passes
    tg = table.clone().requires_grad_(True)
    rg = rel.clone().requires_grad_(True)
    t1 = time.perf_counter()
    reps_b = 0
    while reps_b < 2 or time.perf_counter() - t1 < seconds / 2:
        tg.grad = rg.grad = None
        res = kge.bess_step(spec, "EmbeddingMoving", tg, rg, batch, loss)
        torch.stack(res["loss"]).sum().backward()
        reps_b += 1
    dt_b = time.perf_counter() - t1
    model = ""
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return dict(
        value=reps * s_cpu * (1 + K_TOTAL) / dt,
        unit="triples/s",
        cores=cores,
        kind="port",
        cpu_model=model,
        sample=f"{reps} passes of {s_cpu} triples x {K_TOTAL} per-triple negatives (forward: gather+score+loss), "
               f"torch CPU fp32, {cores} threads, {dt:.1f} s",
        train_value=reps_b * s_cpu * (1 + K_TOTAL) / dt_b,
        train_sample=f"{reps_b} passes forward + autograd backward of the same sample, {dt_b:.1f} s",
    )


def main() -> None:
    global N_ENTITY_PER_SHARD
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=["score", "train"], default="score")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the scoring steps alternate on (default: 1 at N=1 so that the per-launch "
                         "HIP-event timing of the dominant kernel is not blurred by overlap, 2 at N>1 to overlap "
                         "the exchange with scoring)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--entities-per-shard", type=int, default=N_ENTITY_PER_SHARD,
                    help="rows per shard (default: the ogbl-biokg count; a 192 MB table sits in the "
                         "256 MiB Infinity Cache - pass e.g. 4000000 for an HBM-resident 8 GB shard)")
    args = ap.parse_args()
    N_ENTITY_PER_SHARD = args.entities_per_shard

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
    # BESS_BENCH_REHEARSE_DIST=1 at N = 1: the multi-GPU code path (ScoreMoving, DistributedGroup,
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
    if args.mode == "train" and distributed:
        raise SystemExit("--mode train is single-GPU (EmbeddingMoving) in this round")

    from besskge import _native as nat

    model, sharding, k_pair = build(world, rank, dev, args.mode, distributed)
    batches = make_batches(world, rank, sharding, k_pair, pool=8, dev=dev)
    lr = 1e-3

    # scoring micro-batches are independent (read-only tables): issue them
    # round-robin on two HIP streams so that the exchange / small kernels of
    # step i+1 overlap the gather+score kernel of step i.  Training steps
    # depend on each other (table updates) and stay on one stream.
    if args.streams <= 0:
        args.streams = 2 if distributed else 1
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
        scoring of micro-batch i, so that they sit in front of its score all-to-all in the
        in-order collective queue and run under its scoring kernel."""
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

    # N = 1, score mode: also time the full training step (forward + backward +
    # sparse SGD, K9/K10) on the same workload, reported as an extra object
    train_extra = None
    if not distributed and args.mode == "score":
        tsteps = max(5, min(args.steps, 30))
        for i in range(3):
            model.train_step_replicas([batches[i % len(batches)]], lr)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(tsteps):
            model.train_step_replicas([batches[i % len(batches)]], lr)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        train_extra = {
            "what": "same workload, full training step: gather+score+loss+backward+segmented scatter+sparse SGD",
            "value": S * (1 + K_TOTAL) * tsteps / dt,
            "unit": "triples/s",
            "ms_per_step": 1e3 * dt / tsteps,
            "steps": tsteps,
        }
        # the notebooks train with AdamW: same step with the row-sparse AdamW of besskge.runtime
        from besskge import runtime as _rt

        adamw = _rt.Adam(lr=1e-3, weight_decay=1e-2)
        for i in range(3):
            model.train_step_replicas([batches[i % len(batches)]], adamw)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(tsteps):
            model.train_step_replicas([batches[i % len(batches)]], adamw)
        torch.cuda.synchronize()
        train_extra["adamw_ms_per_step"] = 1e3 * (time.perf_counter() - t1) / tsteps

    n_neg = K_TOTAL  # negatives per positive, over all shards
    scored_per_step = world * S * (1 + n_neg)
    value = scored_per_step * args.steps / elapsed

    # roofline of the dominant kernel (K5 forward): algorithmic bytes per launch
    W, sz = 2 * D, 4
    rows = S * world * k_pair  # rows gathered per launch on this GPU
    nq = S * world
    algo_bytes = rows * (W * sz + 4 + 4) + nq * W * 4
    fwd = kernel_ms.get("bess_neg_score_pertriple_fwd", []) or kernel_ms.get("bess_neg_score_pertriple_fwd_dq", [])
    avg_ms = float(np.mean(fwd)) if fwd else float("nan")
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if fwd else float("nan")
    traffic = None
    pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("neg_score_pertriple_fwd_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
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
                "workload": f"ogbl-biokg-shaped ComplEx d=256 fp32 (W=512), {N_ENTITY_PER_SHARD:,} entities per shard, "
                            f"n_shard={world}, S=4096 positives x 256 per-triple negatives per GPU per step, "
                            f"{'ScoreMoving' if distributed else 'EmbeddingMoving'}, mode={args.mode} "
                            "(gather+score+loss" + ("+backward+sparse SGD)" if args.mode == "train" else ")"),
                "n_shard": world,
                "shard_bs": S,
                "negatives_per_triple": n_neg,
                "mode": args.mode,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_neg_pertriple_fwd (bess_neg_score_pertriple_fwd)" if args.mode == "score" else
                          "k_neg_pertriple_fwd<FUSE> + k_combine_dq (bess_neg_score_pertriple_fwd_dq)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes,
                "avg_launch_ms": avg_ms,
                "launches_timed": len(fwd),
            },
        }
        extra = {k: float(np.mean(v)) for k, v in kernel_ms.items() if v}
        line["kernel_avg_ms"] = extra
        if train_extra is not None:
            line["train_step"] = train_extra
        if not distributed and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
