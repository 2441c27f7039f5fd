"""Worker of the multi-process tests (launched by tests/test_distributed.py).

    python tests/_dist_worker.py <mode> <out_dir>       (RANK / WORLD_SIZE / MASTER_* in env)

mode "routing" (CPU, gloo): DistributedGroup collectives + the per-rank view of
    a sharded batch; every rank dumps what it received.
mode "bess" (one GPU shared by all ranks, gloo with host staging): the full
    distributed BessKGE forward / train step through the HIP kernels for golden
    cases; every rank dumps its outputs.
"""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (os.path.join(REPO, "bess-kge_amd"), REPO, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


_GROUPS = []


def device() -> torch.device:
    """The GPU of this rank: the box's one GPU shared by all ranks (default), or - BESS_DIST_DEVICE=local_rank,
    the real layout, tests/test_multi_gpu.py - one GPU per rank."""
    if os.environ.get("BESS_DIST_DEVICE", "shared") == "local_rank":
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    return torch.device("cuda", 0)


def make_group():
    """DistributedGroup (c10d: gloo / nccl), or - BESS_DIST_BACKEND=native - NativeGroup: the library's
    own RCCL entry points (bess_comm_*), the RCCL id handed round over a gloo process group."""
    from besskge.collectives import DistributedGroup, NativeGroup

    g = NativeGroup(device()) if os.environ.get("BESS_DIST_BACKEND", "gloo") == "native" else DistributedGroup()
    _GROUPS.append(g)
    return g


def options(**kw):
    """runtime.Options of a worker run; BESS_USE_GRAPHS=1 records the steps (collectives included) into hipGraphs."""
    from besskge import runtime

    return runtime.Options(use_graphs=os.environ.get("BESS_USE_GRAPHS", "0") == "1", **kw)


def routing(out_dir: str) -> None:
    g = make_group()
    n, r = g.n_shard, g.rank
    native = os.environ.get("BESS_DIST_BACKEND", "gloo") == "native"
    dev = device() if native else torch.device("cpu")
    # block j of rank r carries the value 100*r + j
    x = torch.stack([torch.full((3, 2), 100.0 * r + j) for j in range(n)]).to(dev)
    (a2a,) = g.all_to_all([x])
    (ag,) = g.all_gather([torch.full((2,), float(r), device=dev)])
    (ar,) = g.all_reduce_sum([torch.full((4,), float(r + 1), device=dev)])
    ids = torch.arange(6, dtype=torch.int32).reshape(n, -1) + 10 * r if n in (1, 2, 3, 6) else torch.zeros(n, 1, dtype=torch.int32)
    (ag_i,) = g.all_gather([ids.to(dev)])
    extra = {}
    if native:
        # K1 + C1 in one call: rank r packs rows 7 * j + r of its table for rank j; the table of rank r is 1000 r + row
        table = (1000.0 * r + torch.arange(64, dtype=torch.float32, device=dev))[:, None].repeat(1, 8).half().contiguous()
        idx = (7 * torch.arange(n, dtype=torch.int32, device=dev)[:, None] + r
               + torch.zeros((1, 5), dtype=torch.int32, device=dev)).contiguous()
        extra["packed"] = g.pack_exchange(table, idx).float().cpu().numpy()
        torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"routing_{r}.npz"), a2a=a2a.cpu().numpy(), ag=ag.cpu().numpy(),
             ar=ar.cpu().numpy(), ag_i=ag_i.cpu().numpy(), **extra)


def bess(out_dir: str) -> None:
    from besskge import runtime
    from test_hip_parity import build_model
    from test_oracle import load_bess_case

    g = make_group()
    n, r = g.n_shard, g.rank
    dev = device()
    cases = [c for c in os.environ["BESS_CASES"].split(",") if c]
    out = {}
    for case in cases:
        c = load_bess_case(case)
        assert c["meta"]["n_shard"] == n
        bps = c["meta"]["bps"]
        keys = ("head", "relation", "tail", "negative", "negative_mask")
        batch = {k: c["batch"][k].flatten(end_dim=1) for k in keys if k in c["batch"]}
        model = build_model(c, dev)
        runner = runtime.inference_model(model, options(device_iterations=bps), group=g, device=dev)
        if c["net"] is not None:
            model.train()  # ConvE fixtures: train mode (batch statistics), no dropout
        res = runner(**batch)
        for k, v in res.items():
            out[f"{case}_fwd_{k}"] = v.float().cpu().numpy()
        if case.startswith("tr_"):
            model = build_model(c, dev)
            lr = 0.125
            runner = runtime.training_model(model, options(device_iterations=1), runtime.SGD(lr=lr), group=g, device=dev)
            recorded = os.environ.get("BESS_USE_GRAPHS", "0") == "1"
            snap = runner._training_snapshot() if recorded else None  # tables, dense parameters, buffers
            res = runner(**{k: v[: n] for k, v in batch.items()})
            if recorded:
                # recorded steps: the first call recorded AND took the step; a replay from the same start must take
                # the same one (the collectives of the graph re-run with the peers' data)
                first = model.score_fn.entity_embedding.detach().clone()
                runner._restore_training_snapshot(snap)
                res = runner(**{k: v[: n] for k, v in batch.items()})
                torch.cuda.synchronize()
                # (fp32 atomics of the plain-SGD scatter: equal up to the order of the additions)
                torch.testing.assert_close(model.score_fn.entity_embedding.detach(), first, rtol=1e-5, atol=1e-6,
                                           msg=f"{case}: replay != recording step")
            out[f"{case}_train_loss"] = res["loss"].float().cpu().numpy()
            out[f"{case}_train_entity"] = model.score_fn.entity_embedding.detach().float().cpu().numpy()
            out[f"{case}_train_relation"] = model.score_fn.relation_embedding.detach().float().cpu().numpy()
            if os.environ.get("BESS_ACCUMULATE"):
                # gradient accumulation, one process per shard: two micro-batches, one update
                model = build_model(c, dev)
                opt = runtime.Adam(lr=0.01) if os.environ["BESS_ACCUMULATE"] == "adam" else runtime.SGD(lr=lr)
                runner = runtime.training_model(model, runtime.Options(device_iterations=1, gradient_accumulation=2,
                                                                       output_mode="all"), opt, group=g, device=dev)
                res = runner(**batch)  # bps = 2 micro-batches
                out[f"{case}_acc_loss"] = res["loss"].float().cpu().numpy()
                out[f"{case}_acc_entity"] = model.score_fn.entity_embedding.detach().float().cpu().numpy()
                out[f"{case}_acc_relation"] = model.score_fn.relation_embedding.detach().float().cpu().numpy()
    np.savez(os.path.join(out_dir, f"bess_{r}.npz"), **out)


def topk(out_dir: str) -> None:
    """Distributed TopKQueryBessKGE / AllScoresBESS on golden cases (all ranks share the GPU)."""
    from besskge import runtime
    from besskge.bess import AllScoresBESS, TopKQueryBessKGE
    from besskge.sharding import Sharding
    from test_hip_parity import make_scorer
    from test_query import candidate_sampler, load_query_case

    g = make_group()
    n, r = g.n_shard, g.rank
    dev = device()
    out = {}
    for spec in [c for c in os.environ["BESS_CASES"].split(",") if c]:
        fix, case = spec.split(":")
        c = load_query_case(fix, case)
        m = c["meta"]
        bps = m["bps"]
        fn = make_scorer(c["scorer"], m["norm"], bool(m["flat"]), m["n_rel"], m["d"], c["table"], c["rel"],
                         torch.device("cpu"), sharding=Sharding.create(m["n_entity"], n, seed=1234))
        if fix == "topk":
            model = TopKQueryBessKGE(k=m["k"], candidate_sampler=candidate_sampler(c), score_fn=fn, return_scores=True,
                                     window_size=m["window"])
            runner = runtime.inference_model(model, options(device_iterations=bps), group=g, device=dev)
            keys = ("relation", "head", "tail", "negative", "triple_mask", "negative_mask")
            res = runner(**{k: c["batch"][k].flatten(end_dim=1) for k in keys if k in c["batch"]})
            out[f"{case}_ids"] = res["topk_global_id"].cpu().numpy()
            out[f"{case}_scores"] = res["topk_scores"].float().cpu().numpy()
        else:
            model = AllScoresBESS(candidate_sampler(c), fn, window_size=m["window"])
            runner = runtime.inference_model(model, runtime.Options(device_iterations=bps), group=g, device=dev)
            known = "tail" if c["scheme"] == "h" else "head"
            inp = {k: c["batch"][k].flatten(end_dim=1) for k in ("relation", known)}
            steps = [runner(step=torch.full((bps * n, 1), s, dtype=torch.int32), **inp).float().cpu().numpy()
                     for s in range(model.n_step)]
            out[f"{case}_scores"] = np.stack(steps)
            # rank-counting mode: positives scored on their shard and summed, counts from every shard back to the
            # query's (all_gather + all_reduce + all_to_all of the group)
            from test_query import rank_inputs

            truth, filt = rank_inputs(c, bps * n, int(inp["relation"].shape[1]))
            res = runner(step=torch.zeros((bps * n, 1), dtype=torch.int32), **inp, rank_truth=truth, rank_filter=filt)
            out[f"{case}_rank_counts"] = res["counts"].cpu().numpy()
            out[f"{case}_rank_pos"] = res["pos_score"].float().cpu().numpy()
    np.savez(os.path.join(out_dir, f"topk_{r}.npz"), **out)


def sampler(out_dir: str) -> None:
    """Every rank samples only its own rows on the device (`shards=[rank]`) and
    feeds them to the runner; results must equal the run fed with the host
    sampler's full batch."""
    from besskge import runtime
    from besskge.batch_sampler import RandomShardedBatchSampler
    from besskge.bess import EmbeddingMovingBessKGE, ScoreMovingBessKGE
    from besskge.collectives import DistributedGroup
    from besskge.dataset import KGDataset
    from besskge.device_sampler import DeviceBatchSampler
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import TransE
    from besskge.sharding import PartitionedTripleSet, Sharding

    g = DistributedGroup()
    n, r = g.n_shard, g.rank
    dev = torch.device("cuda", 0)
    n_entity, n_rel, n_triple, bps, shard_bs, K = 600, 7, 4000, 3, 24, 5
    rng = np.random.default_rng(11)
    triples = np.stack([rng.integers(n_entity, size=n_triple), rng.integers(n_rel, size=n_triple),
                        rng.integers(n_entity, size=n_triple)], axis=1)
    ds = KGDataset(n_entity=n_entity, n_relation_type=n_rel, triples={"train": triples},
                   original_triple_ids={"train": np.arange(n_triple)})
    sharding = Sharding.create(n_entity, n, seed=5)
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding, partition_mode="ht_shardpair")
    out = {}
    for name, cls, flat in (("em_flat", EmbeddingMovingBessKGE, True), ("sm_pt", ScoreMovingBessKGE, False)):
        def make_bs():
            ns = RandomShardedNegativeSampler(K, sharding, 3, "t", local_sampling=False, flat_negative_format=flat)
            return RandomShardedBatchSampler(pts, ns, shard_bs, bps, seed=4)

        torch.manual_seed(0)
        fn = TransE(flat, 1, sharding, n_rel, 32)
        model = cls(negative_sampler=make_bs().negative_sampler, score_fn=fn,
                    loss_fn=LogSigmoidLoss(margin=3.0, negative_adversarial_sampling=True))
        runner = runtime.inference_model(model, runtime.Options(device_iterations=bps), group=g, device=dev)
        host = make_bs()
        dbs = DeviceBatchSampler(make_bs(), dev, shards=[r])
        for step in range(2):
            full = host[[0]]
            own = dbs.sample()
            res_full = runner(**{k: v.flatten(end_dim=1) for k, v in full.items()})
            res_own = runner(**{k: v.flatten(end_dim=1) for k, v in own.items()})
            for k in res_full:
                assert torch.equal(res_full[k], res_own[k]), (name, step, k)
                out[f"{name}_{step}_{k}"] = res_own[k].float().cpu().numpy()
    np.savez(os.path.join(out_dir, f"sampler_{r}.npz"), **out)


def checkpoint(out_dir: str) -> None:
    """CPU, gloo: every rank hosts one shard, saves it, loads it into a differently initialised model; the files
    of the ranks do not overlap and the replicated tables are written once (by the host of shard 0)."""
    from besskge import checkpoint as ckpt
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import ComplEx
    from besskge.sharding import Sharding

    g = make_group()
    n, r = g.n_shard, g.rank
    sharding = Sharding.create(700, n, seed=4)

    def build(seed: int):
        torch.manual_seed(seed)
        fn = ComplEx(False, sharding, 9, 8, shards=[r])  # this rank's slice only
        ns = RandomShardedNegativeSampler(4, sharding, 1, "t", local_sampling=False, flat_negative_format=False)
        m = EmbeddingMovingBessKGE(ns, fn, LogSigmoidLoss(margin=1.0, negative_adversarial_sampling=False))
        m.attach(g, {r: 0})
        return m

    a = build(100 + r)
    d = os.path.join(out_dir, "ckpt")
    ckpt.save_checkpoint(a, d, chunk_bytes=1 << 10)
    dist.barrier()
    b = build(999)
    assert not torch.equal(b.score_fn.entity_embedding, a.score_fn.entity_embedding)
    ckpt.load_checkpoint(b, d, chunk_bytes=1 << 10)
    assert torch.equal(b.score_fn.entity_embedding, a.score_fn.entity_embedding)
    # the replicated relation table is rank 0's (every rank built its own with a different seed)
    (rel0,) = g.all_gather([a.score_fn.relation_embedding.detach().clone()])
    assert torch.equal(b.score_fn.relation_embedding.detach(), rel0[0])
    np.savez(os.path.join(out_dir, f"checkpoint_{r}.npz"), files=np.array(sorted(os.listdir(d))))


def multidevice(out_dir: str) -> None:
    """ONE process, n GPUs (`MultiDeviceGroup`, `bess_comm_init_all`): the reference's call shape - the full
    `[bps * n_shard, ...]` batch in, stacked outputs out (reference tests/test_bess.py:122-150) - on golden cases;
    BESS_USE_PLANS=1: every replica's step is a recorded plan run from its own host thread."""
    from besskge import runtime
    from besskge.collectives import MultiDeviceGroup
    from test_hip_parity import build_model
    from test_oracle import load_bess_case

    n = int(os.environ["BESS_N_DEVICES"])
    devices = [torch.device("cuda", i) for i in range(n)]
    plans = os.environ.get("BESS_USE_PLANS", "0") == "1"
    out = {}
    for case in [c for c in os.environ["BESS_CASES"].split(",") if c]:
        c = load_bess_case(case)
        assert c["meta"]["n_shard"] == n
        bps = c["meta"]["bps"]
        keys = ("head", "relation", "tail", "negative", "negative_mask")
        batch = {k: c["batch"][k].flatten(end_dim=1) for k in keys if k in c["batch"]}
        model = build_model(c, torch.device("cpu"))
        group = MultiDeviceGroup(devices)
        runner = runtime.inference_model(model, runtime.Options(device_iterations=bps), group=group)
        if c["net"] is not None:
            for rep in runner.replicas:
                rep.train()
        res = runner(**batch)
        for k, v in res.items():
            out[f"{case}_fwd_{k}"] = v.float().cpu().numpy()
        runner.close()
        if case.startswith("tr_"):
            model = build_model(c, torch.device("cpu"))
            group = MultiDeviceGroup(devices)
            lr = 0.125
            runner = runtime.training_model(model, runtime.Options(device_iterations=1, use_plans=plans), runtime.SGD(lr=lr),
                                            group=group)
            try:
                res = runner(**{k: v[: n] for k, v in batch.items()})
            except RuntimeError as e:
                if not plans or "not made of library calls only" not in str(e):
                    raise
                out[f"{case}_refused"] = np.array(1)  # (a step with torch operators in it: eager is the answer)
                runner.close()
                continue
            runner.sync_to_model()
            out[f"{case}_train_loss"] = res["loss"].float().cpu().numpy()
            out[f"{case}_train_entity"] = model.score_fn.entity_embedding.detach().float().cpu().numpy()
            out[f"{case}_train_relation"] = model.score_fn.relation_embedding.detach().float().cpu().numpy()
            runner.close()
    np.savez(os.path.join(out_dir, "multidevice.npz"), **out)


def main() -> None:
    mode, out_dir = sys.argv[1], sys.argv[2]
    if mode == "multidevice":  # one process: no process group
        multidevice(out_dir)
        return
    # gloo: several ranks share the box's one GPU (host-staged collectives); nccl (= RCCL): one rank
    # per GPU - with a single GPU that is world_size 1, which still sends every collective through RCCL
    backend = os.environ.get("BESS_DIST_BACKEND", "gloo")
    if backend == "nccl":
        torch.cuda.set_device(device())
        dist.init_process_group("nccl", device_id=device())
    elif backend == "native":
        torch.cuda.set_device(device())
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("gloo")
    try:
        {"routing": routing, "bess": bess, "topk": topk, "sampler": sampler, "checkpoint": checkpoint}[mode](out_dir)
        dist.barrier()
    finally:
        for g in _GROUPS:
            if hasattr(g, "close"):
                g.close()  # recorded steps first, then the communicator (NativeGroup.close)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
