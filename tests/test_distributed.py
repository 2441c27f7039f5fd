"""N > 1 path, one process per shard (torch.distributed).

CPU part (`-m "not gpu"`): world_size 2 and 3 over gloo - the two BESS
collectives of `DistributedGroup` have the semantics of the reference's
`all_to_all_single_cross_replica` / `all_gather_cross_replica`
(reference bess.py:14-19: block j goes to replica j; gather stacks in rank
order), and replicated-parameter gradients are summed.

GPU part (`-m gpu`): world_size 2 and 4, all ranks sharing the box's single GPU
(gloo with host staging), run the *distributed* BessKGE forward and train step
through the HIP kernels and must reproduce the reference's outputs stored in
tests/golden/bess.npz - the same goldens the single-process lock-step runs match.
"""

import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "_dist_worker.py")


def launch(mode, world, extra_env=None, timeout=600):
    out_dir = tempfile.mkdtemp(prefix="bess_dist_")
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault("OMP_NUM_THREADS", "1")
    port = 29500 + (os.getpid() % 2000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, mode, out_dir]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return out_dir


def test_checkpoint_every_rank_its_own_shard():
    """Per-shard streaming checkpoints with one process per shard (gloo, CPU): besskge/checkpoint.py."""
    out = launch("checkpoint", 2)
    files = set(np.load(os.path.join(out, "checkpoint_0.npz"))["files"].tolist())
    assert {"entity_shard0.npy", "entity_shard1.npy", "relation.npy", "dense.pt",
            "entity_shard0.meta.json", "entity_shard1.meta.json", "relation.meta.json"} <= files
    assert not [f for f in files if ".tmp" in f]


@pytest.mark.parametrize("world", [2, 3])
def test_collectives_gloo(world):
    out = launch("routing", world)
    for r in range(world):
        z = np.load(os.path.join(out, f"routing_{r}.npz"))
        # all_to_all: block j of the result came from rank j and is rank j's block r
        for j in range(world):
            assert np.all(z["a2a"][j] == 100.0 * j + r)
        assert z["a2a"].shape == (world, 3, 2)
        # all_gather: stacked in rank order
        assert np.array_equal(z["ag"], np.repeat(np.arange(world, dtype=np.float32)[:, None], 2, axis=1))
        assert np.all(z["ar"] == sum(range(1, world + 1)))
        if world in (2, 3):
            for j in range(world):
                assert np.array_equal(z["ag_i"][j], np.arange(6, dtype=np.int32).reshape(world, -1) + 10 * j)


@pytest.mark.gpu
def test_collectives_native_rccl_single_rank():
    """The library's own RCCL entry points (bess_comm_init_rank, bess_alltoall, bess_allgather,
    bess_allreduce_sum_f32, bess_pack_exchange) through NativeGroup on a one-rank communicator -
    all a 1-GPU box can show of them; routing over several ranks is covered by the gloo cases
    (same block layout by construction: block p <-> rank p)."""
    out = launch("routing", 1, {"BESS_DIST_BACKEND": "native"})
    z = np.load(os.path.join(out, "routing_0.npz"))
    assert np.all(z["a2a"][0] == 0.0) and z["a2a"].shape == (1, 3, 2)
    assert np.array_equal(z["ag"], np.zeros((1, 2), dtype=np.float32))
    assert np.all(z["ar"] == 1.0)
    assert np.array_equal(z["ag_i"][0], np.arange(6, dtype=np.int32).reshape(1, -1))
    assert z["packed"].shape == (1, 5, 8) and np.all(z["packed"] == 0.0)  # row 7 * 0 + 0 of table 0


def _cases(n):
    from test_oracle import bess_cases

    return [c for c in bess_cases() if c.endswith(f"_n{n}")]


@pytest.mark.gpu
@pytest.mark.parametrize("world,backend", [(1, "nccl"), (1, "native"), (2, "gloo"), (4, "gloo")])
def test_distributed_bess_golden(world, backend):
    """(1, nccl): the single-shard goldens through DistributedGroup over a one-rank RCCL group - the
    forward and training-step collectives as RCCL sees them (the box has one GPU); (1, native): the
    same through NativeGroup, i.e. the library's own bess_comm_* / bess_alltoall ... entry points."""
    from test_oracle import load_bess_case

    cases = _cases(world)
    assert cases
    out = launch("bess", world, {"BESS_CASES": ",".join(cases), "BESS_DIST_BACKEND": backend}, timeout=900)
    per_rank = [np.load(os.path.join(out, f"bess_{r}.npz")) for r in range(world)]
    for case in cases:
        c = load_bess_case(case)
        bps = c["meta"]["bps"]
        ssce = c["loss_name"] == "ssce"
        for r in range(world):
            z = per_rank[r]
            S = c["outs"]["positive_score"].shape[-1]
            np.testing.assert_allclose(z[f"{case}_fwd_positive_score"].reshape(bps, S),
                                       c["outs"]["positive_score"][:, r].numpy(), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(z[f"{case}_fwd_negative_score"].reshape(bps, S, -1),
                                       c["outs"]["negative_score"][:, r].numpy(), rtol=1e-4, atol=2e-3 if ssce else 1e-5)
            if c["loss"] is not None:
                np.testing.assert_allclose(z[f"{case}_fwd_loss"].reshape(bps), c["outs"]["loss"][:, r].numpy(),
                                           rtol=1e-4, atol=1e-4)
            if case.startswith("tr_"):
                lr = 0.125
                np.testing.assert_allclose(z[f"{case}_train_loss"].reshape(()), c["outs"]["loss"][0, r].numpy(),
                                           rtol=1e-4, atol=1e-4)
                np.testing.assert_allclose(z[f"{case}_train_entity"][0],
                                           (c["table"][r] - lr * c["grads"]["entity"][r]).numpy(), rtol=1e-4, atol=2e-5)
                np.testing.assert_allclose(z[f"{case}_train_relation"],
                                           (c["rel"] - lr * c["grads"]["relation"].sum(0)).numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("world,backend", [(1, "nccl"), (1, "native"), (2, "gloo"), (4, "gloo")])
def test_distributed_queries_golden(world, backend):
    """TopKQueryBessKGE / AllScoresBESS, one process per shard, vs the reference's outputs."""
    from test_query import load_query_case, query_cases

    specs = [("topk", c) for c in query_cases("topk") if c.endswith(f"_n{world}")]
    specs += [("allscores", c) for c in query_cases("allscores") if c.endswith(f"_n{world}")]
    assert specs
    out = launch("topk", world, {"BESS_CASES": ",".join(f"{f}:{c}" for f, c in specs), "BESS_DIST_BACKEND": backend},
                 timeout=900)
    per_rank = [np.load(os.path.join(out, f"topk_{r}.npz")) for r in range(world)]
    for fix, case in specs:
        c = load_query_case(fix, case)
        m = c["meta"]
        for r in range(world):
            z = per_rank[r]
            if fix == "topk":
                want_s = c["outs"]["topk_scores"][:, r].numpy().reshape(-1, m["k"])
                want_i = c["outs"]["topk_global_id"][:, r].numpy().reshape(-1, m["k"])
                np.testing.assert_allclose(z[f"{case}_scores"], want_s, rtol=1e-4, atol=1e-4)
                # ids are exact wherever the order is determined: a mismatch is only accepted inside a run of
                # (numerically) equal scores, or in the last slot (which may tie with the first entry cut off)
                tol = 1e-4 * np.maximum(1.0, np.abs(want_s))
                tie = np.zeros_like(want_s, dtype=bool)
                close = np.abs(np.diff(want_s, axis=1)) <= tol[:, 1:]
                tie[:, 1:] |= close
                tie[:, :-1] |= close
                tie[:, -1] = True
                bad = (z[f"{case}_ids"] != want_i) & ~tie
                assert not bad.any(), f"{case}: {int(bad.sum())} top-k ids differ outside ties"
            else:
                want = c["outs"]["scores"][:, :, r].numpy()  # [bps, n_step, shard_bs, n * ws]
                got = z[f"{case}_scores"]  # [n_step, bps * shard_bs, n * ws]
                got = got.reshape(want.shape[1], want.shape[0], *want.shape[2:]).transpose(1, 0, 2, 3)
                np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
        if fix == "allscores":
            # rank-counting mode, one process per shard == all shards in this process (same kernels, same shapes)
            counts, pos = _single_process_rank_counts(c, world)
            for r in range(world):
                assert np.array_equal(per_rank[r][f"{case}_rank_counts"], counts[r]), f"{case}: rank counts of shard {r}"
                np.testing.assert_array_equal(per_rank[r][f"{case}_rank_pos"], pos[r])
            assert sum(int(x[:, 0].sum()) for x in counts) > 0


def _single_process_rank_counts(c, n):
    import torch
    from besskge import runtime
    from besskge.bess import AllScoresBESS
    from besskge.sharding import Sharding
    from test_hip_parity import make_scorer
    from test_query import candidate_sampler, rank_inputs

    m = c["meta"]
    bps = m["bps"]
    dev = torch.device("cuda", 0)
    fn = make_scorer(c["scorer"], m["norm"], bool(m["flat"]), m["n_rel"], m["d"], c["table"], c["rel"], dev,
                     sharding=Sharding.create(m["n_entity"], n, seed=1234))
    model = AllScoresBESS(candidate_sampler(c), fn, window_size=m["window"])
    runner = runtime.inference_model(model, runtime.Options(device_iterations=bps), device=dev)
    known = "tail" if c["scheme"] == "h" else "head"
    inp = {k: c["batch"][k].flatten(end_dim=1) for k in ("relation", known)}
    truth, filt = rank_inputs(c, bps * n, int(inp["relation"].shape[1]))
    res = runner(step=torch.zeros((bps * n, 1), dtype=torch.int32), **inp, rank_truth=truth, rank_filter=filt)
    bs = int(inp["relation"].shape[1])
    counts = res["counts"].cpu().numpy().reshape(bps, n, bs, 2)
    pos = res["pos_score"].float().cpu().numpy().reshape(bps, n, bs)
    return ([counts[:, r].reshape(-1, 2) for r in range(n)], [pos[:, r].reshape(-1) for r in range(n)])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_device_sampler_own_rows(world):
    """Ranks that sample only their own slice on the device (no index broadcast)
    get the results of the run fed with the host sampler's full batch."""
    out = launch("sampler", world, timeout=900)
    z = [np.load(os.path.join(out, f"sampler_{r}.npz")) for r in range(world)]
    for r in range(world):
        assert len(z[r].files) > 0
        for k in z[r].files:
            assert np.isfinite(z[r][k]).all()


@pytest.mark.gpu
def test_bench_distributed_path_over_rccl_single_rank():
    """The multi-GPU leg of bench.py (ScoreMoving, DistributedGroup, pipelined begin / finish) on a
    one-rank RCCL process group: every collective of the N > 1 path goes through RCCL with the
    tensors the real run sends (dtypes, contiguity, side streams), which the gloo rehearsals cannot
    show.  The line must be the same workload at nearly the single-process rate."""
    import json

    repo = os.path.dirname(HERE)
    env = dict(os.environ, BESS_BENCH_REHEARSE_DIST="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29400 + os.getpid() % 500))
    res = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "10", "--warmup", "3", "--c4-max-s", "4096",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and "ScoreMoving" in line["config"]["workload"]
    assert line["value"] > 1e9 and line["roofline"]["frac"] > 0.3
    # the collectives went through the library's own RCCL entry points (self-check passed), and the
    # C4 training leg ran on them too - eager and captured into a hipGraph together with its collectives
    assert line["config"]["collectives"].startswith("native"), line["config"]["collectives"]
    for pt in line["c4"]["sweep"]:
        assert pt["value"] > 0 and "graph_ms_per_step" in pt, pt


@pytest.mark.gpu
def test_comm_init_all_single_device_clique():
    """`bess_comm_init_all` (one process driving the GPUs it is given; here the one GPU of the box): every
    collective of the library on the communicator it returns, then `bess_comm_destroy`.  In a subprocess: a
    communicator is process-wide RCCL state."""
    code = r'''
import sys, torch
sys.path[:0] = [sys.argv[1]]
from besskge import _native as nat
dev = torch.device("cuda", 0)
(comm,) = nat.Communicator.init_all([dev])
assert comm.info() == (1, 0, 0), comm.info()
x = torch.arange(24, dtype=torch.float32, device=dev).reshape(1, 6, 4)
assert torch.equal(comm.all_to_all(x), x)
assert torch.equal(comm.all_gather(x[0]), x)
y = x.clone()
assert torch.equal(comm.all_reduce_sum_(y), x)
table = torch.randn(50, 8, device=dev).half()
idx = torch.randint(50, (1, 7), dtype=torch.int32, device=dev)
send, recv = comm.pack_exchange(table, idx)
torch.cuda.synchronize()
assert torch.equal(recv[0], table[idx[0].long()]) and torch.equal(send, recv)
# the same collectives recorded into a hipGraph with a kernel between them, replayed with new data
static = torch.zeros(1, 6, 4, device=dev)
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    comm.all_to_all(static)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = comm.all_reduce_sum_(comm.all_to_all(static) * 2.0)
for k in range(3):
    static.fill_(float(k + 1))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, torch.full_like(out, 2.0 * (k + 1))), (k, out)
# a graph that recorded the communicator's collectives goes first: ncclCommDestroy waits for it to be destroyed -
# the communicator knows it was recorded and refuses (an error, not a process that never ends)
try:
    comm.close()
    raise SystemExit("close() with a recorded graph alive must refuse")
except RuntimeError as e:
    assert "hipGraph" in str(e), e
g.reset()
del g, out
torch.cuda.synchronize()
comm.graphs_released()
comm.close()
comm.close()  # idempotent
print("ok")
'''
    res = subprocess.run([sys.executable, "-c", code, os.path.join(os.path.dirname(HERE), "bess-kge_amd")], capture_output=True,
                         text=True, timeout=120, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


_RECORDED_RUNNER = r'''
import sys, torch
sys.path[:0] = [sys.argv[1], sys.argv[2]]
import numpy as np
from besskge import _native as nat, runtime
from besskge.bess import EmbeddingMovingBessKGE
from besskge.collectives import NativeGroup
from besskge.loss import LogSigmoidLoss
from besskge.negative_sampler import RandomShardedNegativeSampler
from besskge.scoring import TransE
from besskge.sharding import Sharding
dev = torch.device("cuda", 0)
torch.manual_seed(0)
group = NativeGroup(dev, unique_id=nat.comm_unique_id(), world=1, rank=0)
sharding = Sharding.create(500, 1, seed=3)
fn = TransE(True, 1, sharding, 5, 32)
ns = RandomShardedNegativeSampler(16, sharding, 1, "t", local_sampling=False, flat_negative_format=True)
model = EmbeddingMovingBessKGE(ns, fn, LogSigmoidLoss(margin=3.0, negative_adversarial_sampling=True))
runner = runtime.training_model(model, runtime.Options(use_graphs=True), runtime.SGD(lr=0.01), group=group, device=dev)
rng = np.random.default_rng(0)
batch = dict(head=torch.from_numpy(rng.integers(500, size=(1, 1, 64)).astype(np.int32)),
             relation=torch.from_numpy(rng.integers(5, size=(1, 1, 64)).astype(np.int32)),
             tail=torch.from_numpy(rng.integers(500, size=(1, 1, 64)).astype(np.int32)),
             negative=torch.from_numpy(rng.integers(500, size=(1, 1, 1, 16)).astype(np.int32)))
losses = [float(runner(**batch)["loss"]) for _ in range(4)]   # record + three replays
torch.cuda.synchronize()
assert group.comm._recorded, "the step's all-reduce was recorded with the communicator"
assert all(np.isfinite(losses)) and losses[3] != losses[0], losses
MODE = sys.argv[3]
if MODE == "close":
    group.close()          # destroys the runner's graphs, then the communicator
    assert not runner._graphs
    group2 = None
elif MODE == "drop":
    del runner, model, group   # any order the collector likes
print("ok", flush=True)
# MODE == "exit": leave with runner, graphs, group and communicator all alive
'''


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["exit", "close", "drop"])
def test_process_ends_with_recorded_collectives_alive(mode):
    """Round 3's hang: `ncclCommDestroy` waits for hipGraphs that recorded the communicator's collectives, and
    `Communicator.__del__` called it at interpreter exit with a `Runner(use_graphs=True)` still alive.  Now the
    group knows the graphs recorded over it (`ReplicaGroup.register_graph_cache`), `NativeGroup.close()` destroys
    them first, and no finaliser destroys a communicator it cannot know to be free.  A NativeGroup of one rank,
    a recorded training step (the relation gradient's all-reduce is a collective), three replays, then the
    process ends WITHOUT deleting anything ("exit"), after `group.close()`, or by dropping the objects."""
    repo = os.path.dirname(HERE)
    res = subprocess.run([sys.executable, "-c", _RECORDED_RUNNER, os.path.join(repo, "bess-kge_amd"), repo, mode],
                         capture_output=True, text=True, timeout=60, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("opt_name", ["sgd", "adam"])
def test_gradient_accumulation_one_process_per_shard(opt_name):
    """Options.gradient_accumulation with one process per shard (gloo, ranks sharing the GPU): two micro-batches, one
    update - the relation gradient all-reduced once, every shard's rows updated once - equals the oracle's step on
    the summed loss, like the single-process form (tests/test_accumulation.py)."""
    from oracle import kge
    from test_oracle import load_bess_case, step_batch

    cases = ["tr_EM_TransE1_ht_flat_n2", "tr_EM_ComplEx0_ht_pt_n2", "tr_SM_TransE1_t_pt_n2"]
    out = launch("bess", 2, {"BESS_CASES": ",".join(cases), "BESS_DIST_BACKEND": "gloo", "BESS_ACCUMULATE": opt_name},
                 timeout=900)
    per_rank = [np.load(os.path.join(out, f"bess_{r}.npz")) for r in range(2)]
    for case in cases:
        c = load_bess_case(case)
        t0 = c["table"].clone().requires_grad_(True)
        r0 = c["rel"].clone().requires_grad_(True)
        total = 0.0
        for it in range(2):
            res = kge.bess_step(c["spec"], c["model_cls"], t0, r0, step_batch(c["batch"], it), c["loss"])
            total = total + torch.stack(res["loss"]).sum()
        total.backward()
        opt = torch.optim.Adam([t0, r0], lr=0.01) if opt_name == "adam" else torch.optim.SGD([t0, r0], lr=0.125)
        opt.step()
        for r in range(2):
            got = torch.from_numpy(per_rank[r][f"{case}_acc_entity"])[0]
            grad = t0.grad[r]
            solid = grad.abs() > 1e-4 if opt_name == "adam" else torch.ones_like(grad, dtype=torch.bool)
            tol = dict(rtol=2e-3, atol=5e-5) if opt_name == "adam" else dict(rtol=1e-4, atol=2e-5)
            torch.testing.assert_close(got[solid], t0.detach()[r][solid], **tol)
            got_rel = torch.from_numpy(per_rank[r][f"{case}_acc_relation"])
            solid = r0.grad.abs() > 1e-4 if opt_name == "adam" else torch.ones_like(r0.grad, dtype=torch.bool)
            torch.testing.assert_close(got_rel[solid], r0.detach()[solid], **tol)
