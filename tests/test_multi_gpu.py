"""The library's RCCL path with real peers: one process per GPU, started by `torch.distributed.run` BEFORE any GPU
call, `NativeGroup` (bess_comm_init_rank / bess_alltoall / bess_allgather / bess_allreduce_sum_f32 /
bess_pack_exchange on the kernels' stream) - the layout of the reference's distributed tests
(`/root/reference/tests/test_bess.py:122-150`: 4 replicas, every tensor `[bps * n_shard, ...]`).

Every test runs at world = 1 (what a one-GPU box can show: the same worker code, devices picked by LOCAL_RANK,
recorded steps) and at world = 2 / 4 when the box has that many GPUs (`torch.cuda.device_count()` does not touch
the GPU).  The checks are the ones `tests/test_distributed.py` applies to the gloo / one-rank runs: the reference's
stored outputs and autograd gradients (`tests/golden/bess*.npz`).
"""

import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from test_distributed import HERE, _cases, launch

N_GPU = torch.cuda.device_count()
WORLDS = [pytest.param(w, marks=pytest.mark.skipif(N_GPU < w, reason=f"needs {w} GPUs, the box has {N_GPU}"))
          for w in (1, 2, 4)]
pytestmark = pytest.mark.gpu
ENV = {"BESS_DIST_BACKEND": "native", "BESS_DIST_DEVICE": "local_rank"}


@pytest.mark.parametrize("world", WORLDS)
def test_native_collectives_route_between_gpus(world):
    """Block j of an all-to-all goes to rank j; all-gather stacks in rank order; all-reduce sums; pack_exchange
    gathers rows 7 j + r of rank r's table for rank j (`tests/_dist_worker.py: routing`)."""
    out = launch("routing", world, ENV)
    for r in range(world):
        z = np.load(os.path.join(out, f"routing_{r}.npz"))
        assert z["a2a"].shape == (world, 3, 2)
        for j in range(world):
            assert np.all(z["a2a"][j] == 100.0 * j + r)
            # rank j packed row 7 r + j of its table (value 1000 j + row) for this rank
            assert np.all(z["packed"][j] == 1000.0 * j + 7 * r + j), (r, j, z["packed"][j][:, 0])
        assert np.array_equal(z["ag"], np.repeat(np.arange(world, dtype=np.float32)[:, None], 2, axis=1))
        assert np.all(z["ar"] == sum(range(1, world + 1)))
        assert z["packed"].shape == (world, 5, 8)


def _check_bess(out, world, cases):
    from test_oracle import load_bess_case

    per_rank = [np.load(os.path.join(out, f"bess_{r}.npz")) for r in range(world)]
    for case in cases:
        c = load_bess_case(case)
        bps = c["meta"]["bps"]
        ssce = c["loss_name"] == "ssce"
        S = c["outs"]["positive_score"].shape[-1]
        for r in range(world):
            z = per_rank[r]
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


@pytest.mark.parametrize("graphs", [False, True], ids=["eager", "recorded"])
@pytest.mark.parametrize("world", WORLDS)
def test_bess_goldens_over_native_rccl(world, graphs):
    """Forward and one SGD step of both BESS schemes (EmbeddingMoving / ScoreMoving x h / t / ht x flat /
    per-triple, every scorer family with goldens at this n_shard) against the reference's outputs and gradients,
    one process per GPU.  "recorded": `Options.use_graphs` - the step, RCCL send / recv included, is a hipGraph and
    the replayed step must equal the recording one."""
    cases = _cases(world)
    assert cases
    env = dict(ENV, BESS_CASES=",".join(cases), BESS_USE_GRAPHS="1" if graphs else "0")
    out = launch("bess", world, env, timeout=1200)
    _check_bess(out, world, cases)


@pytest.mark.parametrize("world", WORLDS)
def test_query_goldens_over_native_rccl(world):
    """TopKQueryBessKGE / AllScoresBESS (next-1): ids and scores of the reference, final merge by all-to-all."""
    from test_query import load_query_case, query_cases

    specs = [("topk", c) for c in query_cases("topk") if c.endswith(f"_n{world}")]
    assert specs
    out = launch("topk", world, dict(ENV, BESS_CASES=",".join(f"{f}:{c}" for f, c in specs)), timeout=1200)
    for fix, case in specs:
        c = load_query_case(fix, case)
        m = c["meta"]
        for r in range(world):
            z = np.load(os.path.join(out, f"topk_{r}.npz"))
            want_s = c["outs"]["topk_scores"][:, r].numpy().reshape(-1, m["k"])
            np.testing.assert_allclose(z[f"{case}_scores"], want_s, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("world", WORLDS)
def test_bench_c4_recorded_collectives(world):
    """`bench.py --gpus N --workload c4 --c4-graph`: north_star's scaling workload with the exchange recorded into
    the step's hipGraph.  The line must come back (exit code 0, no `graph_abandoned`), with a replayed figure."""
    repo = os.path.dirname(HERE)
    port = 29100 + os.getpid() % 800 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(repo, "bench.py"),
           "--gpus", str(world), "--steps", "16", "--warmup", "8", "--workload", "c4", "--c4-point", "512,32",
           "--c4-graph", "--no-cpu-baseline"]
    if world == 1:
        cmd = [sys.executable, os.path.join(repo, "bench.py"), "--steps", "16", "--warmup", "8", "--workload", "c4",
               "--c4-point", "512,32", "--no-cpu-baseline"]
    env = dict(os.environ, BESS_BENCH_REHEARSE_DIST="1") if world == 1 else dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(port)
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == world and not line.get("graph_abandoned")
    assert line["config"]["collectives"].startswith("native"), line["config"]["collectives"]
    (pt,) = line["c4"]["sweep"]
    assert pt["value"] > 0 and "graph_ms_per_step" in pt, pt


@pytest.mark.parametrize("plans", [False, True], ids=["eager", "plans"])
@pytest.mark.parametrize("world", WORLDS)
def test_one_process_driving_n_devices(world, plans):
    """`MultiDeviceGroup`: ONE host process, n GPUs (`bess_comm_init_all`), one host thread per device - the
    reference's own runtime contract (`tests/test_bess.py:122-150`: every tensor `[bps * n_shard, ...]` in, stacked
    outputs back).  Forward and one SGD step of the golden cases of this n_shard against the reference's outputs and
    gradients; "plans": each replica's step is a recorded plan (`Options.use_plans`) replayed from its thread."""
    from test_oracle import load_bess_case

    cases = _cases(world)
    # (a spread of eight golden cases - training and inference, both schemes, several scorers: every case builds two
    # cliques; all 45 single-shard cases took 52 s per variant of this test)
    train = [c for c in cases if c.startswith("tr_")]
    other = [c for c in cases if not c.startswith("tr_")]
    cases = train[:: max(1, len(train) // 5)][:5] + other[:: max(1, len(other) // 3)][:3]
    assert cases and any(c.startswith("tr_") for c in cases)
    out_dir = tempfile.mkdtemp(prefix="bess_md_")
    env = dict(os.environ, BESS_CASES=",".join(cases), BESS_N_DEVICES=str(world), BESS_USE_PLANS="1" if plans else "0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(HERE, "_dist_worker.py"), "multidevice", out_dir], env=env,
                         capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    z = np.load(os.path.join(out_dir, "multidevice.npz"))
    n_trained = 0
    for case in cases:
        c = load_bess_case(case)
        bps = c["meta"]["bps"]
        ssce = c["loss_name"] == "ssce"
        S = c["outs"]["positive_score"].shape[-1]
        # stacked over micro-batches and replicas, micro-batch-major: [bps, n, ...]
        np.testing.assert_allclose(z[f"{case}_fwd_positive_score"].reshape(bps, world, S),
                                   c["outs"]["positive_score"].numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(z[f"{case}_fwd_negative_score"].reshape(bps, world, S, -1),
                                   c["outs"]["negative_score"].numpy(), rtol=1e-4, atol=2e-3 if ssce else 1e-5)
        if c["loss"] is not None:
            np.testing.assert_allclose(z[f"{case}_fwd_loss"].reshape(bps, world), c["outs"]["loss"].numpy(), rtol=1e-4, atol=1e-4)
        if case.startswith("tr_") and f"{case}_refused" not in z.files:
            n_trained += 1
            lr = 0.125
            np.testing.assert_allclose(z[f"{case}_train_loss"].reshape(world), c["outs"]["loss"][0].numpy(), rtol=1e-4, atol=1e-4)
            np.testing.assert_allclose(z[f"{case}_train_entity"], (c["table"] - lr * c["grads"]["entity"]).numpy(),
                                       rtol=1e-4, atol=2e-5)
            np.testing.assert_allclose(z[f"{case}_train_relation"], (c["rel"] - lr * c["grads"]["relation"].sum(0)).numpy(),
                                       rtol=1e-4, atol=2e-5)
    assert n_trained > 0
