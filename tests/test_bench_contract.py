"""bench.py keeps the driver's contract: one JSON line with the agreed keys, at
N = 1 and (rehearsed with gloo, two ranks sharing the GPU) at N = 2."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def run(cmd, env=None):
    res = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["score", "train"])
def test_bench_single_gpu(mode):
    d = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--c4-max-s", "4096", "--mode", mode])
    assert KEYS <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert d["unit"] == "triples/s" and d["scaling"] == "weak" and d["vs_baseline"] is None
    r = d["roofline"]
    # the 192 MB C2 table lives in the Infinity Cache and is labelled so; the HBM figure is roofline_hbm
    assert r["bound"] == "infinity-cache" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1.5
    assert "workload" in d["config"]
    h = d["roofline_hbm"]
    assert h["bound"] == "hbm" and h["launches_timed"] >= 10 and 0 < h["frac"] < 1.0
    assert h["achieved"] < r["achieved"] * 1.05  # HBM cannot beat the cache-resident launch
    c4 = d["c4"]
    assert c4["dtype"] == "f16" and c4["n_gpus"] == 1 and len(c4["sweep"]) == 3
    for pt in c4["sweep"]:
        assert pt["value"] > 0 and "eager_ms_per_step" in pt and "graph_ms_per_step" in pt, pt
    if mode == "score":
        t = d["train_step"]
        assert t["ms_per_step"] > 0 and t["adamw_ms_per_step"] > 0 and "graph_ms_per_step" in t, t


def test_bench_two_ranks_rehearsal():
    env = dict(os.environ, BESS_BENCH_BACKEND="gloo")
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
             "127.0.0.1", "--master-port", "29577", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
             "--c4-max-s", "4096"], env)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0
    assert "ScoreMoving" in d["config"]["workload"]
    assert d["train_step"]["ms_per_step"] > 0  # the headline workload as a ScoreMoving training step
    c4 = d["c4"]  # north_star's scaling workload is part of the line at every N
    assert c4["n_gpus"] == 2 and all(pt["value"] > 0 for pt in c4["sweep"]), c4


def test_bench_two_ranks_rehearsal_train_mode():
    """--mode train with one process per GPU (ScoreMoving training step)."""
    env = dict(os.environ, BESS_BENCH_BACKEND="gloo")
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
             "127.0.0.1", "--master-port", "29579", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
             "--mode", "train", "--no-extra-legs"], env)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["mode"] == "train"


def test_bench_c4_workload_single_gpu():
    """--workload c4: north_star's scaling workload is the line itself (f16, VALU roofline, oracle baseline)."""
    d = run([sys.executable, "bench.py", "--workload", "c4", "--c4-point", "512,32", "--steps", "16", "--warmup", "8"])
    assert KEYS <= set(d)
    assert d["dtype"] == "f16" and d["n_gpus"] == 1 and d["value"] > 0 and d["ms_per_step"] > 0
    assert "wikikg2" in d["config"]["workload"] and d["config"]["shard_bs"] == 512
    assert d["config"]["negatives_per_triple"] == 512 + 32
    r = d["roofline"]
    assert r["bound"] == "valu" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    pt = d["c4"]["sweep"][0]
    assert "eager_ms_per_step" in pt and "graph_ms_per_step" in pt and d["ms_per_step"] == pt["ms_per_step"]
    assert abs(d["value"] - 512 * (1 + 544) / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1


def test_bench_c4_workload_two_ranks_rehearsal():
    """The same line at N = 2 (gloo, ranks sharing the GPU) with the exchange microbenchmark; the hipGraph variant
    is opt-in at N > 1 and absent by default."""
    env = dict(os.environ, BESS_BENCH_BACKEND="gloo")
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
             "127.0.0.1", "--master-port", "29581", "bench.py", "--gpus", "2", "--workload", "c4", "--c4-point", "512,32",
             "--steps", "8", "--warmup", "8"], env)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["dtype"] == "f16" and d["value"] > 0
    assert d["config"]["negatives_per_triple"] == 512 + 2 * 32 and d["config"]["launch"] == "eager"
    x = d["xgmi"]
    assert [r["bytes_per_peer"] for r in x["sizes"]] == [49_152, 344_064, 2_752_512, 33_554_432]
    assert all(r["gbs_per_gpu"] > 0 for r in x["sizes"]) and x["peak_per_gpu_gbs"] == 7 * 153.0
    assert "cpu_baseline" not in d


def test_bench_c4_workload_one_rank_rccl_with_recorded_collectives():
    """One-rank RCCL rehearsal of the multi-GPU path with --c4-graph: the collectives of the C4 step are recorded
    into the hipGraph together with the kernels (NativeGroup), and the line says which launch form it reports."""
    env = dict(os.environ, BESS_BENCH_REHEARSE_DIST="1", MASTER_PORT="29583")
    d = run([sys.executable, "bench.py", "--workload", "c4", "--c4-point", "512,32", "--steps", "16", "--warmup", "8",
             "--c4-graph", "--comm", "native"], env)
    assert d["config"]["collectives"].startswith("native")
    pt = d["c4"]["sweep"][0]
    assert "graph_ms_per_step" in pt and "graph_error" not in pt, pt
    assert d["config"]["launch"] in ("eager", "plan", "graph")  # (the fastest of the three is the line)
