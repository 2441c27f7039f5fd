"""The end-to-end example (device samplers -> fused training step -> all-entity
evaluation) learns the synthetic graph: a functional check of the whole path."""

import os
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("argv", [["--steps", "200"], ["--steps", "150", "--n-shard", "2", "--scorer", "ComplEx"]])
def test_train_and_evaluate_example(argv):
    sys.path.insert(0, os.path.join(REPO, "examples"))
    try:
        import train_and_evaluate
    finally:
        sys.path.pop(0)
    res = train_and_evaluate.main(argv)
    assert res["losses"][-1] < 0.5 * res["losses"][0]
    # 2000 candidates: chance level of hits@10 is 0.005
    assert res["hits@10"] > 0.2, res
    assert res["mrr"] > 0.1, res


def test_multi_process_example_learns():
    """One process per shard (2 ranks sharing the GPU, gloo): own-slice device sampling,
    distributed training step and distributed top-k evaluation."""
    import re
    import subprocess

    env = dict(os.environ, BESS_BACKEND="gloo", OMP_NUM_THREADS="1")
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "examples", "train_multi_gpu.py"),
           "--steps", "150"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    m = re.search(r"mrr ([0-9.]+)\s+hits@10 ([0-9.]+)", res.stdout)
    assert m, res.stdout[-2000:]
    assert float(m.group(2)) > 0.2 and float(m.group(1)) > 0.1


def test_biokg_recipe_with_gradient_accumulation():
    """Notebook 1's recipe (4 shards, device_iterations 8 x gradient accumulation 6, RotatE, AdamW, typed
    validation candidates through ScoreMoving) learns a synthetic typed graph."""
    sys.path.insert(0, os.path.join(REPO, "examples"))
    try:
        import biokg_recipe
    finally:
        sys.path.pop(0)
    res = biokg_recipe.main(["--epochs", "40", "--lr", "0.02"])
    assert res["losses"][-1] < 0.5 * res["losses"][0], res
    # 100 candidates of the right type per side: chance is hits@10 = 0.1; the heads of the synthetic graph are
    # random (only tail prediction can be learnt), so ~0.4 over both sides is what there is to find
    assert res["hits@10"] > 0.3 and res["mrr"] > 0.15, res
