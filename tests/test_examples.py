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
