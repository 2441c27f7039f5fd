"""Step plans (`Options.use_plans`, csrc/plan.hip): a step recorded as the list of the library's own calls and
replayed from C - the same tables and outputs as the eager step, collectives included, and a refusal for steps that
are not made of library calls only.  The reference's step runs without Python once PopTorch has compiled it
(`/root/reference/besskge/bess.py:322-468` under `poptorch.trainingModel`)."""

import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _model(kind, dev):
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import LogSigmoidLoss, SampledSoftmaxCrossEntropyLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import ComplEx, RotatE, TransE
    from besskge.sharding import Sharding

    M = 20_000
    sharding = Sharding.create(M, 1, seed=0)
    torch.manual_seed(2)
    if kind == "c4":  # the wikikg2 recipe: fp16 TransE, flat shared negatives, augmentation, sampled softmax
        fn = TransE(True, 1, sharding, 50, 256, device=dev, dtype=torch.float16)
        ns = RandomShardedNegativeSampler(32, sharding, 0, "t", local_sampling=False, flat_negative_format=True)
        model = EmbeddingMovingBessKGE(ns, fn, SampledSoftmaxCrossEntropyLoss(n_entity=M), augment_negative=True)
        S, B, K = 512, 1, 32
    elif kind == "c2":  # per-triple negatives of the own shard: fused forward, segmented reduction, side-stream index
        fn = ComplEx(False, sharding, 50, 64, device=dev)
        ns = RandomShardedNegativeSampler(64, sharding, 0, "t", local_sampling=False, flat_negative_format=False)
        model = EmbeddingMovingBessKGE(ns, fn, LogSigmoidLoss(margin=6.0, negative_adversarial_sampling=True))
        S, B, K = 1024, 1024, 64
    else:  # "ht": two groups, head and tail corruption, shared negatives, fp32
        fn = RotatE(True, 1, sharding, 50, 32, device=dev)
        ns = RandomShardedNegativeSampler(48, sharding, 0, "ht", local_sampling=False, flat_negative_format=True)
        model = EmbeddingMovingBessKGE(ns, fn, LogSigmoidLoss(margin=3.0, negative_adversarial_sampling=False))
        S, B, K = 128, 2, 48
    rng = np.random.default_rng(3)
    batches = []
    for _ in range(4):
        b = dict(head=rng.integers(M, size=(1, 1, S)), relation=rng.integers(50, size=(1, 1, S)),
                 tail=rng.integers(M, size=(1, 1, S)), negative=rng.integers(M, size=(1, 1, B, K)))
        batches.append({k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in b.items()})
    return model, batches


@pytest.mark.parametrize("kind,opt_name", [("c4", "sgd"), ("c4", "sgdm"), ("c4", "adam"), ("c2", "sgd"), ("c2", "adamw")])
def test_replayed_plan_follows_the_eager_trajectory(dev, kind, opt_name):
    from besskge import runtime

    out = {}
    for plans in (False, True):
        model, batches = _model(kind, dev)
        opt = dict(sgd=runtime.SGD(lr=0.02), sgdm=runtime.SGD(lr=0.02, momentum=0.9), adam=runtime.Adam(lr=0.01),
                   adamw=runtime.Adam(lr=0.01, weight_decay=0.01))[opt_name]
        runner = runtime.training_model(model, runtime.Options(use_plans=plans), opt, device=dev)
        losses = [float(runner(**b)["loss"]) for b in batches]
        torch.cuda.synchronize()
        out[plans] = (model.score_fn.entity_embedding.detach().float().clone(),
                      model.score_fn.relation_embedding.detach().float().clone(), losses)
        if plans:
            (calls,) = runner.plan_calls().values()
            assert len(calls) >= 4 and all(c.startswith("bess_") for c in calls)
            if kind == "c4":
                assert "bess_direct_update" in calls and "bess_neg_score_shared_bwd_parts" in calls, calls
            if kind == "c2":
                assert "bess_build_segment_index" in calls, calls
    np.testing.assert_allclose(out[True][2], out[False][2], rtol=2e-3)
    if opt_name.startswith("adam"):  # (a gradient that cancels to ~0 steps +-lr: sign by the order of the fp32 additions)
        # - with the sign-valued gradients of the p = 1 distance on an fp16 shard that is a few per cent of the touched
        # elements over four steps; a wrong step count or a lost moment would move every touched element
        off = (out[True][0] - out[False][0]).abs()
        assert float((off > 2e-3).float().mean()) < (0.04 if kind == "c4" else 0.01)
        assert float(off.max()) <= 4 * 2 * 0.01 * 1.05
    elif kind == "c4":
        # fp16 shard, p = 1: an element that rounds the other way after step 1 (the fp32 sums differ in their last
        # bits) turns sgn(q16 - e) at its near-ties in the later steps - a handful of elements of 5 M move by lr x O(1)
        for a, b, most in zip(out[True][:2], out[False][:2], (2e-4, 2e-3)):  # (the relation table has 12,800 elements)
            off = (a - b).abs()
            assert float((off > 3e-3).float().mean()) < most, float((off > 3e-3).float().mean())
    else:
        torch.testing.assert_close(out[True][0], out[False][0], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(out[True][1], out[False][1], rtol=1e-4, atol=1e-5)


def test_plan_of_a_scoring_call_and_of_several_device_iterations(dev):
    """Scoring calls replay too; `device_iterations` micro-batches of a training call are ONE plan."""
    from besskge import runtime

    model, batches = _model("c2", dev)
    want = runtime.inference_model(model, runtime.Options(), device=dev)(**batches[0])
    runner = runtime.inference_model(model, runtime.Options(use_plans=True), device=dev)
    for _ in range(2):
        got = runner(**batches[0])
    torch.cuda.synchronize()
    for k in want:
        torch.testing.assert_close(got[k].float(), want[k].float(), rtol=1e-4, atol=1e-4)
    # three micro-batches per call: one plan of three steps' calls
    tables = []
    for plans in (False, True):
        model, batches = _model("c4", dev)
        stacked = {k: torch.cat([b[k] for b in batches[:3]], dim=0) for k in batches[0]}
        runner = runtime.training_model(model, runtime.Options(device_iterations=3, use_plans=plans), runtime.SGD(lr=0.02),
                                        device=dev)
        for _ in range(2):
            runner(**stacked)
        torch.cuda.synchronize()
        tables.append(model.score_fn.entity_embedding.detach().float().clone())
        if plans:
            (calls,) = runner.plan_calls().values()
            assert len(calls) % 3 == 0 and calls[: len(calls) // 3] == calls[len(calls) // 3: 2 * len(calls) // 3]
    off = (tables[0] - tables[1]).abs()
    assert float((off > 3e-3).float().mean()) < 2e-4


def test_a_step_that_derives_index_tensors_with_torch_is_refused(dev):
    """'ht' corruption selects the head- and tail-corrupting halves of every block with torch indexing (`idx[sel]`):
    those index tensors are made from the inputs by work the plan does not hold - and a replay would read whatever
    the recording left in their recycled memory.  The recording runs under the profiler and refuses, before anything
    is replayed, as soon as the device has seen work that is not the library's."""
    from besskge import runtime

    model, batches = _model("ht", dev)
    runner = runtime.training_model(model, runtime.Options(use_plans=True), runtime.SGD(lr=0.02, momentum=0.9), device=dev)
    before = model.score_fn.entity_embedding.detach().clone()
    with pytest.raises(RuntimeError, match="not made of library calls only"):
        runner(**batches[0])
    assert torch.equal(model.score_fn.entity_embedding.detach(), before)


def test_a_step_with_work_outside_the_library_is_refused(dev):
    """Two shards in lock-step in one process exchange rows with torch operators (SingleProcessGroup's block
    transpose): not a step a plan can hold - the recording says so, and replays nothing."""
    from besskge import runtime
    from besskge.bess import EmbeddingMovingBessKGE
    from besskge.loss import LogSigmoidLoss
    from besskge.negative_sampler import RandomShardedNegativeSampler
    from besskge.scoring import TransE
    from besskge.sharding import Sharding

    sharding = Sharding.create(4000, 2, seed=0)
    fn = TransE(True, 1, sharding, 9, 32, device=dev)
    ns = RandomShardedNegativeSampler(16, sharding, 0, "t", local_sampling=False, flat_negative_format=True)
    model = EmbeddingMovingBessKGE(ns, fn, LogSigmoidLoss(margin=3.0, negative_adversarial_sampling=True))
    rng = np.random.default_rng(0)
    M = int(sharding.shard_counts.min())
    b = dict(head=rng.integers(M, size=(2, 2, 32)), relation=rng.integers(9, size=(2, 2, 32)),
             tail=rng.integers(M, size=(2, 2, 32)), negative=rng.integers(M, size=(2, 2, 1, 16)))
    b = {k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in b.items()}
    runner = runtime.training_model(model, runtime.Options(use_plans=True), runtime.SGD(lr=0.05), device=dev)
    before = model.score_fn.entity_embedding.detach().clone()
    with pytest.raises(RuntimeError, match="not made of library calls only"):
        runner(**b)
    assert torch.equal(model.score_fn.entity_embedding.detach(), before)  # the failed recording left the tables alone


_NATIVE_PLAN = r"""
import sys, torch
sys.path[:0] = [sys.argv[1], sys.argv[2], sys.argv[3]]
import numpy as np
from besskge import _native as nat, runtime
from besskge.collectives import NativeGroup
from test_plans import _model
dev = torch.device("cuda", 0)
tables = []
for plans in (False, True):
    group = NativeGroup(dev, unique_id=nat.comm_unique_id(), world=1, rank=0)
    model, batches = _model("c4", dev)
    runner = runtime.training_model(model, runtime.Options(use_plans=plans), runtime.SGD(lr=0.02), group=group, device=dev)
    for b in batches:
        runner(**b)
    torch.cuda.synchronize()
    tables.append(model.score_fn.entity_embedding.detach().float().clone())
    if plans:
        (calls,) = runner.plan_calls().values()
        assert "bess_allreduce_sum_f32" in calls, calls   # the collective of the step is one of the plan's calls
    group.close()
torch.testing.assert_close(tables[0], tables[1], rtol=2e-3, atol=3e-3)
print("ok", flush=True)
"""


def test_plan_with_the_librarys_collectives():
    """`NativeGroup` (one rank: what a one-GPU box can show): the step's RCCL calls are entries of the plan and are
    replayed with it - no RCCL inside a hipGraph."""
    repo = os.path.dirname(HERE)
    res = subprocess.run([sys.executable, "-c", _NATIVE_PLAN, os.path.join(repo, "bess-kge_amd"), repo, HERE],
                         capture_output=True, text=True, timeout=180, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
