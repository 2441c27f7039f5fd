#!/usr/bin/env python3
"""Kernel-by-kernel view of ONE training step.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_X -- python3 profiles/step_trace.py run c2
    python3 profiles/step_trace.py show gpurun_out/trace_X > profiles/r02/step_c2.txt

`run <workload>` does a few warm-up steps and then 4 steps separated by a marker kernel (a fill of an int64
tensor: `FillFunctor<long>`); `show` lists the dispatches of the last marked step in start order with their
duration and the idle gap in front of each (all streams), and the totals.  Workloads: c2score (the bench headline: gather + score + loss), c2 (the same as a
training step), c2sm (that step in its multi-GPU form, ScoreMovingBessKGE, on one shard), c2adam, c4s (S=512, K=32: the notebook's micro-batch), c4 (S=4096, K=256), c4g (c4s replayed
from a hipGraph), c4p (c4s replayed from a C-side step plan: `Options.use_plans`), c4n2 / c4n2g (two shards stepped in lock-step on this GPU - the n > 1 code path - eager / replayed from a hipGraph), c2em2 (C2's scorer in the EmbeddingMoving form on two shards in lock-step: per-triple negatives through the all-to-all).
"""
import csv
import glob
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]


def run(workload: str) -> None:
    import numpy as np
    import torch

    import bench
    from besskge import runtime
    from besskge.collectives import SingleProcessGroup

    dev = torch.device("cuda", 0)
    marker = torch.zeros(1, dtype=torch.int64, device=dev)
    if workload == "c2em2":
        # C2's scorer and negatives in the EmbeddingMoving form on TWO shards stepped in lock-step on this GPU: the
        # per-triple negatives arrive through the all-to-all (fused forward over the received rows, backward stores
        # d_neg = coefficient x query straight into the receive-buffer gradient)
        from besskge.bess import EmbeddingMovingBessKGE
        from besskge.loss import LogSigmoidLoss
        from besskge.negative_sampler import RandomShardedNegativeSampler
        from besskge.scoring import ComplEx
        from besskge.sharding import Sharding

        nsh, S_, K_ = 2, 2048, 128  # per shard: 2048 positives x (2 x 128) negatives
        sharding = Sharding.create(bench.N_ENTITY_C2 * nsh, nsh, seed=1234)
        fn = ComplEx(False, sharding, bench.N_REL, bench.D, device=dev, shards=list(range(nsh)))
        ns = RandomShardedNegativeSampler(K_, sharding, 1234, "t", local_sampling=False, flat_negative_format=False)
        model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn,
                                       loss_fn=LogSigmoidLoss(margin=12.0, negative_adversarial_sampling=True))
        for p_ in (fn.entity_embedding, fn.relation_embedding):
            p_.requires_grad_(False)
        rng = np.random.default_rng(0)
        M, pp = bench.N_ENTITY_C2, S_ // nsh
        batch = dict(head=rng.integers(M, size=(nsh, nsh, pp)), relation=rng.integers(bench.N_REL, size=(nsh, nsh, pp)),
                     tail=rng.integers(M, size=(nsh, nsh, pp)), negative=rng.integers(M, size=(nsh, nsh, S_, K_)))
        batch = {k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in batch.items()}
        runner = runtime.training_model(model, runtime.Options(device_iterations=1), runtime.SGD(lr=1e-3), device=dev)

        def step(i):
            runner(**batch)
    elif workload.startswith("c2"):
        # c2sm: the multi-GPU form of the step (ScoreMoving) on one shard
        model, sharding, k_pair = bench.build_c2(bench.N_ENTITY_C2, 1, 0, dev, SingleProcessGroup(1), workload == "c2sm")
        batches = bench.make_batches_c2(1, 0, sharding, k_pair, pool=4, dev=dev)
        opt = runtime.Adam(lr=1e-3, weight_decay=1e-2) if workload == "c2adam" else 1e-3

        def step(i):
            if workload == "c2score":
                with torch.no_grad():
                    model.forward_replicas([batches[i % 4]])
            else:
                model.train_step_replicas([batches[i % 4]], opt)
    else:
        from besskge.bess import EmbeddingMovingBessKGE
        from besskge.loss import SampledSoftmaxCrossEntropyLoss
        from besskge.negative_sampler import RandomShardedNegativeSampler
        from besskge.scoring import TransE
        from besskge.sharding import Sharding

        S_, K_ = (512, 32) if workload in ("c4s", "c4g", "c4p", "c4n2", "c4n2g") else (4096, 256)
        nsh = 2 if workload in ("c4n2", "c4n2g") else 1  # c4n2: two shards stepped in lock-step on this GPU (the n > 1 code path)
        sharding = Sharding.create(bench.C4_ROWS_PER_SHARD * nsh, nsh, seed=0)
        fn = TransE(True, 1, sharding, bench.C4_N_REL, bench.C4_D, device=dev, shards=list(range(nsh)), dtype=torch.float16)
        ns = RandomShardedNegativeSampler(K_, sharding, 0, "t", local_sampling=False, flat_negative_format=True)
        model = EmbeddingMovingBessKGE(negative_sampler=ns, score_fn=fn, augment_negative=True,
                                       loss_fn=SampledSoftmaxCrossEntropyLoss(n_entity=bench.C4_N_ENTITY))
        rng = np.random.default_rng(0)
        M = bench.C4_ROWS_PER_SHARD
        pp = S_ // nsh
        batch = dict(head=rng.integers(M, size=(nsh, nsh, pp)), relation=rng.integers(bench.C4_N_REL, size=(nsh, nsh, pp)),
                     tail=rng.integers(M, size=(nsh, nsh, pp)), negative=rng.integers(M, size=(nsh, nsh, 1, K_)))
        batch = {k: torch.from_numpy(v.astype(np.int32)).to(dev) for k, v in batch.items()}
        runner = runtime.training_model(model, runtime.Options(device_iterations=1, use_graphs=workload in ("c4g", "c4n2g"),
                                                        use_plans=workload == "c4p"),
                                        runtime.SGD(lr=1e-3), device=dev)

        if workload in ("c4g", "c4n2g", "c4p"):  # inputs where the recorded step reads them: no copies in front of the replay
            static = runner.static_inputs(**batch)
            for k_, v_ in batch.items():
                static[k_].copy_(v_)
            batch = static

        def step(i):
            runner(**batch)

    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    for i in range(4):
        marker.fill_(i)
        step(i)
        torch.cuda.synchronize()
    marker.fill_(99)
    torch.cuda.synchronize()


def show(path: str) -> None:
    f = max(glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "FillFunctor<long>" in r["Kernel_Name"]]
    assert len(marks) >= 2, "no step markers in the trace"
    step = rows[marks[-2] + 1: marks[-1]]
    t0 = int(step[0]["Start_Timestamp"])
    end_prev, busy = t0, 0
    print(f"# {len(step)} dispatches in the last marked step ({f})")
    print(f"{'start us':>9} {'dur us':>8} {'gap us':>7}  stream/queue  kernel")
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"]
        name = name[:110]
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {max(0, s - end_prev) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>4}  {name}")
        busy += e - s
        end_prev = max(end_prev, e)
    print(f"# span {(end_prev - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        show(sys.argv[2])
