#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/*.npz.

Runs the upstream reference's *own* code (imported in place from
/root/reference through `ref_shim`, nothing is copied) on CPU and stores
inputs + outputs as data.  Run in the build container only:

    python tests/golden/make_golden.py

The fixtures pin, against the reference itself:
  sharding.npz          Sharding.create                       (sharding.py:67-137)
  partition.npz         PartitionedTripleSet.*                (sharding.py:226-511)
  negative_sampler.npz  Random/TypeBased/TripleBased samplers (negative_sampler.py)
  batch_sampler.npz     Rigid/Random batch samplers           (batch_sampler.py)
  embedding.npz         initialize_*/refactor_embedding_sharding (embedding.py:107-290)
  scoring.npz           TransE/RotatE/DistMult/ComplEx score_* + grads (scoring.py)
  loss.npz              LogSigmoid/MarginRanking/SampledSoftmaxCE + grads (loss.py)
  bess.npz              {EmbeddingMoving,ScoreMoving}BessKGE.forward, n_shard in
                        {1,2,4}, h/t/ht, flat / per-triple, loss, augment, grads (bess.py)
  metric.npz            Evaluation ranks / metrics                       (metric.py:74-273)
  topk.npz              TopKQueryBessKGE.forward (+ Evaluation)          (bess.py:606-921)
  allscores.npz         AllScoresBESS.forward, every window step         (bess.py:924-1062)
  bess_half.npz         BessKGE.forward + autograd in the fp16 mode (`model.half()`,
                        notebooks/3_wikikg2_fp16.ipynb:300-392) on torch's CPU half kernels
"""

import copy
import os
import sys
import zlib
from typing import Any, Dict, List

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

# one thread: the last bits of CPU reductions (ConvE's conv / batch-norm gradients, whose true value is ~0 - pure
# cancellation) depend on how torch splits them over threads; with one thread the fixtures regenerate bit for bit
# on any machine
torch.set_num_threads(1)

ref_shim.install()

from besskge.batch_sampler import (  # noqa: E402
    RandomShardedBatchSampler,
    RigidShardedBatchSampler,
)
from besskge.bess import (  # noqa: E402
    AllScoresBESS,
    EmbeddingMovingBessKGE,
    ScoreMovingBessKGE,
    TopKQueryBessKGE,
)
from besskge.dataset import KGDataset  # noqa: E402
from besskge.embedding import (  # noqa: E402
    initialize_entity_embedding,
    refactor_embedding_sharding,
)
from besskge.loss import (  # noqa: E402
    LogSigmoidLoss,
    MarginRankingLoss,
    SampledSoftmaxCrossEntropyLoss,
)
from besskge.metric import Evaluation  # noqa: E402
from besskge.negative_sampler import (  # noqa: E402
    PlaceholderNegativeSampler,
    RandomShardedNegativeSampler,
    TripleBasedShardedNegativeSampler,
    TypeBasedShardedNegativeSampler,
)
from besskge.scoring import ComplEx, DistMult, RotatE, TransE  # noqa: E402
from besskge.sharding import PartitionedTripleSet, Sharding  # noqa: E402

OUT: Dict[str, Dict[str, np.ndarray]] = {}


def put(fix: str, key: str, val: Any) -> None:
    if isinstance(val, torch.Tensor):
        val = val.detach().cpu().numpy()
    OUT.setdefault(fix, {})[key] = np.asarray(val)


def sharding_arrays(fix: str, prefix: str, s: Sharding) -> None:
    put(fix, prefix + "entity_to_shard", s.entity_to_shard)
    put(fix, prefix + "entity_to_idx", s.entity_to_idx)
    put(fix, prefix + "shard_and_idx_to_entity", s.shard_and_idx_to_entity)
    put(fix, prefix + "shard_counts", s.shard_counts)
    if s.entity_type_counts is not None:
        put(fix, prefix + "entity_type_counts", s.entity_type_counts)
        put(fix, prefix + "entity_type_offsets", s.entity_type_offsets)


def make_dataset(
    n_entity: int,
    n_rel: int,
    triples: np.ndarray,
    type_offsets: Any = None,
    neg_heads: Any = None,
    neg_tails: Any = None,
    part: str = "train",
) -> KGDataset:
    return KGDataset(
        n_entity=n_entity,
        n_relation_type=n_rel,
        entity_dict=None,
        relation_dict=None,
        type_offsets=type_offsets,
        triples={part: triples},
        original_triple_ids={part: np.arange(triples.shape[0])},
        neg_heads=None if neg_heads is None else {part: neg_heads},
        neg_tails=None if neg_tails is None else {part: neg_tails},
    )


# --------------------------------------------------------------------------- #
def gen_sharding() -> None:
    fix = "sharding"
    cases = [
        (500, 4, 1234, None),
        (500, 4, 1234, np.array([0, 200, 260])),
        (1001, 3, 7, None),
        (64, 1, 0, None),
        (10, 8, 5, None),
        (997, 8, 42, np.array([0, 10, 500, 900])),
    ]
    put(fix, "n_cases", len(cases))
    for i, (ne, ns, seed, to) in enumerate(cases):
        s = Sharding.create(ne, ns, seed=seed, type_offsets=to)
        p = f"c{i}_"
        put(fix, p + "args", np.array([ne, ns, seed]))
        put(fix, p + "type_offsets", np.array([]) if to is None else to)
        sharding_arrays(fix, p, s)


def gen_partition() -> None:
    fix = "partition"
    rng = np.random.default_rng(11)
    n_entity, n_rel, n_triple, n_shard = 300, 7, 1500, 4
    type_offsets = {"a": 0, "b": 120, "c": 200}
    triples = np.stack(
        [
            rng.integers(n_entity, size=n_triple),
            rng.integers(n_rel, size=n_triple),
            rng.integers(n_entity, size=n_triple),
        ],
        axis=1,
    )
    neg_heads = rng.integers(n_entity, size=(n_triple, 9)).astype(np.int32)
    neg_tails = rng.integers(n_entity, size=(n_triple, 9)).astype(np.int32)
    put(fix, "args", np.array([n_entity, n_rel, n_triple, n_shard, 1234]))
    put(fix, "type_offsets", np.array(list(type_offsets.values())))
    put(fix, "triples", triples)
    put(fix, "neg_heads", neg_heads)
    put(fix, "neg_tails", neg_tails)
    sharding = Sharding.create(
        n_entity, n_shard, seed=1234, type_offsets=np.array(list(type_offsets.values()))
    )
    ds = make_dataset(
        n_entity, n_rel, triples, type_offsets, neg_heads, neg_tails, part="train"
    )
    for mode in ["h_shard", "t_shard", "ht_shardpair"]:
        for inv in [False, True]:
            pts = PartitionedTripleSet.create_from_dataset(
                ds, "train", sharding, partition_mode=mode, add_inverse_triples=inv
            )
            p = f"{mode}_{int(inv)}_"
            put(fix, p + "triples", pts.triples)
            put(fix, p + "triple_counts", pts.triple_counts)
            put(fix, p + "triple_offsets", pts.triple_offsets)
            put(fix, p + "triple_sort_idx", pts.triple_sort_idx)
            put(fix, p + "types", pts.types)
            put(fix, p + "neg_heads", pts.neg_heads)
            put(fix, p + "neg_tails", pts.neg_tails)
    # queries
    queries = np.stack(
        [rng.integers(n_entity, size=200), rng.integers(n_rel, size=200)], axis=1
    )
    gt = rng.integers(n_entity, size=200)
    qneg = rng.integers(n_entity, size=(200, 5))
    put(fix, "q_queries", queries)
    put(fix, "q_ground_truth", gt)
    put(fix, "q_negative", qneg)
    qcases = {
        "q_hr_plain": dict(query_mode="hr"),
        "q_rt_gt": dict(query_mode="rt", ground_truth=gt, queries=queries[:, ::-1]),
        "q_hr_neg": dict(query_mode="hr", negative=qneg, ground_truth=gt),
        "q_hr_type": dict(query_mode="hr", negative_type="b"),
    }
    for name, kw in qcases.items():
        kw = dict(kw)
        q = kw.pop("queries", queries)
        pts = PartitionedTripleSet.create_from_queries(ds, sharding, q, **kw)
        p = name + "_"
        put(fix, p + "triples", pts.triples)
        put(fix, p + "triple_counts", pts.triple_counts)
        put(fix, p + "triple_offsets", pts.triple_offsets)
        put(fix, p + "triple_sort_idx", pts.triple_sort_idx)
        put(fix, p + "dummy", np.array(str(pts.dummy)))
        put(fix, p + "partition_mode", np.array(pts.partition_mode))
        if pts.types is not None:
            put(fix, p + "types", pts.types)
        if pts.neg_heads is not None:
            put(fix, p + "neg_heads", pts.neg_heads)
        if pts.neg_tails is not None:
            put(fix, p + "neg_tails", pts.neg_tails)


def gen_negative_sampler() -> None:
    fix = "negative_sampler"
    seed, n_entity, n_shard, n_triple = 1234, 500, 4, 600
    bps, ppp, n_negative = 2, 6, 5
    rng = np.random.default_rng(3)
    type_offsets = np.array([0, 200, 260])
    sharding = Sharding.create(n_entity, n_shard, seed=seed, type_offsets=type_offsets)
    put(fix, "args", np.array([seed, n_entity, n_shard, n_triple, bps, ppp, n_negative]))
    put(fix, "type_offsets", type_offsets)
    sizes = {
        "shard": (bps, n_shard, ppp),
        "shardpair": (bps, n_shard, n_shard, ppp),
    }
    sample_idx = {k: rng.integers(n_triple, size=v) for k, v in sizes.items()}
    for k, v in sample_idx.items():
        put(fix, f"sample_idx_{k}", v)
    triple_types = rng.integers(3, size=(n_triple, 2)).astype(np.int32)
    put(fix, "triple_types", triple_types)
    for pm in ["shard", "shardpair"]:
        for flat in [True, False]:
            for scheme in ["h", "ht"]:
                ns = RandomShardedNegativeSampler(
                    n_negative=n_negative,
                    sharding=sharding,
                    seed=seed,
                    corruption_scheme=scheme,
                    local_sampling=False,
                    flat_negative_format=flat,
                )
                a = ns(sample_idx[pm])["negative_entities"]
                b = ns(sample_idx[pm])["negative_entities"]  # second draw, same rng
                put(fix, f"random_{pm}_{int(flat)}_{scheme}_0", a)
                put(fix, f"random_{pm}_{int(flat)}_{scheme}_1", b)
        for local in [True, False]:
            for scheme in ["h", "t", "ht"]:
                ns = TypeBasedShardedNegativeSampler(
                    triple_types=triple_types,
                    n_negative=n_negative,
                    sharding=sharding,
                    corruption_scheme=scheme,
                    local_sampling=local,
                    seed=seed,
                )
                put(
                    fix,
                    f"type_{pm}_{int(local)}_{scheme}",
                    ns(sample_idx[pm])["negative_entities"],
                )
    n_neg_tb = 23
    for flat in [True, False]:
        nh = rng.integers(n_entity, size=(1 if flat else n_triple, n_neg_tb)).astype(
            np.int32
        )
        nt = rng.integers(n_entity, size=(1 if flat else n_triple, n_neg_tb)).astype(
            np.int32
        )
        put(fix, f"tb_neg_heads_{int(flat)}", nh)
        put(fix, f"tb_neg_tails_{int(flat)}", nt)
        for pm in ["shard", "shardpair"]:
            for scheme in ["h", "t", "ht"]:
                for mog in [False, True]:
                    ns = TripleBasedShardedNegativeSampler(
                        nh,
                        nt,
                        sharding,
                        corruption_scheme=scheme,
                        seed=seed,
                        return_sort_idx=True,
                        mask_on_gather=mog,
                    )
                    out = ns(sample_idx[pm])
                    p = f"tb_{pm}_{int(flat)}_{scheme}_{int(mog)}_"
                    put(fix, p + "negative_entities", out["negative_entities"])
                    put(fix, p + "negative_mask", out["negative_mask"])
                    put(fix, p + "negative_sort_idx", out["negative_sort_idx"])
                    put(fix, p + "padded_shard_length", ns.padded_shard_length)


def gen_batch_sampler() -> None:
    fix = "batch_sampler"
    seed, n_entity, n_rel, n_shard, n_triple = 1234, 500, 10, 4, 2000
    bps, shard_bs, n_negative = 3, 24, 4
    rng = np.random.default_rng(5)
    triples = np.stack(
        [
            rng.integers(n_entity, size=n_triple),
            rng.integers(n_rel, size=n_triple),
            rng.integers(n_entity, size=n_triple),
        ],
        axis=1,
    )
    put(fix, "args", np.array([seed, n_entity, n_rel, n_shard, n_triple, bps, shard_bs, n_negative]))
    put(fix, "triples", triples)
    ds = make_dataset(n_entity, n_rel, triples)
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    for mode in ["h_shard", "t_shard", "ht_shardpair"]:
        pts = PartitionedTripleSet.create_from_dataset(
            ds, "train", sharding, partition_mode=mode
        )
        for kind in ["rigid", "random"]:
            for dup in [False, True]:
                for scheme, flat in [("h", False), ("ht", True)]:
                    for hrt in [False, True]:
                        ns = RandomShardedNegativeSampler(
                            n_negative=n_negative,
                            sharding=sharding,
                            seed=seed,
                            corruption_scheme=scheme,
                            local_sampling=False,
                            flat_negative_format=flat,
                        )
                        cls = (
                            RigidShardedBatchSampler
                            if kind == "rigid"
                            else RandomShardedBatchSampler
                        )
                        bs = cls(
                            partitioned_triple_set=pts,
                            negative_sampler=ns,
                            shard_bs=shard_bs,
                            batches_per_step=bps,
                            seed=seed,
                            hrt_freq_weighting=hrt,
                            weight_smoothing=0.5 if hrt else 0.0,
                            duplicate_batch=dup,
                            return_triple_idx=True,
                        )
                        p = f"{mode}_{kind}_{int(dup)}_{scheme}_{int(hrt)}_"
                        put(fix, p + "len", len(bs))
                        put(fix, p + "ppp", bs.positive_per_partition)
                        it = iter(bs.get_dataloader_sampler(shuffle=False))
                        idxs = [next(it), next(it)]
                        # also the last (padded) rigid batch
                        if kind == "rigid":
                            last = None
                            for last in bs.get_dataloader_sampler(shuffle=False):
                                pass
                            idxs.append(last)
                        for j, idx in enumerate(idxs):
                            put(fix, p + f"b{j}_idx", np.array(idx))
                            b = bs[idx]
                            for k, v in b.items():
                                put(fix, p + f"b{j}_{k}", v)


def gen_embedding() -> None:
    fix = "embedding"
    torch.manual_seed(0)
    s_old = Sharding.create(101, 4, seed=3)
    s_new = Sharding.create(101, 3, seed=9)
    unsharded = torch.randn(101, 6)
    put(fix, "unsharded", unsharded)
    sharding_arrays(fix, "old_", s_old)
    sharding_arrays(fix, "new_", s_new)
    sharded = initialize_entity_embedding(s_old, unsharded)
    put(fix, "sharded_old", sharded)
    put(fix, "sharded_new", refactor_embedding_sharding(sharded, s_old, s_new))


# --------------------------------------------------------------------------- #
def scorer_factory(name: str, p: int, sharing: bool, sharding: Sharding, n_rel: int, d: int,
                   ent: torch.Tensor, rel: torch.Tensor) -> Any:
    if name == "TransE":
        return TransE(sharing, p, sharding, n_rel, d, ent, rel)
    if name == "RotatE":
        return RotatE(sharing, p, sharding, n_rel, d, ent, rel)
    if name == "DistMult":
        return DistMult(sharing, sharding, n_rel, d, ent, rel)
    if name == "ComplEx":
        return ComplEx(sharing, sharding, n_rel, d, ent, rel)
    if name in BOXE_VARIANTS:
        from besskge.scoring import BoxE

        tanh, per_dim = BOXE_VARIANTS[name]
        return BoxE(sharing, p, sharding, n_rel, d, ent, rel, apply_tanh=tanh, dist_func_per_dim=per_dim)
    if name == "ConvE":
        from besskge.scoring import ConvE

        fn = ConvE(sharing, sharding, n_rel, d, d // 4, 4, ent, rel, inverse_relations=False, input_dropout=0.0,
                   feature_map_dropout=0.0, hidden_dropout=0.0)
        # non-trivial batch-norm parameters and running statistics
        for m in list(fn.conv_layers) + list(fn.fc_layers):
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0.0, 0.2)
                m.running_mean.normal_(0.0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
        return fn
    if name in AFFINE_VARIANTS:
        from besskge.scoring import InterHT, PairRE, TranS, TripleRE

        cfg = AFFINE_VARIANTS[name]
        if cfg["base"] == "PairRE":
            return PairRE(sharing, p, sharding, n_rel, d, ent, rel, normalize_entities=cfg["normalize"])
        if cfg["base"] == "TripleRE":
            return TripleRE(sharing, p, sharding, n_rel, d, ent, rel, normalize_entities=cfg["normalize"], u=cfg["u"])
        if cfg["base"] == "InterHT":
            return InterHT(sharing, p, sharding, n_rel, d, ent, rel, normalize_entities=cfg["normalize"],
                           offset=cfg["offset"])
        return TranS(sharing, p, sharding, n_rel, d, ent, rel, normalize_entities=cfg["normalize"],
                     offset=cfg["offset"])
    raise ValueError(name)


# constructor options of the PairRE / TripleRE / InterHT / TranS fixtures (same table in oracle/kge.py)
AFFINE_VARIANTS = {
    "PairRE": dict(base="PairRE", normalize=True),
    "TripleRE": dict(base="TripleRE", normalize=True, u=0.0),
    "TripleREv2": dict(base="TripleRE", normalize=False, u=0.5),
    "InterHT": dict(base="InterHT", normalize=True, offset=1.0),
    "TranS": dict(base="TranS", normalize=True, offset=1.0),
    "TranSnn": dict(base="TranS", normalize=False, offset=0.5),
}
BOXE_VARIANTS = {"BoxE": (True, True), "BoxEnt": (False, False), "BoxEall": (True, False), "BoxEpd": (False, True)}
BOXE_SCORERS = [("BoxE", 1), ("BoxE", 2), ("BoxEnt", 1), ("BoxEall", 2), ("BoxEpd", 2)]
AFFINE_SCORERS = [("PairRE", 1), ("PairRE", 2), ("TripleRE", 1), ("TripleREv2", 2), ("InterHT", 1), ("InterHT", 2),
                  ("TranS", 1), ("TranSnn", 2)]


SCORERS = [("TransE", 1), ("TransE", 2), ("RotatE", 1), ("RotatE", 2), ("DistMult", 0), ("ComplEx", 0)]


def net_state(fn: Any) -> Dict[str, torch.Tensor]:
    """Parameters and buffers of ConvE's query network, by state-dict name."""
    return {k: v.detach().clone() for k, v in fn.state_dict().items()
            if k.startswith("conv_layers") or k.startswith("fc_layers")}


def gen_scoring_conve() -> None:
    """ConvE (scoring.py:949-1146): triple and tail scores, eval and train mode (no dropout),
    gradients wrt embeddings, relation table and the network parameters."""
    fix = "scoring_conve"
    torch.manual_seed(3)
    S, N, d, n_rel, n_ent = 10, 7, 12, 5, 40
    sharding = Sharding.create(n_ent, 1, seed=0)
    put(fix, "args", np.array([S, N, d, n_rel, n_ent]))
    W = d + 1
    ent = torch.randn(1, n_ent, W)
    rel = torch.randn(n_rel, d)
    h, t = torch.randn(S, W), torch.randn(S, W)
    rid = torch.randint(n_rel, (S,))
    neg1, negS = torch.randn(1, N, W), torch.randn(S, N, W)
    g_pos = torch.randn(S)
    for k, v in dict(rel=rel, h=h, t=t, rid=rid, neg1=neg1, negS=negS, g_pos=g_pos).items():
        put(fix, k, v)
    base = scorer_factory("ConvE", 0, True, sharding, n_rel, d, ent, rel.clone())
    for k, v in net_state(base).items():
        put(fix, "net_" + k, v)
    for mode in ["eval", "train"]:
        for sharing in [True, False]:
            for B, neg in [(1, neg1), (S, negS)]:
                fn = copy.deepcopy(base)
                fn.negative_sample_sharing = sharing
                fn.train(mode == "train")
                hh = h.clone().requires_grad_(True)
                tt = t.clone().requires_grad_(True)
                nn_ = neg.clone().requires_grad_(True)
                c = f"{mode}_s{int(sharing)}_B{B}_"
                pos = fn.score_triple(hh, rid, tt)
                (pos * g_pos).sum().backward()
                put(fix, c + "pos", pos)
                put(fix, c + "pos_dh", hh.grad)
                put(fix, c + "pos_dt", tt.grad)
                put(fix, c + "pos_drel", fn.relation_embedding.grad)
                for name, prm in fn.named_parameters():
                    if name.startswith("conv_layers") or name.startswith("fc_layers"):
                        put(fix, c + "pos_dnet_" + name, prm.grad)
                fn = copy.deepcopy(base)
                fn.negative_sample_sharing = sharing
                fn.train(mode == "train")
                hh = h.clone().requires_grad_(True)
                st = fn.score_tails(hh, rid, nn_)
                g_neg = torch.randn_like(st)
                (st * g_neg).sum().backward()
                put(fix, c + "tails", st)
                put(fix, c + "tails_g", g_neg)
                put(fix, c + "tails_dneg", nn_.grad)
                put(fix, c + "tails_dh", hh.grad)
                put(fix, c + "tails_drel", fn.relation_embedding.grad)
                for name, prm in fn.named_parameters():
                    if name.startswith("conv_layers") or name.startswith("fc_layers"):
                        put(fix, c + "tails_dnet_" + name, prm.grad)


def gen_bess_conve() -> None:
    EM, SM = EmbeddingMovingBessKGE, ScoreMovingBessKGE
    cases = [
        ("tr_EM_ConvE0_t_flat_n1", EM, "ConvE", 0, 1, "t", "random_flat", "logsigmoid", False, True),
        ("tr_EM_ConvE0_t_pt_n1", EM, "ConvE", 0, 1, "t", "random_pt", "ssce", False, False),
        ("tr_EM_ConvE0_t_flat_n2", EM, "ConvE", 0, 2, "t", "random_flat", "margin", False, True),
        ("tr_SM_ConvE0_t_pt_n2", SM, "ConvE", 0, 2, "t", "random_pt", "logsigmoid", False, False),
    ]
    names = []
    for name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing in cases:
        run_bess_case(name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing, fix="bess_conve")
        names.append(name)
    put("bess_conve", "cases", np.array(names))


def gen_scoring_boxe() -> None:
    _gen_scoring("scoring_boxe", BOXE_SCORERS, 4)


def gen_bess_boxe() -> None:
    EM, SM = EmbeddingMovingBessKGE, ScoreMovingBessKGE
    cases = []
    for scorer, p in BOXE_SCORERS:
        sn = f"{scorer}{p}"
        cases.append((f"tr_EM_{sn}_t_flat_n1", EM, scorer, p, 1, "t", "random_flat", "logsigmoid", False, True))
        cases.append((f"tr_EM_{sn}_h_pt_n1", EM, scorer, p, 1, "h", "random_pt", "logsigmoid", False, False))
        cases.append((f"tr_EM_{sn}_ht_pt_n2", EM, scorer, p, 2, "ht", "random_pt", "ssce", False, False))
        cases.append((f"tr_SM_{sn}_ht_flat_n2", SM, scorer, p, 2, "ht", "random_flat", "margin", False, True))
    names = []
    for name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing in cases:
        run_bess_case(name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing, fix="bess_boxe")
        names.append(name)
    put("bess_boxe", "cases", np.array(names))


def gen_dataset() -> None:
    """KGDataset.from_dataframe / from_triples (dataset.py:83-239): label encoding, type blocks, split."""
    import pandas as pd

    fix = "dataset"
    rng = np.random.default_rng(5)
    ents = np.array([f"ent{i}" for i in range(120)])
    rels = np.array([f"rel{i}" for i in range(6)])
    h, r, t = rng.choice(ents, 700), rng.choice(rels, 700), rng.choice(ents, 700)
    kinds = rng.choice(np.array(["drug", "gene", "disease", "function"]), len(ents))
    for k, v in dict(h=h, r=r, t=t, ents=ents, kinds=kinds).items():
        put(fix, k, v)
    df = pd.DataFrame(dict(head=h, relation=r, tail=t))
    types = dict(zip(ents.tolist(), kinds.tolist()))
    cases = {
        "plain": KGDataset.from_dataframe(df, "head", "relation", "tail"),
        "typed": KGDataset.from_dataframe(df, "head", "relation", "tail", entity_types=types, split=(0.6, 0.2, 0.2), seed=7),
        "parts": KGDataset.from_dataframe({"train": df.iloc[:500], "test": df.iloc[500:]}, "head", "relation", "tail",
                                          entity_types=types),
    }
    for name, ds in cases.items():
        put(fix, f"{name}_n", np.array([ds.n_entity, ds.n_relation_type]))
        put(fix, f"{name}_entity_dict", np.array(ds.entity_dict))
        put(fix, f"{name}_relation_dict", np.array(ds.relation_dict))
        if ds.type_offsets is not None:
            put(fix, f"{name}_type_names", np.array(list(ds.type_offsets.keys())))
            put(fix, f"{name}_type_firsts", np.array(list(ds.type_offsets.values())))
        for part, tr in ds.triples.items():
            put(fix, f"{name}_triples_{part}", tr)
            put(fix, f"{name}_ids_{part}", ds.original_triple_ids[part])


def widths(name: str, d: int) -> Any:
    if name in BOXE_VARIANTS:
        return 2 * d, 4 * d + 2
    if name == "ConvE":
        return d + 1, d
    if name in AFFINE_VARIANTS:
        base = AFFINE_VARIANTS[name]["base"]
        return (2 * d if base in ("InterHT", "TranS") else d), {"PairRE": 2, "TripleRE": 3, "InterHT": 1, "TranS": 3}[base] * d
    W = 2 * d if name in ("RotatE", "ComplEx") else d
    Wr = 2 * d if name == "ComplEx" else d
    return W, Wr


def gen_scoring() -> None:
    _gen_scoring("scoring", SCORERS, 1)


def gen_scoring_affine() -> None:
    _gen_scoring("scoring_affine", AFFINE_SCORERS, 2)


#: p-norms beyond 1 and 2 (the reference takes any p through torch.norm: scoring.py:174)
LP_SCORERS = [("TransE", 3), ("RotatE", 3), ("TransE", 4),
              # round 4: the affine family and BoxE take any p too (scoring.py:540-593, 1250-1340)
              ("PairRE", 3), ("TripleREv2", 3), ("InterHT", 3), ("TranSnn", 4), ("BoxE", 3), ("BoxEnt", 3)]


def gen_scoring_lp() -> None:
    _gen_scoring("scoring_lp", LP_SCORERS, 7)


def gen_bess_lp() -> None:
    """BessKGE.forward + autograd with scoring_norm 3 / 4: shared and per-triple negatives, both schemes."""
    EM, SM = EmbeddingMovingBessKGE, ScoreMovingBessKGE
    cases = [
        ("tr_EM_TransE3_t_flat_n1", EM, "TransE", 3, 1, "t", "random_flat", "logsigmoid", False, True),
        ("tr_EM_TransE3_h_pt_n1", EM, "TransE", 3, 1, "h", "random_pt", "logsigmoid", False, False),
        ("tr_EM_RotatE3_ht_pt_n2", EM, "RotatE", 3, 2, "ht", "random_pt", "ssce", False, False),
        ("tr_EM_RotatE3_aug_t_flat_n2", EM, "RotatE", 3, 2, "t", "random_flat", "ssce", True, True),
        ("tr_SM_TransE4_t_pt_n2", SM, "TransE", 4, 2, "t", "random_pt", "logsigmoid", False, False),
        ("tr_SM_TransE3_ht_flat_n2", SM, "TransE", 3, 2, "ht", "random_flat", "margin", False, True),
        # round 4: p = 3 / 4 for PairRE / TripleRE / InterHT / TranS / BoxE
        ("tr_EM_PairRE3_t_flat_n1", EM, "PairRE", 3, 1, "t", "random_flat", "logsigmoid", False, True),
        ("tr_EM_InterHT3_h_pt_n1", EM, "InterHT", 3, 1, "h", "random_pt", "logsigmoid", False, False),
        ("tr_EM_TranSnn4_ht_pt_n2", EM, "TranSnn", 4, 2, "ht", "random_pt", "ssce", False, False),
        ("tr_SM_TripleREv23_ht_flat_n2", SM, "TripleREv2", 3, 2, "ht", "random_flat", "margin", False, True),
        ("tr_EM_BoxE3_t_flat_n1", EM, "BoxE", 3, 1, "t", "random_flat", "logsigmoid", False, True),
        ("tr_EM_BoxEnt3_ht_pt_n2", EM, "BoxEnt", 3, 2, "ht", "random_pt", "ssce", False, False),
        ("tr_SM_BoxE3_t_pt_n2", SM, "BoxE", 3, 2, "t", "random_pt", "logsigmoid", False, False),
    ]
    names = []
    for name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing in cases:
        run_bess_case(name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing, fix="bess_lp")
        names.append(name)
    put("bess_lp", "cases", np.array(names))


def _gen_scoring(fix: str, scorers: Any, seed: int) -> None:
    torch.manual_seed(seed)
    S, N, d, n_rel, n_ent = 10, 7, 12, 5, 40
    sharding = Sharding.create(n_ent, 1, seed=0)
    put(fix, "args", np.array([S, N, d, n_rel, n_ent]))
    for name, p in scorers:
        W, Wr = widths(name, d)
        ent = torch.randn(1, n_ent, W)
        rel = torch.randn(n_rel, Wr)
        h = torch.randn(S, W)
        t = torch.randn(S, W)
        rid = torch.randint(n_rel, (S,))
        neg1 = torch.randn(1, N, W)
        negS = torch.randn(S, N, W)
        key = f"{name}_p{p}_"
        for k, v in dict(rel=rel, h=h, t=t, rid=rid, neg1=neg1, negS=negS).items():
            put(fix, key + k, v)
        g_pos = torch.randn(S)
        put(fix, key + "g_pos", g_pos)
        for sharing in [True, False]:
            for B, neg in [(1, neg1), (S, negS)]:
                if (not sharing) and B == 1:
                    # broadcast branch (scoring.py:199/254) is still legal
                    pass
                fn = scorer_factory(name, p, sharing, sharding, n_rel, d, ent, rel.clone())
                hh = h.clone().requires_grad_(True)
                tt = t.clone().requires_grad_(True)
                nn_ = neg.clone().requires_grad_(True)
                pos = fn.score_triple(hh, rid, tt)
                sh = fn.score_heads(nn_, rid, tt)
                g_neg = torch.randn_like(sh)
                (pos * g_pos).sum().backward(retain_graph=True)
                c = key + f"s{int(sharing)}_B{B}_"
                put(fix, c + "pos", pos)
                put(fix, c + "pos_dh", hh.grad)
                put(fix, c + "pos_dt", tt.grad)
                put(fix, c + "pos_drel", fn.relation_embedding.grad)
                hh.grad = None
                tt.grad = None
                fn.relation_embedding.grad = None
                (sh * g_neg).sum().backward()
                put(fix, c + "heads", sh)
                put(fix, c + "heads_g", g_neg)
                put(fix, c + "heads_dneg", nn_.grad)
                put(fix, c + "heads_dt", tt.grad)
                put(fix, c + "heads_drel", fn.relation_embedding.grad)
                nn_.grad = None
                tt.grad = None
                fn.relation_embedding.grad = None
                st = fn.score_tails(hh, rid, nn_)
                g_neg2 = torch.randn_like(st)
                (st * g_neg2).sum().backward()
                put(fix, c + "tails", st)
                put(fix, c + "tails_g", g_neg2)
                put(fix, c + "tails_dneg", nn_.grad)
                put(fix, c + "tails_dh", hh.grad)
                put(fix, c + "tails_drel", fn.relation_embedding.grad)


def gen_loss() -> None:
    fix = "loss"
    torch.manual_seed(2)
    S, N = 9, 13
    pos = torch.randn(S) * 3
    neg = torch.randn(S, N) * 3
    neg[0, 3] = -50000.0  # a masked negative
    w = torch.rand(S) + 0.5
    put(fix, "pos", pos)
    put(fix, "neg", neg)
    put(fix, "w", w)
    losses = {
        "logsigmoid_adv": LogSigmoidLoss(margin=6.0, negative_adversarial_sampling=True, negative_adversarial_scale=0.7, loss_scale=2.0),
        "logsigmoid_uni": LogSigmoidLoss(margin=1.5, negative_adversarial_sampling=False),
        "margin_adv": MarginRankingLoss(margin=2.0, negative_adversarial_sampling=True, negative_adversarial_scale=1.3),
        "margin_uni": MarginRankingLoss(margin=0.5, negative_adversarial_sampling=False, loss_scale=4.0),
        "ssce": SampledSoftmaxCrossEntropyLoss(n_entity=1000, loss_scale=3.0),
    }
    for name, fn in losses.items():
        for wname, ww in [("w", w), ("one", torch.tensor([1.0]))]:
            pp = pos.clone().requires_grad_(True)
            nn_ = neg.clone().requires_grad_(True)
            # .float() on fp32 is a view in the reference; SSCE shifts in place -> pass a copy
            loss = fn(pp, nn_ + 0.0, ww)
            loss.backward()
            put(fix, f"{name}_{wname}_loss", loss)
            put(fix, f"{name}_{wname}_dpos", pp.grad)
            put(fix, f"{name}_{wname}_dneg", nn_.grad)


# --------------------------------------------------------------------------- #
def run_bess_case(
    case: str,
    model_cls: Any,
    scorer: str,
    p: int,
    n_shard: int,
    scheme: str,
    neg_kind: str,  # "random_flat" | "random_pt" | "tb_flat" | "tb_pt"
    loss_name: str,
    augment: bool,
    sharing: bool,
    batch_kind: str = "rigid",
    with_grads: bool = True,
    fix: str = "bess",
    half: bool = False,
    d: int = 8,
) -> None:
    seed = 1234
    n_entity, n_rel, n_triple = 120, 6, 400
    bps = 2
    shard_bs = 8 * n_shard if n_shard > 1 else 16
    n_negative = 5
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    torch.manual_seed(zlib.crc32(case.encode()) % (2**31))
    W, Wr = widths(scorer, d)
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    ent = torch.randn(n_shard, sharding.max_entity_per_shard, W)
    rel = torch.randn(n_rel, Wr)
    if half:
        # the reference's fp16 mode is `model.half()` (notebooks/3_wikikg2_fp16.ipynb:300-392): the tables and
        # every tensor of the score computation are fp16, the loss is computed in fp32 (bess.py:254-260).
        # The fixture stores the fp16 tables; the replicas below run the reference's forward / autograd on
        # them with torch's CPU half kernels (element-wise results rounded to fp16, reductions accumulated in
        # fp32 and rounded once)
        ent, rel = ent.half(), rel.half()
    triples = np.stack(
        [
            rng.integers(n_entity, size=n_triple),
            rng.integers(n_rel, size=n_triple),
            rng.integers(n_entity, size=n_triple),
        ],
        axis=1,
    )
    tb = neg_kind.startswith("tb")
    local = neg_kind.endswith("_local")  # negatives sampled on (and scored against) the processing shard
    flat = neg_kind.replace("_local", "").endswith("flat")
    nh = nt = None
    if tb:
        n_tb = 11
        nh = rng.integers(n_entity, size=(1 if flat else n_triple, n_tb)).astype(np.int32)
        nt = rng.integers(n_entity, size=(1 if flat else n_triple, n_tb)).astype(np.int32)
    ds = make_dataset(n_entity, n_rel, triples, None, nh, nt)
    pts = PartitionedTripleSet.create_from_dataset(ds, "train", sharding, partition_mode="ht_shardpair")
    if tb:
        ns = TripleBasedShardedNegativeSampler(
            pts.neg_heads, pts.neg_tails, sharding, corruption_scheme=scheme, seed=seed,
            return_sort_idx=True, mask_on_gather=False,
        )
    else:
        ns = RandomShardedNegativeSampler(
            n_negative=n_negative, sharding=sharding, seed=seed, corruption_scheme=scheme,
            local_sampling=local, flat_negative_format=flat,
        )
    dup = scheme == "ht" and tb
    bcls = RigidShardedBatchSampler if batch_kind == "rigid" else RandomShardedBatchSampler
    bs = bcls(
        partitioned_triple_set=pts, negative_sampler=ns, shard_bs=shard_bs, batches_per_step=bps,
        seed=seed, hrt_freq_weighting=False, duplicate_batch=dup, return_triple_idx=True,
    )
    score_fn = scorer_factory(scorer, p, sharing, sharding, n_rel, d, ent.float(), rel.float())
    loss_fn = None
    if loss_name == "logsigmoid":
        loss_fn = LogSigmoidLoss(margin=3.0, negative_adversarial_sampling=True, negative_adversarial_scale=0.5)
    elif loss_name == "margin":
        loss_fn = MarginRankingLoss(margin=1.0, negative_adversarial_sampling=False)
    elif loss_name == "ssce":
        loss_fn = SampledSoftmaxCrossEntropyLoss(n_entity=n_entity)
    model = model_cls(
        negative_sampler=ns, score_fn=score_fn, loss_fn=loss_fn, return_scores=True,
        augment_negative=augment,
    )
    batch = bs[next(iter(bs.get_dataloader_sampler(shuffle=False)))]
    p_ = case + "_"
    meta = dict(
        n_entity=n_entity, n_rel=n_rel, n_triple=n_triple, d=d, bps=bps, shard_bs=shard_bs,
        n_negative=n_negative, n_shard=n_shard, norm=p, augment=int(augment), sharing=int(sharing),
        dup=int(dup), flat=int(flat), tb=int(tb),
    )
    if half:
        meta["half"] = 1
    put(fix, p_ + "meta_keys", np.array(list(meta.keys())))
    put(fix, p_ + "meta_vals", np.array(list(meta.values())))
    put(fix, p_ + "strs", np.array([model_cls.__name__, scorer, scheme, neg_kind, loss_name, batch_kind]))
    put(fix, p_ + "entity_table", ent)
    put(fix, p_ + "relation_table", rel)
    if scorer == "ConvE":  # the query network, in the (default) train mode: batch statistics, no dropout
        for k, v in net_state(score_fn).items():
            put(fix, p_ + "net_" + k, v)
    for k, v in batch.items():
        put(fix, p_ + "batch_" + k, v)

    fwd_keys = ["head", "relation", "tail", "negative", "negative_mask"]

    # per-replica copies of the module: own [M, W] shard, own relation table
    reps = []
    for r in range(n_shard):
        m = copy.copy(model)
        m._modules = dict(model._modules)
        sf = copy.copy(score_fn)
        sf._parameters = dict(score_fn._parameters)
        sf.entity_embedding = torch.nn.Parameter(ent[r].clone())
        sf.relation_embedding = torch.nn.Parameter(rel.clone())
        m.score_fn = sf
        m._parameters = dict(model._parameters)
        m.entity_embedding = sf.entity_embedding
        reps.append(m)

    results: List[List[Dict[str, torch.Tensor]]] = [[None] * n_shard for _ in range(bps)]  # type: ignore
    for it in range(bps):

        def fn(r: int) -> Dict[str, torch.Tensor]:
            kw = {k: batch[k][it, r].unsqueeze(0) for k in fwd_keys if k in batch}
            out = reps[r](triple_weight=torch.tensor([1.0]), **kw)
            if loss_fn is not None and with_grads and it == 0:
                out["loss"].backward()
            return {k: v.detach().clone() for k, v in out.items() if isinstance(v, torch.Tensor)}

        res = ref_shim.run_replicas(n_shard, fn)
        for r in range(n_shard):
            results[it][r] = res[r]
    for k in results[0][0].keys():
        put(fix, p_ + "out_" + k, torch.stack([torch.stack([results[it][r][k] for r in range(n_shard)]) for it in range(bps)]))
    if loss_fn is not None and with_grads:
        put(fix, p_ + "grad_entity", torch.stack([reps[r].score_fn.entity_embedding.grad for r in range(n_shard)]))
        put(fix, p_ + "grad_relation", torch.stack([reps[r].score_fn.relation_embedding.grad for r in range(n_shard)]))
        if scorer == "ConvE":  # the network is shared by the replica copies: its .grad is the sum over replicas
            for name, prm in score_fn.named_parameters():
                if name.startswith("conv_layers") or name.startswith("fc_layers"):
                    put(fix, p_ + "gradnet_" + name, prm.grad)


def gen_bess() -> None:
    EM, SM = EmbeddingMovingBessKGE, ScoreMovingBessKGE
    cases = []
    # inference parity (mirrors tests/test_bess.py:54-275): TB negatives, both schemes
    for mname, mcls in [("EM", EM), ("SM", SM)]:
        for scheme in ["h", "t", "ht"]:
            for flat in [True, False]:
                nk = "tb_flat" if flat else "tb_pt"
                cases.append((f"inf_{mname}_{scheme}_{int(flat)}_n4", mcls, "TransE", 1, 4, scheme, nk, "none", False, flat))
    # every scorer, n_shard 1 and 2, random negatives, training (loss + grads)
    for scorer, p in SCORERS:
        sn = f"{scorer}{p}"
        cases.append((f"tr_EM_{sn}_t_flat_n1", EM, scorer, p, 1, "t", "random_flat", "logsigmoid", False, True))
        cases.append((f"tr_EM_{sn}_h_pt_n1", EM, scorer, p, 1, "h", "random_pt", "logsigmoid", False, False))
        cases.append((f"tr_EM_{sn}_ht_flat_n2", EM, scorer, p, 2, "ht", "random_flat", "margin", False, True))
        cases.append((f"tr_EM_{sn}_ht_pt_n2", EM, scorer, p, 2, "ht", "random_pt", "ssce", False, False))
        cases.append((f"tr_SM_{sn}_t_pt_n2", SM, scorer, p, 2, "t", "random_pt", "logsigmoid", False, False))
    # augment_negative + sharing variants (wikikg2 notebook setup: flat, t, augment, ssce)
    cases.append(("tr_EM_aug_t_flat_n4", EM, "TransE", 1, 4, "t", "random_flat", "ssce", True, True))
    cases.append(("tr_EM_aug_h_flat_n2", EM, "DistMult", 0, 2, "h", "random_flat", "logsigmoid", True, True))
    cases.append(("tr_EM_aug_ht_flat_n2", EM, "ComplEx", 0, 2, "ht", "random_flat", "ssce", True, True))
    cases.append(("tr_EM_aug_t_ptshare_n2", EM, "TransE", 2, 2, "t", "random_pt", "logsigmoid", True, True))
    cases.append(("tr_EM_aug_ht_ptshare_n2", EM, "RotatE", 1, 2, "ht", "random_pt", "margin", True, True))
    cases.append(("tr_EM_share_h_pt_n2", EM, "DistMult", 0, 2, "h", "random_pt", "logsigmoid", False, True))
    cases.append(("tr_SM_h_flat_n4", SM, "ComplEx", 0, 4, "h", "random_flat", "logsigmoid", False, True))
    cases.append(("tr_SM_ht_flat_n2", SM, "RotatE", 2, 2, "ht", "random_flat", "margin", False, True))
    cases.append(("tr_SM_ht_pt_n2", SM, "DistMult", 0, 2, "ht", "random_pt", "ssce", False, False))
    cases.append(("tr_EM_random_bs_n2", EM, "TransE", 1, 2, "t", "random_flat", "logsigmoid", False, True, "random"))
    names = []
    for c in cases:
        name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing = c[:10]
        bk = c[10] if len(c) > 10 else "rigid"
        run_bess_case(name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing, bk)
        names.append(name)
    put("bess", "cases", np.array(names))



def gen_bess_local() -> None:
    """EmbeddingMoving with `local_sampling=True` (bess.py:340-347, 383-393): tails travel, negatives are rows
    of the processing shard; with and without augmentation, shared and per-triple negatives."""
    EM = EmbeddingMovingBessKGE
    cases = [
        ("tr_EM_loc_aug_t_flat_n2", EM, "TransE", 1, 2, "t", "random_flat_local", "ssce", True, True),
        ("tr_EM_loc_aug_ht_flat_n4", EM, "ComplEx", 0, 4, "ht", "random_flat_local", "logsigmoid", True, True),
        ("tr_EM_loc_aug_h_flat_n2", EM, "DistMult", 0, 2, "h", "random_flat_local", "logsigmoid", True, True),
        ("tr_EM_loc_aug_t_ptshare_n2", EM, "RotatE", 1, 2, "t", "random_pt_local", "margin", True, True),
        ("tr_EM_loc_aug_ht_ptshare_n2", EM, "TransE", 2, 2, "ht", "random_pt_local", "logsigmoid", True, True),
        ("tr_EM_loc_t_pt_n2", EM, "ComplEx", 0, 2, "t", "random_pt_local", "logsigmoid", False, False),
        ("tr_EM_loc_ht_flat_n2", EM, "RotatE", 2, 2, "ht", "random_flat_local", "margin", False, True),
    ]
    names = []
    for name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing in cases:
        run_bess_case(name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing, fix="bess_local")
        names.append(name)
    put("bess_local", "cases", np.array(names))


def gen_bess_half() -> None:
    """The reference's fp16 mode (`model.half()`, notebooks/3_wikikg2_fp16.ipynb:300-392; `scoring.py:194-197, 342`)
    run by the reference's own code on CPU: fp16 tables, fp16 score arithmetic, fp32 loss, autograd in fp16.
    TransE / RotatE with p = 1, shared flat negatives, `augment_negative`, sampled-softmax CE - the wikikg2
    recipe (BASELINE configs[3]) - n_shard 1 and 2, widths the packed-fp16 kernels take (W % 32 == 0) and one
    they do not; plus log-sigmoid, per-triple and bilinear cases so that every fp16 code path has a
    reference-made fixture."""
    EM, SM = EmbeddingMovingBessKGE, ScoreMovingBessKGE
    cases = [
        # name, class, scorer, p, n, scheme, negatives, loss, augment, sharing, d
        ("h16_EM_TransE1_aug_t_flat_n1", EM, "TransE", 1, 1, "t", "random_flat", "ssce", True, True, 32),
        ("h16_EM_TransE1_aug_t_flat_n2", EM, "TransE", 1, 2, "t", "random_flat", "ssce", True, True, 32),
        ("h16_EM_TransE1_aug_t_flat_n2_d64", EM, "TransE", 1, 2, "t", "random_flat", "ssce", True, True, 64),
        ("h16_EM_RotatE1_aug_t_flat_n1", EM, "RotatE", 1, 1, "t", "random_flat", "ssce", True, True, 16),
        ("h16_EM_RotatE1_aug_t_flat_n2", EM, "RotatE", 1, 2, "t", "random_flat", "ssce", True, True, 32),
        ("h16_EM_TransE1_aug_ht_flat_n2", EM, "TransE", 1, 2, "ht", "random_flat", "ssce", True, True, 32),
        ("h16_EM_TransE1_t_flat_n1_d8", EM, "TransE", 1, 1, "t", "random_flat", "logsigmoid", False, True, 8),
        ("h16_EM_TransE1_h_pt_n2", EM, "TransE", 1, 2, "h", "random_pt", "logsigmoid", False, False, 32),
        ("h16_SM_RotatE1_t_pt_n2", SM, "RotatE", 1, 2, "t", "random_pt", "logsigmoid", False, False, 16),
        ("h16_EM_TransE2_aug_t_flat_n2", EM, "TransE", 2, 2, "t", "random_flat", "ssce", True, True, 32),
        ("h16_EM_ComplEx0_aug_t_flat_n2", EM, "ComplEx", 0, 2, "t", "random_flat", "ssce", True, True, 16),
        ("h16_EM_DistMult0_h_pt_n1", EM, "DistMult", 0, 1, "h", "random_pt", "logsigmoid", False, False, 32),
    ]
    names = []
    for name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing, d in cases:
        run_bess_case(name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing, fix="bess_half", half=True, d=d)
        names.append(name)
    put("bess_half", "cases", np.array(names))


def gen_bess_affine() -> None:
    """PairRE / TripleRE / InterHT / TranS through the reference's BessKGE.forward + autograd."""
    EM, SM = EmbeddingMovingBessKGE, ScoreMovingBessKGE
    cases = []
    for scorer, p in AFFINE_SCORERS:
        sn = f"{scorer}{p}"
        cases.append((f"tr_EM_{sn}_t_flat_n1", EM, scorer, p, 1, "t", "random_flat", "logsigmoid", False, True))
        cases.append((f"tr_EM_{sn}_h_pt_n1", EM, scorer, p, 1, "h", "random_pt", "logsigmoid", False, False))
        cases.append((f"tr_EM_{sn}_ht_pt_n2", EM, scorer, p, 2, "ht", "random_pt", "ssce", False, False))
        cases.append((f"tr_SM_{sn}_t_pt_n2", SM, scorer, p, 2, "t", "random_pt", "logsigmoid", False, False))
        cases.append((f"tr_SM_{sn}_ht_flat_n2", SM, scorer, p, 2, "ht", "random_flat", "margin", False, True))
    names = []
    for name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing in cases:
        run_bess_case(name, mcls, scorer, p, n, scheme, nk, loss, aug, sharing, fix="bess_affine")
        names.append(name)
    put("bess_affine", "cases", np.array(names))


# --------------------------------------------------------------------------- #
def gen_metric() -> None:
    """Evaluation.ranks_from_scores / ranks_from_indices / metrics (metric.py:74-273)."""
    fix = "metric"
    torch.manual_seed(4)
    pos = torch.randn(37) * 2
    cand = torch.randn(37, 23) * 2
    cand[:5, :7] = pos[:5, None]  # ties
    cand[5] = pos[5] + 1.0  # everything better
    pos[6] = float("nan")
    truth = torch.randint(40, (37,))
    ids = torch.stack([torch.randperm(40)[:9] for _ in range(37)])
    mask = torch.rand(37) > 0.3
    for k, v in dict(pos=pos, cand=cand, truth=truth, ids=ids, mask=mask).items():
        put(fix, k, v)
    for mode in ["optimistic", "pessimistic", "average"]:
        for winf in [False, True]:
            for red in ["none", "sum"]:
                ev = Evaluation(["mrr", "hits@1", "hits@5", "hits@10"], mode=mode, worst_rank_infty=winf,
                                reduction=red, return_ranks=True)
                key = f"{mode}_{int(winf)}_{red}_"
                r1 = ev.ranks_from_scores(pos.clone(), cand)
                r2 = ev.ranks_from_indices(truth, ids)
                put(fix, key + "ranks_scores", r1)
                put(fix, key + "ranks_indices", r2)
                put(fix, key + "names", np.array(list(ev.metrics.keys())))
                put(fix, key + "stacked_scores", ev.stacked_metrics_from_ranks(r1, mask))
                put(fix, key + "stacked_indices", ev.stacked_metrics_from_ranks(r2))


def _replica_modules(model: Any, score_fn: Any, ent: torch.Tensor, rel: torch.Tensor, n_shard: int) -> List[Any]:
    reps = []
    for r in range(n_shard):
        m = copy.copy(model)
        m._modules = dict(model._modules)
        sf = copy.copy(score_fn)
        sf._parameters = dict(score_fn._parameters)
        sf.entity_embedding = torch.nn.Parameter(ent[r].clone())
        sf.relation_embedding = torch.nn.Parameter(rel.clone())
        m.score_fn = sf
        m._parameters = dict(model._parameters)
        m.entity_embedding = sf.entity_embedding
        reps.append(m)
    return reps


def run_query_case(case: str, kind: str, scorer: str, p: int, n_shard: int, scheme: str, cand_kind: str,
                   k: int = 5, window: int = 7) -> None:
    """TopKQueryBessKGE (kind "topk") or AllScoresBESS (kind "all") on h_shard / t_shard queries."""
    fix = "topk" if kind == "topk" else "allscores"
    seed = 1234
    n_entity, n_rel, n_triple, d, bps, shard_bs, n_cand = 150, 5, 300, 8, 2, 12, 160
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    torch.manual_seed(zlib.crc32(case.encode()) % (2**31))
    W, Wr = widths(scorer, d)
    sharding = Sharding.create(n_entity, n_shard, seed=seed)
    ent = torch.randn(n_shard, sharding.max_entity_per_shard, W)
    rel = torch.randn(n_rel, Wr)
    if half:
        # the reference's fp16 mode is `model.half()` (notebooks/3_wikikg2_fp16.ipynb:300-392): the tables and
        # every tensor of the score computation are fp16, the loss is computed in fp32 (bess.py:254-260).
        # The fixture stores the fp16 tables; the replicas below run the reference's forward / autograd on
        # them with torch's CPU half kernels (element-wise results rounded to fp16, reductions accumulated in
        # fp32 and rounded once)
        ent, rel = ent.half(), rel.half()
    triples = np.stack([rng.integers(n_entity, size=n_triple), rng.integers(n_rel, size=n_triple),
                        rng.integers(n_entity, size=n_triple)], axis=1)
    flat = cand_kind in ("all", "flat")
    outer = 1 if flat else n_triple
    nh = rng.integers(n_entity, size=(outer, n_cand)).astype(np.int32)
    nt = rng.integers(n_entity, size=(outer, n_cand)).astype(np.int32)
    ds = make_dataset(n_entity, n_rel, triples, None, nh, nt, part="test")
    mode = "h_shard" if scheme == "t" else "t_shard"
    pts = PartitionedTripleSet.create_from_dataset(ds, "test", sharding, partition_mode=mode)
    if cand_kind == "all":
        ns = PlaceholderNegativeSampler(corruption_scheme=scheme, seed=seed)
    else:
        ns = TripleBasedShardedNegativeSampler(pts.neg_heads, pts.neg_tails, sharding, corruption_scheme=scheme,
                                               seed=seed, return_sort_idx=False, mask_on_gather=True)
    bs = RigidShardedBatchSampler(partitioned_triple_set=pts, negative_sampler=ns, shard_bs=shard_bs,
                                  batches_per_step=bps, seed=seed, duplicate_batch=False, return_triple_idx=True)
    score_fn = scorer_factory(scorer, p, flat, sharding, n_rel, d, ent, rel)
    if kind == "topk":
        ev = Evaluation(["mrr", "hits@1", "hits@5"], mode="average", reduction="none", return_ranks=True)
        model = TopKQueryBessKGE(k=k, candidate_sampler=ns, score_fn=score_fn, evaluation=ev, return_scores=True,
                                 window_size=window)
    else:
        model = AllScoresBESS(ns, score_fn, window_size=window)
    batch = bs[next(iter(bs.get_dataloader_sampler(shuffle=False)))]
    p_ = case + "_"
    meta = dict(n_entity=n_entity, n_rel=n_rel, n_triple=n_triple, d=d, bps=bps, shard_bs=shard_bs, n_cand=n_cand,
                n_shard=n_shard, norm=p, k=k, window=window, flat=int(flat))
    put(fix, p_ + "meta_keys", np.array(list(meta.keys())))
    put(fix, p_ + "meta_vals", np.array(list(meta.values())))
    put(fix, p_ + "strs", np.array([kind, scorer, scheme, cand_kind]))
    put(fix, p_ + "entity_table", ent)
    put(fix, p_ + "relation_table", rel)
    if scorer == "ConvE":  # the query network, in the (default) train mode: batch statistics, no dropout
        for k, v in net_state(score_fn).items():
            put(fix, p_ + "net_" + k, v)
    put(fix, p_ + "triples", triples)
    put(fix, p_ + "neg_heads", nh)
    put(fix, p_ + "neg_tails", nt)
    put(fix, p_ + "triple_sort_idx", pts.triple_sort_idx)
    for kk, v in batch.items():
        put(fix, p_ + "batch_" + kk, v)
    reps = _replica_modules(model, score_fn, ent, rel, n_shard)
    keys = ["relation", "head", "tail", "negative", "triple_mask", "negative_mask"]
    outs: Dict[str, List[Any]] = {}
    n_step = model.n_step if kind == "all" else 1
    for it in range(bps):
        for step in range(n_step):

            def fn(r: int) -> Dict[str, torch.Tensor]:
                kw = {kk: batch[kk][it, r].unsqueeze(0) for kk in keys if kk in batch}
                if kind == "all":
                    kw = {kk: v for kk, v in kw.items() if kk in ("relation", "head", "tail")}
                    kw.pop("tail" if scheme == "t" else "head", None)
                    out = reps[r](step=torch.tensor([[step]], dtype=torch.int32), **kw)
                    return dict(scores=out.detach().clone())
                out = reps[r](**kw)
                return {kk: v.detach().clone() for kk, v in out.items() if isinstance(v, torch.Tensor)}

            res = ref_shim.run_replicas(n_shard, fn)
            for kk in res[0].keys():
                outs.setdefault(kk, []).append(torch.stack([res[r][kk] for r in range(n_shard)]))
    for kk, v in outs.items():
        put(fix, p_ + "out_" + kk, torch.stack(v).reshape(bps, n_step, n_shard, *v[0].shape[1:]) if kind == "all"
            else torch.stack(v))


def gen_topk() -> None:
    names = []
    for scorer, p in [("DistMult", 0), ("TransE", 1), ("ComplEx", 0), ("RotatE", 2)]:
        for scheme in ["h", "t"]:
            for cand in ["all", "flat", "pt"]:
                for n in ([1, 4] if scorer == "DistMult" else [2]):
                    name = f"topk_{scorer}{p}_{scheme}_{cand}_n{n}"
                    run_query_case(name, "topk", scorer, p, n, scheme, cand)
                    names.append(name)
    put("topk", "cases", np.array(names))


def gen_allscores() -> None:
    names = []
    for scorer, p in [("ComplEx", 0), ("TransE", 1)]:
        for scheme in ["h", "t"]:
            for n in [1, 4]:
                name = f"all_{scorer}{p}_{scheme}_n{n}"
                run_query_case(name, "all", scorer, p, n, scheme, "all", window=13)
                names.append(name)
    put("allscores", "cases", np.array(names))


def main() -> None:
    only = sys.argv[1:]
    gens = dict(
        sharding=gen_sharding,
        partition=gen_partition,
        negative_sampler=gen_negative_sampler,
        batch_sampler=gen_batch_sampler,
        embedding=gen_embedding,
        scoring=gen_scoring,
        loss=gen_loss,
        bess=gen_bess,
        metric=gen_metric,
        topk=gen_topk,
        allscores=gen_allscores,
        scoring_affine=gen_scoring_affine,
        bess_affine=gen_bess_affine,
        dataset=gen_dataset,
        scoring_boxe=gen_scoring_boxe,
        bess_boxe=gen_bess_boxe,
        scoring_conve=gen_scoring_conve,
        bess_conve=gen_bess_conve,
        bess_local=gen_bess_local,
        bess_half=gen_bess_half,
        scoring_lp=gen_scoring_lp,
        bess_lp=gen_bess_lp,
    )
    for name, g in gens.items():
        if only and name not in only:
            continue
        g()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **OUT[name])
        print(f"{name}: {len(OUT[name])} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
