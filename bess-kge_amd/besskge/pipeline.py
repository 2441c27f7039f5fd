"""High-level inference pipeline: scores of queries against all entities.

Mirror of the reference interface (`besskge/pipeline.py:23-320`):
`AllScoresPipeline(batch_sampler, corruption_scheme, score_fn, evaluation,
filter_triples, candidate_ents, return_scores, return_topk, k, window_size)`;
calling it returns `scores`, `topk_global_id`, `triple_idx`, `metrics`,
`metrics_avg`, `ranks` as configured.  The device work is `AllScoresBESS`
(window scoring on the shards + all-to-all); the re-assembly, filtering and
ranking of the `[queries, n_entity]` score matrix stays on the device and only
the requested results are returned on the host.
"""

from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch
from numpy.typing import NDArray

from besskge import _native as nat
from besskge import runtime
from besskge.batch_sampler import ShardedBatchSampler
from besskge.collectives import ReplicaGroup
from besskge.negative_sampler import PlaceholderNegativeSampler
from besskge.query import AllScoresBESS, topk_merge
from besskge.scoring import BaseScoreFunction
from besskge.utils import get_entity_filter


def _counting_scorer(score_fn: BaseScoreFunction) -> bool:
    """TransE / RotatE / DistMult / ComplEx: the scorers whose all-entity kernels can count ranks."""
    try:
        return int(score_fn.kernel_desc().scorer) <= nat.COMPLEX
    except (AttributeError, NotImplementedError, ValueError):
        return False


def rank_filter_pairs(flt: torch.Tensor, keep: torch.Tensor, truth: torch.Tensor, rows: int, shard_bs: int,
                      candidate: Optional[torch.Tensor] = None) -> "tuple[torch.Tensor, torch.Tensor]":
    """The `rank_filter` input of `AllScoresBESS.rank_counts_replicas` for one sampler batch.

    :param flt: (z, 2) rows (i, e) of `get_entity_filter`: entity e is filtered for the i-th KEPT triple.
    :param keep: flat [rows * shard_bs] triple mask (rows = micro-batches x shards).
    :param truth: flat [rows * shard_bs] global id of every slot's true completion.
    :param candidate: bool [n_entity], with `candidate_ents`: pairs naming an entity outside the subset are
        dropped - the counting pass never sees those entities (the reference sets them to -inf anyway,
        pipeline.py:247-250), so there is nothing to subtract for them.
    :return: (`rank_filter` [rows, P, 2] int32 - (position in the row's micro-batch, entity), -1 padding, every
        pair once, none naming its query's true completion -, number of such pairs per kept triple [n_kept]).
    """
    slot = keep.nonzero().reshape(-1)[flt[:, 0]]  # position in the flat [rows * shard_bs] batch
    pairs = torch.stack([slot, flt[:, 1].to(slot.dtype)], dim=1)
    pairs = pairs[pairs[:, 1] != truth[pairs[:, 0]].to(pairs.dtype)]  # (the truth is left out anyway)
    if candidate is not None:
        pairs = pairs[candidate[pairs[:, 1]]]
    pairs = torch.unique(pairs, dim=0)  # sorted by slot
    per_slot = torch.zeros(keep.numel(), dtype=torch.int64)
    per_slot.index_add_(0, pairs[:, 0], torch.ones(len(pairs), dtype=torch.int64))
    row_of = pairs[:, 0] // shard_bs
    per_row = torch.bincount(row_of, minlength=rows)
    start = torch.cumsum(per_row, 0) - per_row
    P = int(per_row.max()) if len(pairs) else 0
    filt = torch.full((rows, max(P, 1), 2), -1, dtype=torch.int32)
    if len(pairs):
        k = torch.arange(len(pairs)) - start[row_of]
        filt[row_of, k, 0] = (pairs[:, 0] % shard_bs).to(torch.int32)
        filt[row_of, k, 1] = pairs[:, 1].to(torch.int32)
    return filt, per_slot[keep]


def ranks_from_counts(counts: torch.Tensor, pos: torch.Tensor, n_masked: torch.Tensor, n_entity: int, mode: str,
                      worst_rank_infty: bool) -> torch.Tensor:
    """`Evaluation.ranks_from_scores` (reference metric.py:129-182) from the counts of the entities scoring above
    (`counts[:, 0]`) / exactly (`counts[:, 1]`) as the true completion, taken over the entities that the reference
    does NOT set to -inf; `n_masked` entities per query are set to -inf there (the true completion, the filtered
    completions and - with `candidate_ents` - every entity outside the subset, each counted once): they only ever
    tie with a -inf positive score (a NaN one, `metric.py:152`)."""
    gt, eq = counts[:, 0].float(), counts[:, 1].float()
    ge = gt + eq + torch.where(pos == -torch.inf, n_masked.float(), torch.zeros_like(gt))
    n_cand = float(n_entity)
    if mode == "optimistic":
        better, worst = gt, gt == n_cand
    elif mode == "pessimistic":
        better, worst = ge, ge == n_cand
    else:
        better, worst = 0.5 * (gt + ge), (gt == n_cand) | (ge == n_cand)
    rank = 1.0 + better
    if worst_rank_infty:
        rank = torch.where(worst, torch.full_like(rank, torch.inf), rank)
    return rank


class AllScoresPipeline(torch.nn.Module):
    """Scores (and metrics) of (h, r, ?) / (?, r, t) queries against all entities,
    with optional filtering of known triples and restriction to candidate entities."""

    def __init__(
        self,
        batch_sampler: ShardedBatchSampler,
        corruption_scheme: str,
        score_fn: BaseScoreFunction,
        evaluation: Optional[Any] = None,
        filter_triples: Optional[List[Union[torch.Tensor, NDArray[np.int32]]]] = None,
        candidate_ents: Optional[Union[torch.Tensor, NDArray[np.int32]]] = None,
        return_scores: bool = False,
        return_topk: bool = False,
        k: int = 10,
        window_size: int = 1000,
        use_ipu_model: bool = False,
        group: Optional[ReplicaGroup] = None,
        device: Optional[torch.device] = None,
        fused_ranks: bool = True,
    ) -> None:
        """
        :param batch_sampler: sampler over "h_shard" / "t_shard" partitioned queries.
        :param corruption_scheme: "t" scores (h, r, ?), "h" scores (?, r, t).
        :param score_fn: trained scoring function.
        :param evaluation: `besskge.metric.Evaluation`.
        :param filter_triples: triples (global ids) whose completions are filtered out.
        :param candidate_ents: global ids of the only entities to consider.
        :param return_scores: return the filtered `[queries, n_entity]` scores (host memory!).
        :param return_topk: return the k best global ids per query (after filtering).
        :param window_size: entities of each shard scored per device call.
        :param use_ipu_model: accepted for call compatibility, ignored.
        :param group / device: replica group and HIP device (default: all shards
            in this process on the current device).
        :param fused_ranks: when only metrics / ranks are asked for (no scores, no top-k) count the entities that beat the true completion in the scoring kernel's
            epilogue instead of assembling the `[queries, n_entity]` score matrix
            (`AllScoresBESS.rank_counts_replicas`).  The same ranks to the last bit as the matrix path
            whenever that scores its windows with the same kernel as the all-entity pass (the positives'
            and the filtered completions' scores are taken in that kernel's arithmetic too:
            `bess_neg_score_shared_fwd_pairs`); where `window_size` makes the matrix path take another
            kernel (the split-fp16 product needs 256 output tiles) ranks may differ by one at scores that
            agree to a rounding error.
        """
        super().__init__()
        if not (evaluation or return_scores):
            raise ValueError("Nothing to return. Provide `evaluation` or set `return_scores=True`")
        if corruption_scheme not in ["h", "t"]:
            raise ValueError("corruption_scheme needs to be either 'h' or 't'")
        want_mode = "t_shard" if corruption_scheme == "h" else "h_shard"
        if batch_sampler.triple_partition_mode != want_mode:
            raise ValueError(
                f"Corruption scheme '{corruption_scheme}' requires '{want_mode.replace('_', '-')}'-partitioned triples")
        self.batch_sampler = batch_sampler
        self.corruption_scheme = corruption_scheme
        self.candidate_sampler = PlaceholderNegativeSampler(corruption_scheme=corruption_scheme)
        self.score_fn = score_fn
        self.evaluation = evaluation
        self.return_scores = return_scores
        self.return_topk = return_topk
        self.k = k
        self.window_size = window_size
        self.bess_module = AllScoresBESS(self.candidate_sampler, self.score_fn, self.window_size)
        self.dl = batch_sampler.get_dataloader(shuffle=False)
        self.runner = runtime.inference_model(
            self.bess_module, runtime.Options(device_iterations=batch_sampler.batches_per_step), group=group,
            device=device)
        sharding = self.bess_module.sharding
        self.filter_triples: Optional[torch.Tensor] = None
        if filter_triples:
            # global ids of the sampler's (locally indexed) triples
            col = 0 if batch_sampler.triple_partition_mode == "h_shard" else 2
            glob = np.copy(batch_sampler.triples)
            bounds = np.concatenate([np.array([0]), np.cumsum(batch_sampler.triple_counts)])
            for i in range(len(bounds) - 1):
                sl = slice(bounds[i], bounds[i + 1])
                glob[sl, col] = sharding.shard_and_idx_to_entity[i][glob[sl, col]]
            self.triples = torch.from_numpy(glob)
            self.filter_triples = torch.concat(
                [t if isinstance(t, torch.Tensor) else torch.from_numpy(t) for t in filter_triples], dim=0)
        self.fused_ranks = bool(fused_ranks and evaluation and not return_scores and not return_topk
                                and _counting_scorer(score_fn))
        if self.fused_ranks:
            self.bess_module.set_rank_candidates(candidate_ents)
        self.candidate_mask: Optional[torch.Tensor] = None
        self._is_candidate: Optional[torch.Tensor] = None
        if candidate_ents is not None:
            self.candidate_mask = torch.from_numpy(np.setdiff1d(np.arange(sharding.n_entity), candidate_ents))
            self._is_candidate = torch.ones(sharding.n_entity, dtype=torch.bool)
            self._is_candidate[self.candidate_mask] = False
        # column order of the assembled scores -> global entity id (first occurrence of every entity)
        ws, n_step, M = self.window_size, self.bess_module.n_step, sharding.max_entity_per_shard
        cols = []
        for i in range(n_step):
            ent_slice = np.minimum(i * ws + np.arange(ws), M - 1)
            cols.append(sharding.shard_and_idx_to_entity[:, ent_slice].flatten())
        self._first = torch.from_numpy(np.unique(np.concatenate(cols), return_index=True)[1])

    def _ranks_by_counting(self, inp: Dict[str, torch.Tensor], ground_truth: torch.Tensor,
                           triple_mask: torch.Tensor, triple_id: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """Ranks of one sampler batch from counts made in the scoring kernels' epilogues; None when the
        matrix-core product met operands outside its fp16 range (the caller then takes the matrix path)."""
        ev = self.evaluation
        sharding = self.bess_module.sharding
        n = sharding.n_shard
        rows, shard_bs = ground_truth.flatten(end_dim=1).shape  # [micro-batches * n_shard, shard_bs]
        keep = triple_mask.flatten()
        truth = ground_truth.flatten()
        extra = dict(rank_truth=ground_truth.flatten(end_dim=1).to(torch.int32))
        n_masked = torch.ones(int(keep.sum()), dtype=torch.int64)  # entries the reference sets to -inf, per kept row
        if self.filter_triples is not None:
            if triple_id is None:
                raise ValueError("filtering needs a batch sampler with return_triple_idx=True")
            flt = get_entity_filter(self.triples[triple_id[triple_mask]], self.filter_triples,
                                    filter_mode=self.corruption_scheme)
            filt, per_kept = rank_filter_pairs(flt, keep, truth, rows, shard_bs, self._is_candidate)
            n_masked += per_kept
            extra["rank_filter"] = filt
        if self._is_candidate is not None:
            # entities outside the subset are -inf in the reference's matrix too (pipeline.py:247-250): they tie
            # with a NaN positive (-inf after metric.py:152) like the truth and the filtered completions do
            n_masked += sharding.n_entity - int(self._is_candidate.sum())
        step = torch.zeros((rows, 1), dtype=torch.int32)
        out = self.runner(step=step, **inp, **extra)
        dev = out["counts"].device
        counts = out["counts"].reshape(-1, 2)[keep.to(dev)]
        if bool(out["out_of_range"].any()):
            return None  # some shard's matrix-core product met operands outside the fp16 range: matrix path
        pos = out["pos_score"].reshape(-1)[keep.to(dev)]
        return ranks_from_counts(counts, pos, n_masked.to(dev), sharding.n_entity, ev.mode, ev.worst_rank_infty)

    def forward(self) -> Dict[str, Any]:
        """Run over the whole sampler."""
        ev = self.evaluation
        sharding = self.bess_module.sharding
        n, bps = sharding.n_shard, self.batch_sampler.batches_per_step
        dev = self.runner.device
        scores, ids, metrics, ranks, topk = [], [], [], [], []
        n_triple = 0
        gt_key = "head" if self.corruption_scheme == "h" else "tail"
        for batch in self.dl:
            triple_mask = batch.pop("triple_mask")
            ground_truth = batch.pop(gt_key) if gt_key in batch else None
            triple_id = batch.pop("triple_idx") if self.batch_sampler.return_triple_idx else None
            if triple_id is not None:
                ids.append(triple_id[triple_mask])
            n_triple += int(triple_mask.sum())
            inp = {k: v.flatten(end_dim=1) for k, v in batch.items()}
            if self.fused_ranks:
                assert ground_truth is not None, "Evaluation requires providing ground truth entities"
                r = self._ranks_by_counting(inp, ground_truth, triple_mask, triple_id)
                if r is not None:
                    metrics.append({m: v.cpu() for m, v in ev.dict_metrics_from_ranks(r).items()})
                    if ev.return_ranks:
                        ranks.append(r.cpu())
                    continue
            parts = []
            for i in range(self.bess_module.n_step):
                step = torch.full((n * bps, 1), i, dtype=torch.int32)
                parts.append(self.runner(step=step, **inp))
            all_scores = torch.concat(parts, dim=-1)  # [queries, n_step * n * ws] on the device
            keep = triple_mask.flatten().to(dev)
            sc = all_scores[keep][:, self._first.to(dev)][:, : sharding.n_entity].float()
            if self.candidate_mask is not None:
                sc[:, self.candidate_mask.to(dev)] = -torch.inf
            rows = torch.arange(sc.shape[0], device=dev)
            truth = true_scores = None
            if ground_truth is not None:
                truth = ground_truth[triple_mask].to(dev).long()
                true_scores = sc[rows, truth].clone()
            if self.filter_triples is not None:
                if triple_id is None:
                    raise ValueError("filtering needs a batch sampler with return_triple_idx=True")
                flt = get_entity_filter(self.triples[triple_id[triple_mask]], self.filter_triples,
                                        filter_mode=self.corruption_scheme).to(dev)
                sc[flt[:, 0], flt[:, 1]] = -torch.inf
            if ev:
                assert truth is not None, "Evaluation requires providing ground truth entities"
                sc[rows, truth] = -torch.inf  # the true completion does not compete with itself
                r = ev.ranks_from_scores(true_scores, sc)
                metrics.append({m: v.cpu() for m, v in ev.dict_metrics_from_ranks(r).items()})
                if ev.return_ranks:
                    ranks.append(r.cpu())
            if truth is not None:
                sc[rows, truth] = true_scores
            if self.return_scores:
                scores.append(sc.cpu())
            if self.return_topk:
                top_s = torch.full((sc.shape[0], self.k), -torch.inf, dtype=torch.float32, device=dev)
                top_i = torch.zeros((sc.shape[0], self.k), dtype=torch.int32, device=dev)
                topk_merge(sc.contiguous(), top_s, top_i)  # column == global entity id
                topk.append(top_i.cpu().long())
        out: Dict[str, Any] = dict()
        if scores:
            out["scores"] = torch.concat(scores, dim=0)
        if topk:
            out["topk_global_id"] = torch.concat(topk, dim=0)
        if ids:
            out["triple_idx"] = torch.concat(ids, dim=0)
        if ev:
            final = {m: ev.reduction(torch.concat([met[m].reshape(-1) for met in metrics])) for m in metrics[0]}
            out["metrics"] = final
            out["metrics_avg"] = {m: v.sum() / n_triple for m, v in final.items()}
            if ranks:
                out["ranks"] = torch.concat(ranks, dim=0)
        return out
