"""Inference against large candidate sets: top-k completions and all scores.

Mirror of the reference interface (`besskge/bess.py:606-1062`):
`TopKQueryBessKGE` and `AllScoresBESS`, used with "h_shard" / "t_shard"
partitioned (h, r, ?) / (?, r, t) queries.  Same data flow as the reference
(and as `ScoreMovingBessKGE`): queries are all-gathered, every shard scores them
against *its own* rows, per-query results go back with an all-to-all.

MI355X mapping
  * scoring a window of the shard is the shared-negative kernel (K4: MFMA GEMM
    for DistMult / ComplEx, VALU distance matrix for TransE / RotatE) on a
    contiguous slice of the shard - or the per-triple kernel (K5) for
    query-specific candidate lists;
  * the reference's `torch.topk(concat(window, running)) + gather_indices` per
    window (`bess.py:802-814`) is `bess_topk_update` (K11): a streaming top-k
    with the running list in registers;
  * the sliding window is a tiling choice: results do not depend on its size
    (the reference's `window_size` is accepted and used as the *minimum* tile;
    the device tile is sized so that the score tile stays a few MiB).
"""

from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch

from besskge import _native as nat
from besskge._native import RowSource
from besskge.collectives import ReplicaGroup, SingleProcessGroup
from besskge.negative_sampler import (
    PlaceholderNegativeSampler,
    TripleBasedShardedNegativeSampler,
)
from besskge.scoring import BaseScoreFunction

BAD_NEGATIVE_SCORE = -50000.0
_Batch = Dict[str, torch.Tensor]


def _i32(x: torch.Tensor) -> torch.Tensor:
    return (x if x.dtype == torch.int32 else x.to(torch.int32)).contiguous()


#: entries the streaming top-k kernel keeps per query (two registers per lane)
KERNEL_LIST_MAX = 128


def topk_merge(scores: torch.Tensor, best_s: torch.Tensor, best_i: torch.Tensor, ids: Optional[torch.Tensor] = None,
               id_base: int = 0, mask: Optional[torch.Tensor] = None, flags: Optional[torch.Tensor] = None) -> None:
    """`nat.topk_update` for lists of any length (the reference's `torch.topk` takes any k: bess.py:807-814).
    Up to 128 entries per query the streaming kernel keeps the list in registers; longer lists are merged by
    `torch.topk` over [window | running list] on the device - the reference's own formulation - in column
    chunks that bound the temporary (equal scores: order unspecified, as in the reference)."""
    kk = int(best_s.shape[1])
    if kk <= KERNEL_LIST_MAX:
        nat.topk_update(scores, best_s, best_i, ids=ids, id_base=id_base, mask=mask, flags=flags)
        return
    if flags is not None:
        raise ValueError("pruned score tiles are read by the streaming kernel only (lists of up to 128 entries)")
    R, L = int(scores.shape[0]), int(scores.shape[1])
    chunk = max(kk, (1 << 28) // max(1, R))  # <= 1 GiB of fp32 candidates per merge
    for c0 in range(0, L, chunk):
        c1 = min(L, c0 + chunk)
        sc = scores[:, c0:c1]
        if mask is not None:
            sc = sc + BAD_NEGATIVE_SCORE * (~mask[:, c0:c1]).to(sc.dtype)
        if ids is not None:
            cid = ids[:, c0:c1].expand(R, c1 - c0)
        else:
            cid = (id_base + torch.arange(c0, c1, dtype=torch.int32, device=scores.device))[None, :].expand(R, c1 - c0)
        top_s, pos = torch.topk(torch.cat([best_s, sc], dim=1), kk, dim=1)  # running entries first: they win ties
        best_i.copy_(torch.gather(torch.cat([best_i, cid], dim=1), 1, pos))
        best_s.copy_(top_s)


class _QueryModule(torch.nn.Module):
    """Shared plumbing: replica group, shard placement, query gathering."""

    def __init__(self, score_fn: BaseScoreFunction, candidate_sampler: Any) -> None:
        super().__init__()
        self.sharding = score_fn.sharding
        self.score_fn = score_fn
        self.negative_sampler = candidate_sampler
        if candidate_sampler.corruption_scheme not in ["h", "t"]:
            raise ValueError(f"{type(self).__name__} only support 'h', 't' corruption scheme")
        self.entity_embedding = self.score_fn.entity_embedding
        self.entity_embedding_size: int = self.entity_embedding.shape[-1]
        self.replica_group: Optional[ReplicaGroup] = None
        self._shard_slot: Optional[Dict[int, int]] = None

    def attach(self, group: ReplicaGroup, shard_slot: Optional[Dict[int, int]] = None) -> None:
        """Bind to a replica group (see :meth:`besskge.bess.BessKGE.attach`)."""
        if group.n_shard != self.sharding.n_shard:
            raise ValueError(f"group has {group.n_shard} replicas, sharding {self.sharding.n_shard}")
        self.replica_group = group
        self._shard_slot = shard_slot

    def _group(self) -> ReplicaGroup:
        if self.replica_group is None:
            self.replica_group = SingleProcessGroup(self.sharding.n_shard)
        return self.replica_group

    def _local_table(self, shard: int) -> torch.Tensor:
        emb = self.score_fn.entity_embedding
        slot = shard if self._shard_slot is None else self._shard_slot[shard]
        if emb.dim() != 3 or slot >= emb.shape[0]:
            raise RuntimeError(f"entity_embedding {tuple(emb.shape)} does not hold shard {shard}")
        return emb.data[slot]

    def _gather_queries(self, batches: List[_Batch]) -> List[torch.Tensor]:
        """Query matrices [n_shard * shard_bs, W] (one per local replica): the known
        entity of every query of every replica, transformed with its relation."""
        group = self._group()
        fn = self.score_fn
        scheme = self.negative_sampler.corruption_scheme
        key = "tail" if scheme == "h" else "head"
        side = nat.CORRUPT_HEAD if scheme == "h" else nat.CORRUPT_TAIL
        rels, rows = [], []
        for shard, b in zip(group.local_shards, batches):
            table = self._local_table(shard)
            if key not in b or b[key] is None:
                raise ValueError(f"corruption scheme '{scheme}' needs the `{key}` indices of the queries")
            ent = _i32(b[key].squeeze(0).to(table.device)).reshape(-1)
            rels.append(_i32(b["relation"].squeeze(0).to(table.device)).reshape(-1))
            rows.append(nat.gather_rows(table, ent))
        rel_all = group.all_gather(rels)  # [n, shard_bs]
        rows_all = group.all_gather(rows)  # [n, shard_bs, W]
        desc = fn.kernel_desc()
        W = self.entity_embedding_size
        return [fn.query_fwd(side, RowSource(x.reshape(-1, W)), r.reshape(-1).contiguous())[0]
                for x, r in zip(rows_all, rel_all)]


class TopKQueryBessKGE(_QueryModule):
    """Top-k completions of (h, r, ?) / (?, r, t) queries against all entities or
    against given candidates (reference `bess.py:606-921`).  Inference only."""

    def __init__(
        self,
        k: int,
        candidate_sampler: Union[TripleBasedShardedNegativeSampler, PlaceholderNegativeSampler],
        score_fn: BaseScoreFunction,
        evaluation: Optional[Any] = None,
        return_scores: bool = False,
        window_size: int = 100,
    ) -> None:
        """
        :param k: number of completions returned per query (any; up to k + 1 = 128 the running lists live in the
            streaming kernel's registers, beyond that they are merged with `torch.topk` on the device).
        :param candidate_sampler: `PlaceholderNegativeSampler` (score against
            every entity) or a `TripleBasedShardedNegativeSampler` built with
            `mask_on_gather=True`.
        :param score_fn: scoring function.
        :param evaluation: `besskge.metric.Evaluation` (needs the ground truth).
        :param return_scores: also return the scores of the k completions.
        :param window_size: minimum number of candidates scored per tile.
        """
        super().__init__(score_fn, candidate_sampler)
        self.evaluation = evaluation
        self.return_scores = return_scores
        self.k = k
        self.window_size = window_size
        if self.negative_sampler.flat_negative_format:
            assert score_fn.negative_sample_sharing, "Using flat negative format requires negative sample sharing"
        elif score_fn.negative_sample_sharing:
            raise ValueError("Negative sample sharing cannot be used with non-flat triple-specific negatives")
        if isinstance(self.negative_sampler, TripleBasedShardedNegativeSampler):
            assert self.negative_sampler.mask_on_gather, (
                "TopKQueryBessKGE requires setting mask_on_gather=True in the candidate_sampler")

    def forward(
        self,
        relation: torch.Tensor,
        head: Optional[torch.Tensor] = None,
        tail: Optional[torch.Tensor] = None,
        negative: Optional[torch.Tensor] = None,
        triple_mask: Optional[torch.Tensor] = None,
        negative_mask: Optional[torch.Tensor] = None,
    ) -> Dict[str, Any]:
        """One micro-batch of one replica (see :meth:`forward_replicas`).

        :param relation: (1, shard_bs) relation ids.
        :param head / tail: (1, shard_bs); the known side holds rows of this
            shard, the other one (optional) the global id of the ground truth.
        :param negative: (1, n_shard, B, padded) candidate rows, B = 1 or
            shard_bs; None = every entity.
        :param triple_mask: (1, shard_bs) queries that count for the metrics.
        :param negative_mask: (1, n_shard, B, padded) real (non padding) candidates.
        """
        b = dict(relation=relation, head=head, tail=tail, negative=negative, triple_mask=triple_mask,
                 negative_mask=negative_mask)
        if len(self._group().local_shards) != 1:
            raise RuntimeError("forward() steps a single replica; use forward_replicas()")
        return self.forward_replicas([{k: v for k, v in b.items() if v is not None}])[0]

    def _sharding_on(self, dev: torch.device):
        """(shard_counts, shard_and_idx_to_entity) as int32 device tensors, uploaded once."""
        cache = self.__dict__.setdefault("_sharding_cache", {})
        if dev not in cache:
            cache[dev] = (
                torch.from_numpy(np.asarray(self.sharding.shard_counts)).to(device=dev, dtype=torch.int32),
                torch.from_numpy(np.asarray(self.sharding.shard_and_idx_to_entity)).to(device=dev, dtype=torch.int32),
            )
        return cache[dev]

    #: bytes of the fp32 score tile [n_query, tile] between the scoring kernel and the top-k kernel
    score_tile_bytes = 1 << 30
    #: candidates of the first (unpruned) tile of an all-entities pass; the next ones double up to the tile size
    first_tile = 8192
    #: prune the score tiles against the running k-th best scores (False: every score is written and read)
    prune_scores = True

    def _tile(self, n_query: int) -> int:
        # few, large launches (writing and re-reading the tile costs 8 B per score), at least
        # the reference's window
        return max(self.window_size, max(64, (self.score_tile_bytes // 4) // max(1, n_query)))

    def forward_replicas(self, batches: List[_Batch]) -> List[Dict[str, Any]]:
        group = self._group()
        n = group.n_shard
        fn = self.score_fn
        desc = fn.kernel_desc()
        kk = self.k + 1
        queries = self._gather_queries(batches)
        best_s, best_i = [], []
        for shard, b, q in zip(group.local_shards, batches, queries):
            table = self._local_table(shard)
            dev = table.device
            nq = int(q.shape[0])
            M = int(table.shape[0])
            bs = torch.full((nq, kk), BAD_NEGATIVE_SCORE, dtype=torch.float32, device=dev)
            bi = torch.full((nq, kk), M, dtype=torch.int32, device=dev)
            tile = self._tile(nq)
            if b.get("negative") is None:
                # Every entity of the shard.  After a first small tile the rows' k-th best scores are thresholds
                # for the scoring kernel itself: a block of 64 scores is written (and later read by the top-k
                # pass) only if one of them can still enter its row's list - the tiles grow geometrically while
                # the thresholds tighten, so that all but a few per cent of the score traffic is never made
                # (the reference materialises every window: bess.py:776-812)
                w0, step = 0, min(tile, max(self.window_size, self.first_tile))
                while w0 < M:
                    w1 = min(M, w0 + step)
                    if w0 == 0 or not self.prune_scores or kk > KERNEL_LIST_MAX:
                        sc = nat.neg_score_shared_fwd(desc, q, RowSource(table[w0:w1]), pad_ld=True)
                        topk_merge(sc, bs, bi, id_base=w0)
                    else:
                        thr = bs[:, kk - 1].contiguous()
                        sc, flags = nat.neg_score_shared_fwd_pruned(desc, q, RowSource(table[w0:w1]), thr)
                        nat.topk_update(sc, bs, bi, id_base=w0, flags=flags)
                    w0, step = w1, min(tile, 2 * step)
            else:
                if b.get("negative_mask") is None:
                    raise ValueError("candidates need their `negative_mask`")
                cand = _i32(b["negative"].squeeze(0).to(dev))
                mask = b["negative_mask"].squeeze(0).to(device=dev, dtype=torch.bool)
                if self.negative_sampler.flat_negative_format:
                    cand, mask = cand[0], mask[0]
                cand = cand.reshape(-1, cand.shape[-1]).contiguous()  # [1 | n * shard_bs, L]
                mask = mask.reshape(-1, mask.shape[-1]).contiguous()
                L = int(cand.shape[1])
                if cand.shape[0] == 1:
                    for w0 in range(0, L, tile):
                        w1 = min(L, w0 + tile)
                        ids = cand[:, w0:w1].contiguous()
                        sc = nat.neg_score_shared_fwd(desc, q, RowSource(table, ids.reshape(-1)), pad_ld=True)
                        topk_merge(sc, bs, bi, ids=ids, mask=mask[:, w0:w1].contiguous())
                else:
                    if cand.shape[0] != nq:
                        raise ValueError(f"{cand.shape[0]} candidate lists for {nq} gathered queries")
                    sc = nat.neg_score_pertriple_fwd(desc, q, RowSource(table, cand.reshape(-1)), L)
                    topk_merge(sc, bs, bi, ids=cand, mask=mask)
            best_s.append(bs.reshape(n, -1, kk))
            best_i.append(bi.reshape(n, -1, kk))
        # per-query lists back to the query's shard (C6)
        back_s = group.all_to_all(best_s)
        back_i = group.all_to_all(best_i)
        outs = []
        for b, s, i in zip(batches, back_s, back_i):
            dev = s.device
            cnt, tg = self._sharding_on(dev)
            pad = i >= cnt[:, None, None]  # padding rows of a shard (and the initial sentinel) never win
            s = s + BAD_NEGATIVE_SCORE * pad.to(s.dtype)
            M = tg.shape[1]
            src = torch.arange(n, device=dev)[:, None, None].expand_as(i)
            gid = tg[src, i.clamp(max=M - 1).long()]  # [n, shard_bs, kk]
            shard_bs = int(s.shape[1])
            flat_s = s.transpose(0, 1).reshape(shard_bs, n * kk).contiguous()
            flat_g = gid.transpose(0, 1).reshape(shard_bs, n * kk).contiguous()
            top_s = torch.full((shard_bs, self.k), -float("inf"), dtype=torch.float32, device=dev)
            top_g = torch.zeros((shard_bs, self.k), dtype=torch.int32, device=dev)
            topk_merge(flat_s, top_s, top_g, ids=flat_g)
            out: Dict[str, Any] = dict(topk_global_id=top_g)
            if self.return_scores:
                out["topk_scores"] = top_s.to(fn.relation_embedding.dtype)
            if self.evaluation:
                truth = b.get("tail" if self.negative_sampler.corruption_scheme == "t" else "head")
                assert truth is not None, "Evaluation requires providing ground truth entities"
                tm = b.get("triple_mask")
                tm = tm.flatten().to(dev) if tm is not None else None
                ranks = self.evaluation.ranks_from_indices(truth.squeeze(0).to(dev), top_g)
                if self.evaluation.return_ranks:
                    out["ranks"] = ranks
                out["metrics"] = self.evaluation.stacked_metrics_from_ranks(ranks, tm)
            outs.append(out)
        return outs


class AllScoresBESS(_QueryModule):
    """Scores of the queries against one window of every shard per call
    (reference `bess.py:924-1062`); `besskge.pipeline.AllScoresPipeline` loops
    over the windows.  Inference only."""

    def __init__(self, candidate_sampler: PlaceholderNegativeSampler, score_fn: BaseScoreFunction,
                 window_size: int = 1000) -> None:
        """
        :param candidate_sampler: `PlaceholderNegativeSampler` giving the corruption scheme.
        :param score_fn: scoring function (with negative sample sharing).
        :param window_size: entities of each shard scored per call.
        """
        super().__init__(score_fn, candidate_sampler)
        if not score_fn.negative_sample_sharing:
            raise ValueError("AllScoresBESS requires using negative sample sharing")
        if not isinstance(candidate_sampler, PlaceholderNegativeSampler):
            raise ValueError("AllScoresBESS requires a `PlaceholderNegativeSampler` candidate_sampler")
        self.window_size = window_size
        self.candidate = torch.arange(self.window_size, dtype=torch.int32)
        self.n_step = int(np.ceil(self.sharding.max_entity_per_shard / self.window_size))

    def forward(self, step: torch.Tensor, relation: torch.Tensor, head: Optional[torch.Tensor] = None,
                tail: Optional[torch.Tensor] = None, rank_truth: Optional[torch.Tensor] = None,
                rank_filter: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Scores (shard_bs, n_shard * window_size) of window `step` (one replica); with `rank_truth` the
        rank counts over all entities instead (`rank_counts_replicas`; `step` is not used then)."""
        b = dict(step=step, relation=relation, head=head, tail=tail, rank_truth=rank_truth, rank_filter=rank_filter)
        if len(self._group().local_shards) != 1:
            raise RuntimeError("forward() steps a single replica; use forward_replicas()")
        return self.forward_replicas([{k: v for k, v in b.items() if v is not None}])[0]

    def _entity_maps_on(self, dev: torch.device):
        """(entity_to_shard, entity_to_idx, shard_counts) as int32 device tensors, uploaded once."""
        cache = self.__dict__.setdefault("_entity_map_cache", {})
        if dev not in cache:
            cache[dev] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=torch.int32)
                               for a in (self.sharding.entity_to_shard, self.sharding.entity_to_idx,
                                         self.sharding.shard_counts))
        return cache[dev]

    def set_rank_candidates(self, candidate_ents: Optional[Any]) -> None:
        """Restrict the rank-counting mode to these entities (global ids; None: all) - `candidate_ents` of
        `AllScoresPipeline` (pipeline.py:101-111): every other entity counts as scoring -inf."""
        self.__dict__.pop("_rank_cand_cache", None)
        self.rank_candidates = None if candidate_ents is None else np.unique(np.asarray(candidate_ents).reshape(-1))

    def _rank_candidates_on(self, shard: int, dev: torch.device):
        """(local rows of the shard's candidate entities, ascending; local row -> position in that list or -1)."""
        cache = self.__dict__.setdefault("_rank_cand_cache", {})
        if (shard, dev) not in cache:
            sh = self.sharding
            ents = self.rank_candidates
            rows = np.sort(sh.entity_to_idx[ents[sh.entity_to_shard[ents] == shard]]).astype(np.int32)
            pos = np.full(sh.max_entity_per_shard, -1, dtype=np.int32)
            pos[rows] = np.arange(len(rows), dtype=np.int32)
            cache[(shard, dev)] = (torch.from_numpy(rows).to(dev), torch.from_numpy(pos).to(dev))
        return cache[(shard, dev)]

    def rank_counts_replicas(self, batches: List[_Batch]) -> List[Dict[str, torch.Tensor]]:
        """Rank-counting mode (`rank_truth` in the batch): for every query of the hosted replicas the number of
        entities (all shards) scoring above / exactly as its true completion, without a score matrix - the
        counting is the epilogue of the scoring kernel (`bess_neg_score_shared_fwd_counts`).  What the reference
        gets from the window loop + `Evaluation.ranks_from_scores` (`bess.py:1005-1062`, `metric.py:129-182`,
        `pipeline.py:233-271`), for the case where only ranks / metrics are wanted.

        Inputs next to the query's known entity and relation: `rank_truth` [1, shard_bs] global id of the true
        completion; optional `rank_filter` [1, P, 2] int32 pairs (query position in the replica's micro-batch,
        global id of an entity to leave out; -1 padding; pairs distinct, none naming the query's true
        completion).  Returns per replica `counts` [shard_bs, 2] int32 (above, equal: over the entities that are
        neither the true completion nor filtered; every score involved - candidates, positives, filtered
        completions - in the arithmetic of the all-entity kernel) and `pos_score` [shard_bs] f32 (NaN -> -inf, infinities ->
        the largest finite values, as `Evaluation.ranks_from_scores` does) and `out_of_range` [shard_bs] bool: some
        shard's matrix-core product met operands outside the fp16 range (its counts are then -1: use the score
        matrix for this batch)."""
        group = self._group()
        n = group.n_shard
        fn = self.score_fn
        desc = fn.kernel_desc()
        queries = self._gather_queries(batches)
        devs = [self._local_table(s).device for s in group.local_shards]
        truth_all = group.all_gather([_i32(b["rank_truth"].squeeze(0).to(d)).reshape(-1) for b, d in zip(batches, devs)])
        with_filter = batches[0].get("rank_filter") is not None
        if with_filter:
            filt_all = group.all_gather([_i32(b["rank_filter"].squeeze(0).to(d)) for b, d in zip(batches, devs)])
        # the positive scores: computed where the true completion lives, summed over the shards
        # (single (query, entity) scores - the positives, the filtered completions - in the arithmetic of the
        # all-entity pass they are compared with: bess_neg_score_shared_fwd_pairs)
        subset = getattr(self, "rank_candidates", None) is not None
        pos_parts, local = [], []
        for shard, q, tr in zip(group.local_shards, queries, truth_all):
            table = self._local_table(shard)
            e2s, e2i, _ = self._entity_maps_on(table.device)
            t = tr.reshape(-1).long()
            here = e2s[t] == shard
            row = torch.where(here, e2i[t], torch.zeros_like(e2i[t]))
            # what the all-entity pass of this shard runs over: its entities, or its candidate entities
            cand_rows = cand_pos = None
            n_count = int(self.sharding.shard_counts[shard])  # (the padding rows of a shard are no entities)
            if subset:
                cand_rows, cand_pos = self._rank_candidates_on(shard, table.device)
                n_count = int(cand_rows.numel())
                here = here & (cand_pos[row.long()] >= 0)  # a true completion outside the candidates scores -inf
            sc = nat.neg_score_shared_pairs(desc, q, RowSource(table, row.contiguous()), int(q.shape[0]), max(1, n_count))
            # (second row: is the true completion a candidate at all - on whichever shard it lives)
            pos_parts.append(torch.stack([torch.where(here, sc, torch.zeros_like(sc)), here.to(sc.dtype)]))
            local.append((here, row, cand_rows, cand_pos, n_count))
        thr_all = group.all_reduce_sum(pos_parts)
        outs_c, thrs = [], []
        filters = filt_all if with_filter else [None] * len(queries)
        for shard, q, thr2, (here, row, cand_rows, cand_pos, n_count), f in zip(group.local_shards, queries, thr_all,
                                                                                local, filters):
            table = self._local_table(shard)
            e2s, e2i, cnt = self._entity_maps_on(table.device)
            thr, is_cand = thr2[0], thr2[1] > 0
            # scores leave AllScoresBESS in the model's dtype (bess.py:1058-1062): a half-precision model ranks
            # fp16 scores, ties included
            half = fn.relation_embedding.dtype == torch.float16
            if half:
                thr = thr.half().float()
            thr = torch.nan_to_num(thr, nan=-torch.inf)  # (ranks_from_scores: metric.py:152)
            # (a true completion outside the candidates: -inf there, which the same nan_to_num_ turns into the
            # lowest finite float)
            thr = torch.where(is_cand, thr, torch.full_like(thr, torch.finfo(torch.float32).min)).contiguous()
            if cand_rows is None:
                excl = torch.where(here, row, torch.full_like(row, -1)).contiguous()
                src = RowSource(table[:n_count])
            else:
                excl = torch.where(here, cand_pos[row.long()], torch.full_like(row, -1)).contiguous()
                src = RowSource(table, cand_rows)
            if n_count:
                counts = nat.neg_score_shared_counts(desc, q, src, thr, excl, round_f16=half)
            else:
                counts = torch.zeros((q.shape[0], 2), dtype=torch.int32, device=q.device)
            # The split-fp16 product poisons its counts (INT32_MIN) when an operand is outside the fp16 range.
            # The sentinel must not go through arithmetic (INT32_MIN - 1 wraps, n * INT32_MIN is 0 for even n):
            # it becomes a flag column of its own here, the counts of a poisoned pass are zeroed, and the flag is
            # summed over the shards next to them.
            bad = (counts < 0).any(dim=1)
            counts = torch.where(bad[:, None], torch.zeros_like(counts), counts)
            if f is not None:  # [n, P, 2]: the pairs of every replica
                if int(f.shape[1]):
                    qi, ent = f[..., 0].long(), f[..., 1].long()
                    g = (torch.arange(n, device=f.device)[:, None] * (q.shape[0] // n) + qi.clamp(min=0)).reshape(-1)
                    ent = ent.reshape(-1)
                    ok = (ent >= 0) & (e2s[ent.clamp(min=0)] == shard)
                    rows_f = torch.where(ok, e2i[ent.clamp(min=0)], torch.zeros_like(e2i[ent.clamp(min=0)]))
                    if cand_pos is not None:
                        ok = ok & (cand_pos[rows_f.long()] >= 0)  # (not a candidate: not counted in the first place)
                    sc = nat.neg_score_shared_pairs(desc, q[g].contiguous(), RowSource(table, rows_f.contiguous()),
                                                    int(q.shape[0]), max(1, n_count))
                    if half:
                        sc = sc.half().float()
                    t = thr[g]
                    ok = ok & ~bad[g]
                    sub = torch.stack([(ok & (sc > t)), (ok & (sc == t))], dim=1).to(torch.int32)
                    counts.index_add_(0, g, -sub)
            outs_c.append(torch.cat([counts, bad[:, None].to(torch.int32)], dim=1).reshape(n, -1, 3))
            thrs.append(thr.reshape(n, -1))
        back = group.all_to_all(outs_c)  # counts of every shard's entities, back to the query's replica
        res = []
        for shard, c, thr in zip(group.local_shards, back, thrs):
            total = c.sum(dim=0, dtype=torch.int32)
            flagged = total[:, 2] > 0
            # (counts of a flagged query mean nothing: -1, so that no caller can take them for ranks)
            res.append(dict(counts=torch.where(flagged[:, None], torch.full_like(total[:, :2], -1), total[:, :2]),
                            out_of_range=flagged, pos_score=thr[shard].contiguous()))
        return res

    def forward_replicas(self, batches: List[_Batch]) -> List[torch.Tensor]:
        if batches and batches[0].get("rank_truth") is not None:
            return self.rank_counts_replicas(batches)  # type: ignore[return-value]
        group = self._group()
        n = group.n_shard
        desc = self.score_fn.kernel_desc()
        ws = self.window_size
        queries = self._gather_queries(batches)
        outs = []
        for shard, b, q in zip(group.local_shards, batches, queries):
            table = self._local_table(shard)
            M = int(table.shape[0])
            step = int(b["step"].reshape(-1)[0])
            # rows step*ws .. step*ws+ws-1, clamped to the last row (bess.py:1032-1036)
            rows = torch.clamp(step * ws + torch.arange(ws, dtype=torch.int32, device=table.device), max=M - 1)
            sc = nat.neg_score_shared_fwd(desc, q, RowSource(table, rows.contiguous()))
            outs.append(sc.reshape(n, -1, ws))
        back = group.all_to_all(outs)  # C7
        dt = self.score_fn.relation_embedding.dtype
        return [x.transpose(0, 1).flatten(start_dim=1).contiguous().to(dt) for x in back]
