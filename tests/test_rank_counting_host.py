"""Host side of the rank-counting evaluation (besskge/pipeline.py: `rank_filter_pairs`, `ranks_from_counts`), on the
CPU: from a full score matrix, count what the scoring kernels' epilogues would count (entities above / equal to the
true completion, the true completion and the filtered completions left out) and check that the ranks built from
those counts are the ranks `Evaluation.ranks_from_scores` semantics give on the masked matrix - the reference's
pipeline (pipeline.py:233-271 + metric.py:129-182), NaN / -inf positives and `worst_rank_infty` included."""

import numpy as np
import pytest
import torch

from besskge.pipeline import rank_filter_pairs, ranks_from_counts
from besskge.utils import get_entity_filter


def reference_ranks(pos, cand, mode, worst_inf):
    """`Evaluation.ranks_from_scores` restated (metric.py:129-182)."""
    pos = torch.nan_to_num(pos.clone(), nan=-torch.inf)  # (metric.py:152; infinities -> largest finite values)
    n = cand.shape[1]
    gt = (cand > pos[:, None]).sum(-1).float()
    ge = (cand >= pos[:, None]).sum(-1).float()
    if mode == "optimistic":
        better, worst = gt, gt == n
    elif mode == "pessimistic":
        better, worst = ge, ge == n
    else:
        better, worst = 0.5 * (gt + ge), (gt == n) | (ge == n)
    rank = 1 + better
    return torch.where(worst, torch.full_like(rank, torch.inf), rank) if worst_inf else rank


@pytest.mark.parametrize("mode", ["optimistic", "pessimistic", "average"])
@pytest.mark.parametrize("worst_inf", [False, True])
@pytest.mark.parametrize("scheme", ["t", "h"])
@pytest.mark.parametrize("subset", [False, True])
def test_ranks_from_epilogue_counts_equal_ranks_of_the_masked_matrix(mode, worst_inf, scheme, subset):
    rng = np.random.default_rng(3)
    gen = torch.Generator().manual_seed(3)
    n_entity, n_rel, rows, shard_bs = 200, 5, 6, 8  # rows = micro-batches x shards
    n_slot = rows * shard_bs
    keep = torch.rand(n_slot, generator=gen) > 0.2  # padded triples
    triples = torch.from_numpy(np.stack([rng.integers(n_entity, size=n_slot), rng.integers(n_rel, size=n_slot),
                                         rng.integers(n_entity, size=n_slot)], axis=1))
    truth = triples[:, 2 if scheme == "t" else 0]
    kept = triples[keep]
    # filter set: shares (h, r) / (r, t) with the queries, contains the queries themselves and duplicates
    extra = torch.from_numpy(np.stack([rng.integers(n_entity, size=600), rng.integers(n_rel, size=600),
                                       rng.integers(n_entity, size=600)], axis=1))
    pick = torch.from_numpy(rng.integers(len(kept), size=300))
    if scheme == "t":
        extra[:300, :2] = kept[pick][:, :2]
    else:
        extra[:300, 1:] = kept[pick][:, 1:]
    extra = torch.cat([extra, kept[:10], extra[:50]])
    flt = get_entity_filter(kept, extra, filter_mode=scheme)
    filt, per_kept = rank_filter_pairs(flt, keep, truth, rows, shard_bs)
    # the pairs: distinct, never the truth, addressed by (row, position in the row's micro-batch)
    assert filt.dtype == torch.int32 and filt.shape[0] == rows and filt.shape[2] == 2
    seen = set()
    for r in range(rows):
        for qi, e in filt[r].tolist():
            if qi < 0:
                assert e == -1
                continue
            slot = r * shard_bs + qi
            assert bool(keep[slot]) and e != int(truth[slot]) and (slot, e) not in seen
            seen.add((slot, e))
    kept_slots = keep.nonzero().reshape(-1)
    want_pairs = {(int(kept_slots[i]), int(e)) for i, e in flt.tolist() if int(e) != int(truth[kept_slots[i]])}
    assert seen == want_pairs
    assert torch.equal(per_kept, torch.tensor([sum(1 for s_, _ in want_pairs if s_ == int(k)) for k in kept_slots]))

    # scores with ties, one NaN and one -inf positive
    scores = torch.randint(0, 40, (n_slot, n_entity), generator=gen).float()
    scores[3, int(truth[3])] = float("nan")
    scores[5, int(truth[5])] = -float("inf")
    keep[3] = keep[5] = True  # (those two are looked at whether or not the draw kept them)
    # `candidate_ents`: only a subset of the entities competes (pipeline.py:247-250 sets the others to -inf BEFORE
    # the true scores are read); the NaN positive's truth is a candidate, some truths and filter entities are not
    candidate = None
    n_outside = 0
    if subset:
        candidate = torch.rand(n_entity, generator=gen) > 0.3
        candidate[int(truth[3])] = True
        n_outside = int((~candidate).sum())
        scores[:, ~candidate] = -torch.inf
        scores[3, int(truth[3])] = float("nan")
    # recompute what depends on keep
    kept = triples[keep]
    flt = get_entity_filter(kept, extra, filter_mode=scheme)
    filt, per_kept = rank_filter_pairs(flt, keep, truth, rows, shard_bs, candidate)
    kept_slots = keep.nonzero().reshape(-1)
    pos = scores[kept_slots, truth[kept_slots]].clone()
    masked = scores[kept_slots].clone()
    masked[flt[:, 0], flt[:, 1]] = -torch.inf
    masked[torch.arange(len(kept_slots)), truth[kept_slots]] = -torch.inf
    want = reference_ranks(pos, masked, mode, worst_inf)
    # what the kernels hand back: counts over the entities that are not masked, against the nan_to_num'ed positive
    thr = torch.nan_to_num(pos.clone(), nan=-torch.inf)
    open_ = torch.isfinite(masked) | (masked == torch.inf)
    if subset:
        open_ &= candidate[None, :]  # the counting pass runs over the candidate rows only
    counts = torch.stack([((scores[kept_slots] > thr[:, None]) & open_).sum(-1),
                          ((scores[kept_slots] == thr[:, None]) & open_).sum(-1)], dim=1).to(torch.int32)
    got = ranks_from_counts(counts, thr, 1 + per_kept + n_outside, n_entity, mode, worst_inf)
    assert torch.equal(got, want)
