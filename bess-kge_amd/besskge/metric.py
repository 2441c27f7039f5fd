"""Link-prediction metrics (MRR, Hits@K) with the ranks computed on the device.

Mirror of the reference interface (`besskge/metric.py:14-273`): `Evaluation`
with `ranks_from_scores`, `ranks_from_indices`, `dict_metrics_from_ranks`,
`stacked_metrics_from_ranks`, same modes ("optimistic" | "pessimistic" |
"average"), `worst_rank_infty`, `reduction` and `return_ranks`.  The two
rank computations are HIP kernels (`bess_ranks_from_scores`: one wavefront per
score row counts `>` and `>=` in one pass; `bess_ranks_from_indices`); the
per-rank metric formulas are element-wise on `[batch]` tensors.
"""

import re
from abc import ABC, abstractmethod
from typing import Callable, Dict, List, Optional

import torch

from besskge import _native as nat


class BaseMetric(ABC):
    """A function of the prediction rank."""

    @abstractmethod
    def __call__(self, prediction_rank: torch.Tensor) -> torch.Tensor:
        """[batch] ranks -> [batch] metric values."""
        raise NotImplementedError


class ReciprocalRank(BaseMetric):
    """1 / rank (its mean is the MRR)."""

    def __call__(self, prediction_rank: torch.Tensor) -> torch.Tensor:
        return torch.reciprocal(prediction_rank)


class HitsAtK(BaseMetric):
    """1 if the ground truth is among the K best predictions, else 0."""

    def __init__(self, k: int) -> None:
        self.K = k

    def __call__(self, prediction_rank: torch.Tensor) -> torch.Tensor:
        return (prediction_rank <= self.K).to(torch.float)


#: metric name -> class
METRICS_DICT = {"mrr": ReciprocalRank, "hits@k": HitsAtK}

_MODES = {"optimistic": 0, "pessimistic": 1, "average": 2}


class Evaluation:
    """Computes prediction ranks and link-prediction metrics."""

    def __init__(
        self,
        metric_list: List[str],
        mode: str = "average",
        worst_rank_infty: bool = False,
        reduction: str = "none",
        return_ranks: bool = False,
    ) -> None:
        """
        :param metric_list: "mrr" and / or "hits@K" entries.
        :param mode: tie handling: "optimistic", "pessimistic" or "average".
        :param worst_rank_infty: worst rank is +inf instead of n_candidate + 1.
        :param reduction: "none" or "sum" over the batch.
        :param return_ranks: also return the ranks next to the metrics.
        """
        if mode not in _MODES:
            raise ValueError(f"Mode {mode} not supported for evaluation")
        if reduction not in ("none", "sum"):
            raise ValueError(f"Reduction {reduction} not supported for evaluation")
        self.mode = mode
        self.return_ranks = return_ranks
        self.worst_rank_infty = worst_rank_infty
        self.reduction: Callable[[torch.Tensor], torch.Tensor] = (
            (lambda x: x) if reduction == "none" else (lambda x: torch.sum(x, dim=0))
        )
        self.metrics: Dict[str, Callable[[torch.Tensor], torch.Tensor]] = {}
        for name in metric_list:  # hits@K first, in the order given (reference ordering)
            m = re.search(r"hits@(\d+)", name)
            if m:
                self.metrics[m[0]] = HitsAtK(k=int(m[1]))
        for name in list(set(metric_list) - set(self.metrics.keys())):
            self.metrics[name] = METRICS_DICT[name]()

    def ranks_from_scores(self, pos_score: torch.Tensor, candidate_score: torch.Tensor) -> torch.Tensor:
        """Rank of each positive among its candidates' scores.

        :param pos_score: [batch]; NaN counts as -inf (and is replaced in place,
            as the reference does, `metric.py:152`).
        :param candidate_score: [batch, n_candidate].
        """
        if pos_score.reshape(-1).shape[0] != candidate_score.shape[0]:
            raise ValueError("`pos_score` and `candidate_score` need to have same size at dimension 0")
        pos_score.nan_to_num_(-torch.inf)
        return nat.ranks_from_scores(pos_score, candidate_score, _MODES[self.mode], self.worst_rank_infty)

    def ranks_from_indices(self, ground_truth: torch.Tensor, candidate_indices: torch.Tensor) -> torch.Tensor:
        """Rank of the ground truth in an ORDERED list of distinct candidate ids.

        :param ground_truth: [batch].
        :param candidate_indices: [batch, n_candidate], most likely first.
        """
        return nat.ranks_from_indices(ground_truth, candidate_indices, self.worst_rank_infty)

    def dict_metrics_from_ranks(
        self, batch_rank: torch.Tensor, triple_mask: Optional[torch.Tensor] = None
    ) -> Dict[str, torch.Tensor]:
        """{metric: (reduced) values}; entries where `~triple_mask` count as 0."""
        out = {}
        for name, fn in self.metrics.items():
            v = fn(batch_rank)
            if triple_mask is not None:
                v = torch.where(triple_mask.to(v.device), v, torch.zeros((), dtype=v.dtype, device=v.device))
            out[name] = self.reduction(v)
        return out

    def stacked_metrics_from_ranks(
        self, batch_rank: torch.Tensor, triple_mask: Optional[torch.Tensor] = None
    ) -> torch.Tensor:
        """Metrics stacked in the order of `self.metrics`: (1, n_metrics[, batch])."""
        return torch.stack(list(self.dict_metrics_from_ranks(batch_rank, triple_mask).values())).unsqueeze(0)
