"""Loss functions of the BESS step, fused loss + score-gradient HIP kernel (K8).

Mirror of the reference interface (`besskge/loss.py:14-251`): same class names,
constructor arguments and `forward(positive_score, negative_score,
triple_weight)`; always fp32, *summed* over the micro-batch.

    LogSigmoidLoss       -1/2 sum_i w_i [ logsig(pos_i + m) + sum_j a_ij logsig(-neg_ij - m) ]
    MarginRankingLoss    sum_i w_i sum_j a_ij relu(neg_ij - pos_i + m)
    SampledSoftmaxCrossEntropyLoss
                         sum_i w_i CE([pos_i, neg_i + log(nE-1) - log N], class 0)

with a_ij = softmax_j(alpha * neg_ij) (detached) for self-adversarial
weighting, else 1/N; everything multiplied by `loss_scale`.
"""

from abc import ABC
from typing import Optional

import numpy as np
import torch

from besskge import _native as nat
from besskge import ops


class BaseLossFunction(torch.nn.Module, ABC):
    """Base class; losses are always computed in fp32."""

    #: use self-adversarial weighting of the negatives
    negative_adversarial_sampling: bool
    #: reciprocal temperature of that weighting
    negative_adversarial_scale: torch.Tensor
    #: loss scaling factor (fp16 tables)
    loss_scale: torch.Tensor

    _kind: int = -1

    def kernel_desc(self, n_negative: int) -> nat.LossDesc:
        """Descriptor handed to `bess_loss_fwd_bwd`."""
        d = nat.LossDesc()
        d.kind = self._kind
        d.adversarial = int(bool(self.negative_adversarial_sampling))
        d.margin = float(getattr(self, "margin", 0.0))
        d.adversarial_scale = float(self.negative_adversarial_scale)
        d.loss_scale = float(self.loss_scale)
        d.ssce_shift = self.score_shift(n_negative)
        return d

    def score_shift(self, n_negative: int) -> float:
        """Constant added to the negative scores before the loss (SSCE only)."""
        return 0.0

    def get_negative_weights(self, negative_score: torch.Tensor) -> torch.Tensor:
        """Weights a_ij of the negatives ([batch, n_negative], or a scalar 1/N).

        Convenience accessor (reference `loss.py:28-51`); the fused kernel
        computes the same weights internally and does not call this.
        """
        if self.negative_adversarial_sampling:
            scale = self.negative_adversarial_scale.to(negative_score.device)
            return torch.softmax(scale * negative_score, dim=-1).detach()
        return torch.tensor(1.0 / negative_score.shape[-1], device=negative_score.device)

    def forward(
        self,
        positive_score: torch.Tensor,
        negative_score: torch.Tensor,
        triple_weight: torch.Tensor,
    ) -> torch.Tensor:
        """Batch loss.

        :param positive_score: [batch].
        :param negative_score: [batch, n_negative].
        :param triple_weight: [batch] or [1] weights of the positive triples.
        """
        return ops.Loss.apply(
            self.kernel_desc(int(negative_score.shape[-1])),
            positive_score.float(),
            negative_score.float(),
            triple_weight,
        )


class MarginBasedLossFunction(BaseLossFunction, ABC):
    """Losses with a margin."""

    def __init__(
        self,
        margin: float,
        negative_adversarial_sampling: bool,
        negative_adversarial_scale: float = 1.0,
        loss_scale: float = 1.0,
    ) -> None:
        super().__init__()
        self.negative_adversarial_sampling = negative_adversarial_sampling
        self.negative_adversarial_scale = torch.tensor(
            negative_adversarial_scale, dtype=torch.float32
        )
        self.loss_scale = torch.tensor(loss_scale, dtype=torch.float32)
        self.margin: torch.Tensor = torch.tensor(margin, dtype=torch.float32)


class LogSigmoidLoss(MarginBasedLossFunction):
    """Log-sigmoid loss (reference loss.py:109-134)."""

    _kind = nat.LOSS_LOGSIGMOID


class MarginRankingLoss(MarginBasedLossFunction):
    """Pairwise hinge loss (reference loss.py:137-195)."""

    _kind = nat.LOSS_MARGIN

    def __init__(
        self,
        margin: float,
        negative_adversarial_sampling: bool,
        negative_adversarial_scale: float = 1.0,
        loss_scale: float = 1.0,
        activation_function: str = "relu",
    ) -> None:
        super().__init__(
            margin, negative_adversarial_sampling, negative_adversarial_scale, loss_scale
        )
        if activation_function != "relu":
            raise ValueError(
                f"Activation function {activation_function} not supported"
                " for MarginRankingLoss"
            )


class SampledSoftmaxCrossEntropyLoss(BaseLossFunction):
    """Sampled-softmax cross entropy (reference loss.py:198-251)."""

    _kind = nat.LOSS_SSCE

    def __init__(self, n_entity: int, loss_scale: float = 1.0) -> None:
        super().__init__()
        self.negative_adversarial_sampling = False
        self.negative_adversarial_scale = torch.tensor(0.0, dtype=torch.float32)
        self.loss_scale = torch.tensor(loss_scale, dtype=torch.float32)
        self.n_entity = n_entity

    def score_shift(self, n_negative: int) -> float:
        # log(1 / E[count(candidate == class)]): constant over the negatives,
        # zero for the target (loss.py:230-237)
        return float(np.log(self.n_entity - 1) - np.log(n_negative))
