"""Complex-arithmetic helpers of the scoring functions, on the HIP query kernel.

Reference `besskge/utils.py:72-112`.  Inside the hot path these transforms are
fused into `bess_query_fwd` / `bess_score_triple_fwd` (K6); the two functions
below expose the same kernels for callers that used the reference helpers
directly.  Rows are `[re(e) | im(e)]`; results are fp32.

`gather_indices` / `get_entity_filter` of the reference belong to the top-k /
all-scores inference variants and are not part of
this package yet.
"""

import torch

from besskge import _native as nat
from besskge._native import RowSource


def _rowwise(scorer: int, v: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    if v.dim() != 2 or r.dim() != 2 or v.shape[0] != r.shape[0]:
        raise ValueError("expected two [a, *] tensors with the same number of rows")
    r = r.to(v.dtype).contiguous()
    v = v.contiguous()
    desc = nat.make_desc(scorer, 1, v, int(r.shape[1]))
    rid = torch.arange(v.shape[0], dtype=torch.int32, device=v.device)
    # query transform of "corrupt the tail": v (x) r, r used as a one-row-per-query table
    return nat.query_fwd(desc, nat.CORRUPT_TAIL, RowSource(v), r, rid)


def complex_multiplication(v1: torch.Tensor, v2: torch.Tensor) -> torch.Tensor:
    """Row-wise complex product of two [a, 2e] tensors."""
    return _rowwise(nat.COMPLEX, v1, v2)


def complex_rotation(v: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """Rotate row k of `v` [a, 2e] by the phases `r[k]` [a, e]: v * exp(i r)."""
    return _rowwise(nat.ROTATE, v, r)
