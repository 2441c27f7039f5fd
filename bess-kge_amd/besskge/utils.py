"""Complex-arithmetic helpers of the scoring functions, on the HIP query kernel.

Reference `besskge/utils.py:72-112`.  Inside the hot path these transforms are
fused into `bess_query_fwd` / `bess_score_triple_fwd` (K6); the two functions
below expose the same kernels for callers that used the reference helpers
directly.  Rows are `[re(e) | im(e)]`; results are fp32.

`gather_indices` / `get_entity_filter` (reference `utils.py:10-69`) are index
plumbing of the all-scores pipeline; they are plain torch indexing on whatever
device their inputs live on.
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


def gather_indices(x: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """Row-wise `take_along_dim` for 2-D tensors.

    :param x: (a, e)
    :param index: (b, k) column indices.
    :return: (max(a, b), k): row i of `index` gathered from row i of `x`
        (a == 1 or b == 1 broadcast, otherwise a == b).
    """
    if x.shape[0] != 1 and index.shape[0] != 1 and x.shape[0] != index.shape[0]:
        raise ValueError("gather_indices: incompatible leading dimensions")
    rows = max(x.shape[0], index.shape[0])
    return torch.take_along_dim(x.expand(rows, -1), index.expand(rows, -1).long(), dim=1)


def get_entity_filter(triples: torch.Tensor, filter_triples: torch.Tensor, filter_mode: str) -> torch.Tensor:
    """For every triple (h, r, t) of `triples`, the entities e such that (e, r, t)
    (`filter_mode` "h") or (h, r, e) ("t") is in `filter_triples`.

    :return: (z, 2) rows (i, e): entity e is to be filtered for triple i.
    """
    if filter_mode == "t":
        keep, other = 0, 2
    elif filter_mode == "h":
        keep, other = 2, 0
    else:
        raise ValueError("`filter_mode` needs to be either 'h' or 't'")
    same = (filter_triples[:, 1] == triples[:, 1].view(-1, 1)) & (filter_triples[:, keep] == triples[:, keep].view(-1, 1))
    hits = same.nonzero(as_tuple=False)
    hits[:, 1] = filter_triples[:, other][hits[:, 1]]
    return hits
