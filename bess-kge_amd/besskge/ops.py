"""Differentiable wrappers around the HIP kernels (torch.autograd plumbing).

These make the public scoring API (`BaseScoreFunction.score_triple /
score_heads / score_tails`, loss modules) behave like the reference's torch
expressions - callable on embedding tensors, differentiable - while every
number is produced by a kernel of `libbesskge_hip.so`.

The fused training step of `besskge.bess` does not go through autograd at all
(it calls forward and backward kernels explicitly and never builds a dense
gradient); these Functions are the drop-in API surface.
"""

from typing import Any, Optional, Tuple

import torch

from besskge import _native as nat
from besskge._native import RowSource


def _as_rows(x: torch.Tensor, width: int) -> torch.Tensor:
    x = x.reshape(-1, width)
    return x if x.is_contiguous() else x.contiguous()


def _idx32(idx: torch.Tensor) -> torch.Tensor:
    idx = idx.reshape(-1)
    if idx.dtype != torch.int32:
        idx = idx.to(torch.int32)
    return idx if idx.is_contiguous() else idx.contiguous()


class GatherRows(torch.autograd.Function):
    """K1 `table[idx]`; backward = K9 scatter-add into a dense fp32 gradient."""

    @staticmethod
    def forward(ctx: Any, table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:  # type: ignore
        ctx.save_for_backward(idx)
        ctx.table_shape = table.shape
        ctx.table_dtype = table.dtype
        out = nat.gather_rows(table, _idx32(idx))
        return out.reshape(*idx.shape, table.shape[1])

    @staticmethod
    def backward(ctx: Any, g: torch.Tensor) -> Tuple[Optional[torch.Tensor], None]:  # type: ignore
        (idx,) = ctx.saved_tensors
        W = ctx.table_shape[1]
        dense = torch.zeros(ctx.table_shape, dtype=torch.float32, device=g.device)
        nat.scatter_add_rows(dense, _idx32(idx), _as_rows(g.float(), W))
        return dense.to(ctx.table_dtype), None


class ScoreTriple(torch.autograd.Function):
    """K2+K3 positive score of (h, r, t) rows."""

    @staticmethod
    def forward(ctx: Any, desc: nat.ModelDesc, head: torch.Tensor, rel_table: torch.Tensor,  # type: ignore
                rel_idx: torch.Tensor, tail: torch.Tensor) -> torch.Tensor:
        W = desc.width
        h, t, r = _as_rows(head, W), _as_rows(tail, W), _idx32(rel_idx)
        ctx.desc = desc
        ctx.save_for_backward(h, t, rel_table, r)
        ctx.shapes = (head.shape, tail.shape)
        return nat.score_triple_fwd(desc, RowSource(h), RowSource(t), rel_table, r)

    @staticmethod
    def backward(ctx: Any, g: torch.Tensor):  # type: ignore
        h, t, rel_table, r = ctx.saved_tensors
        drel = torch.zeros(rel_table.shape, dtype=torch.float32, device=g.device)
        dh, dt = nat.score_triple_bwd(ctx.desc, RowSource(h), RowSource(t), rel_table, r,
                                      g.float().contiguous(), drel)
        return (None, dh.reshape(ctx.shapes[0]).to(h.dtype), drel.to(rel_table.dtype), None,
                dt.reshape(ctx.shapes[1]).to(t.dtype))


class ScoreNegatives(torch.autograd.Function):
    """K6 + K4/K5: scores of a set of corrupting entities against fixed queries.

    `ent` [S, W] is the entity kept in the triple (tail for corrupted heads,
    head for corrupted tails); `neg` is [B, N, W].  With `sharing` every query
    is scored against all B*N rows (-> [S, B*N]); otherwise B must be S (or 1,
    which broadcasts) and query s is scored against neg[s] (-> [S, N]).
    """

    @staticmethod
    def forward(ctx: Any, desc: nat.ModelDesc, side: int, sharing: bool, ent: torch.Tensor,  # type: ignore
                rel_table: torch.Tensor, rel_idx: torch.Tensor, neg: torch.Tensor) -> torch.Tensor:
        W = desc.width
        x, r = _as_rows(ent, W), _idx32(rel_idx)
        if neg.dim() != 3:
            raise ValueError("negative embeddings must be [B, n_negative, W]")
        B, N = int(neg.shape[0]), int(neg.shape[1])
        S = x.shape[0]
        shared = sharing or B == 1
        if not shared and B != S:
            raise ValueError(f"per-triple negatives need B == batch size ({B} != {S})")
        rows = _as_rows(neg, W)
        q = nat.query_fwd(desc, side, RowSource(x), rel_table, r)
        if shared:
            out = nat.neg_score_shared_fwd(desc, q, RowSource(rows))
        else:
            out = nat.neg_score_pertriple_fwd(desc, q, RowSource(rows), N)
        ctx.desc, ctx.side, ctx.shared, ctx.N = desc, side, shared, N
        ctx.shapes = (ent.shape, neg.shape)
        ctx.save_for_backward(x, rel_table, r, rows, q, out)
        return out

    @staticmethod
    def backward(ctx: Any, g: torch.Tensor):  # type: ignore
        x, rel_table, r, rows, q, out = ctx.saved_tensors
        g = g.float().contiguous()
        if ctx.shared:
            dq, dn = nat.neg_score_shared_bwd(ctx.desc, q, RowSource(rows), out, g)
        else:
            dq, dn = nat.neg_score_pertriple_bwd(ctx.desc, q, RowSource(rows), ctx.N, g)
        drel = torch.zeros(rel_table.shape, dtype=torch.float32, device=g.device)
        dx = nat.query_bwd(ctx.desc, ctx.side, RowSource(x), rel_table, r, dq, drel)
        return (None, None, None, dx.reshape(ctx.shapes[0]).to(x.dtype), drel.to(rel_table.dtype), None,
                dn.reshape(ctx.shapes[1]).to(rows.dtype))


class ReduceNegatives(torch.autograd.Function):
    """K4/K5 alone: scores of candidate rows against a ready-made query matrix
    (the scorers whose query transform is written with torch ops)."""

    @staticmethod
    def forward(ctx: Any, desc: nat.ModelDesc, sharing: bool, query: torch.Tensor, neg: torch.Tensor  # type: ignore
                ) -> torch.Tensor:
        if neg.dim() != 3:
            raise ValueError("negative embeddings must be [B, n_negative, W]")
        B, N = int(neg.shape[0]), int(neg.shape[1])
        S = int(query.shape[0])
        shared = sharing or B == 1
        if not shared and B != S:
            raise ValueError(f"per-triple negatives need B == batch size ({B} != {S})")
        rows = _as_rows(neg, desc.width)
        q = query.float().contiguous()
        out = nat.neg_score_shared_fwd(desc, q, RowSource(rows)) if shared else \
            nat.neg_score_pertriple_fwd(desc, q, RowSource(rows), N)
        ctx.desc, ctx.shared, ctx.N, ctx.neg_shape = desc, shared, N, neg.shape
        ctx.save_for_backward(q, rows, out)
        return out

    @staticmethod
    def backward(ctx: Any, g: torch.Tensor):  # type: ignore
        q, rows, out = ctx.saved_tensors
        g = g.float().contiguous()
        if ctx.shared:
            dq, dn = nat.neg_score_shared_bwd(ctx.desc, q, RowSource(rows), out, g)
        else:
            dq, dn = nat.neg_score_pertriple_bwd(ctx.desc, q, RowSource(rows), ctx.N, g)
        return None, None, dq, dn.reshape(ctx.neg_shape).to(rows.dtype)


class Loss(torch.autograd.Function):
    """K8 fused loss + score gradients."""

    @staticmethod
    def forward(ctx: Any, ldesc: nat.LossDesc, pos: torch.Tensor, neg: torch.Tensor,  # type: ignore
                weight: torch.Tensor) -> torch.Tensor:
        need = pos.requires_grad or neg.requires_grad
        loss, dp, dn = nat.loss_fwd_bwd(ldesc, pos.contiguous(), neg.contiguous(),
                                        weight.reshape(-1).float().contiguous(), True)
        ctx.save_for_backward(dp, dn)
        del need
        return loss

    @staticmethod
    def backward(ctx: Any, g: torch.Tensor):  # type: ignore
        dp, dn = ctx.saved_tensors
        return None, g * dp, g * dn, None
