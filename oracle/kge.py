"""CPU oracle of the BESS device step (TEST INFRASTRUCTURE - not product code).

A plain-torch CPU restatement of what the reference computes on the hot path,
written independently of the product (`bess-kge_amd/`): no HIP, no index maps,
no collectives - the all-to-all is *simulated by indexing* `[:, :, replica]`,
the way the reference's own host tests do (`tests/test_negative_sampler.py:
112-119`, `tests/test_batch_sampler.py:113-123`).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package, and only as the checker.

Parity pinning: every function here is checked against vectors produced by the
reference's own code (tests/golden/{scoring,loss,bess}.npz, generator
tests/golden/make_golden.py) in `tests/test_oracle.py`.

Reference lines restated:
  score_triple / score_heads / score_tails ... scoring.py:321-354 (TransE),
      423-462 (RotatE), 804-837 (DistMult), 905-946 (ComplEx), 540-593 (PairRE),
      681-743 (TripleRE), 1499-1572 (InterHT), 1661-1750 (TranS), 1090-1146 (ConvE), 1250-1415 (BoxE);
      broadcasted_distance scoring.py:176-200, broadcasted_dot_product 231-255;
      complex_multiplication / complex_rotation utils.py:72-112
  losses ............ loss.py:28-51, 115-134, 179-195, 224-251
  embedding_moving .. bess.py:117-278 (forward glue), 322-468 (score_batch)
  score_moving ...... bess.py:490-603
`pea.distance_matrix` (not in /root/reference, pinned git 899aec4) is taken as
the explicit broadcast p-norm ||a_i - b_j||_p.
"""

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

BAD_NEGATIVE_SCORE = -50000.0

TRANSE, ROTATE, DISTMULT, COMPLEX = "TransE", "RotatE", "DistMult", "ComplEx"

# PairRE / TripleRE / InterHT / TranS (scoring.py:465-743, 1418-1750) and the constructor
# options the fixtures were generated with (name -> reference class + options)
AFFINE_VARIANTS = {
    "PairRE": dict(base="PairRE", normalize=True),
    "TripleRE": dict(base="TripleRE", normalize=True, u=0.0),
    "TripleREv2": dict(base="TripleRE", normalize=False, u=0.5),
    "InterHT": dict(base="InterHT", normalize=True, offset=1.0),
    "TranS": dict(base="TranS", normalize=True, offset=1.0),
    "TranSnn": dict(base="TranS", normalize=False, offset=0.5),
}


# BoxE fixtures (scoring.py:1149-1415): name -> (apply_tanh, dist_func_per_dim)
BOXE_VARIANTS = {"BoxE": (True, True), "BoxEnt": (False, False), "BoxEall": (True, False), "BoxEpd": (False, True)}
BOXE_EPS = 1e-6


def entity_width(scorer: str, d: int) -> int:
    if scorer in BOXE_VARIANTS:
        return 2 * d
    if scorer == "ConvE":
        return d + 1
    if scorer in (ROTATE, COMPLEX):
        return 2 * d
    if scorer in AFFINE_VARIANTS and AFFINE_VARIANTS[scorer]["base"] in ("InterHT", "TranS"):
        return 2 * d
    return d


def relation_width(scorer: str, d: int) -> int:
    if scorer in BOXE_VARIANTS:
        return 4 * d + 2
    if scorer == COMPLEX:
        return 2 * d
    if scorer in AFFINE_VARIANTS:
        return {"PairRE": 2 * d, "TripleRE": 3 * d, "InterHT": d, "TranS": 3 * d}[AFFINE_VARIANTS[scorer]["base"]]
    return d


# ------------------------------------------------------------------ scoring --
def _cmul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    e = a.shape[-1] // 2
    ar, ai, br, bi = a[..., :e], a[..., e:], b[..., :e], b[..., e:]
    return torch.cat([ar * br - ai * bi, ar * bi + ai * br], dim=-1)


def _rot(v: torch.Tensor, phase: torch.Tensor) -> torch.Tensor:
    return _cmul(v, torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1))


def _is_distance(scorer: str) -> bool:
    return scorer in (TRANSE, ROTATE)


def _affine_delta(scorer: str, h: torch.Tensor, rel: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """The vector whose p-norm is the negated score; h, rel, t broadcast against each other
    over their leading dims (the reference's `unsqueeze(1)` forms)."""
    cfg = AFFINE_VARIANTS[scorer]
    base = cfg["base"]
    d = rel.shape[-1] // {"PairRE": 2, "TripleRE": 3, "InterHT": 1, "TranS": 3}[base]

    def parts(x: torch.Tensor) -> List[torch.Tensor]:
        ps = list(torch.split(x, d, dim=-1))
        if cfg["normalize"]:
            ps = [p / torch.clamp(torch.linalg.vector_norm(p, dim=-1, keepdim=True), min=1e-12) for p in ps]
        return ps

    hp, tp = parts(h), parts(t)
    if base == "PairRE":
        r_h, r_t = rel[..., :d], rel[..., d:]
        return hp[0] * r_h - tp[0] * r_t
    if base == "TripleRE":
        r_h, r_m, r_t = rel[..., :d], rel[..., d:2 * d], rel[..., 2 * d:]
        if cfg["u"] > 0.0:
            r_h, r_t = r_h + cfg["u"], r_t + cfg["u"]
        return hp[0] * r_h - tp[0] * r_t + r_m
    off = cfg["offset"]
    if base == "InterHT":
        return hp[0] * (tp[1] + off) + rel - tp[0] * (hp[1] + off)
    r, r_bar, r_hat = rel[..., :d], rel[..., d:2 * d], rel[..., 2 * d:]
    return hp[0] * (tp[1] + off + r_bar) - tp[0] * (hp[1] + off - r_hat) + r


def query(scorer: str, side: str, ent: torch.Tensor, rel: torch.Tensor) -> torch.Tensor:
    """Query vector in front of the candidate reduction.

    side "t" (tails are corrupted): built from (h, r); side "h": from (r, t).
    """
    if scorer == TRANSE:
        return ent + rel if side == "t" else ent - rel
    if scorer == ROTATE:
        return _rot(ent, rel if side == "t" else -rel)
    if scorer == DISTMULT:
        return ent * rel
    if scorer == COMPLEX:
        if side == "t":
            return _cmul(ent, rel)
        e = rel.shape[-1] // 2
        conj = torch.cat([rel[..., :e], -rel[..., e:]], dim=-1)
        return _cmul(conj, ent)
    raise ValueError(scorer)


# fp16 mode of the reference (`model.half()`, notebooks/3_wikikg2_fp16.ipynb:300-392): every tensor of the
# scoring path is fp16, so the query `h + r` / `rot(h, r)` that meets the candidates in
# `pea.distance_matrix` (scoring.py:194-197) has been rounded to fp16.  The oracle works on fp32 tensors
# holding fp16-representable table values; inside `half_queries()` it also rounds that query (straight-through
# for the gradient: rounding has derivative 1 almost everywhere), which is what the packed-fp16 L1 kernels of
# the product do for TransE / RotatE with p = 1 and shared negatives.  Everything after the rounding stays
# fp32 here (the reference would go on in fp16 with IPU stochastic rounding: no bit-exact equivalent).
_HALF_QUERY = [False]


class half_queries:
    def __enter__(self) -> None:
        self.old = _HALF_QUERY[0]
        _HALF_QUERY[0] = True

    def __exit__(self, *exc) -> None:
        _HALF_QUERY[0] = self.old


def _reduce(scorer: str, p: int, q: torch.Tensor, cand: torch.Tensor) -> torch.Tensor:
    """q [..., W] against cand [..., W] (broadcast) -> [...]"""
    if _is_distance(scorer):
        return -torch.norm(q - cand, p=p, dim=-1)
    return torch.sum(q * cand, dim=-1)


CONVE = "ConvE"


def boxe_score(scorer: str, p: int, bumped: torch.Tensor, rel: torch.Tensor) -> torch.Tensor:
    """`BoxE.boxe_score` (scoring.py:1250-1340): bumped [..., 2, d] (head-box point, tail-box point),
    rel [..., 4d + 2] broadcastable against it."""
    tanh, per_dim = BOXE_VARIANTS[scorer]
    d = bumped.shape[-1]
    center = rel[..., : 2 * d].reshape(*rel.shape[:-1], 2, d)
    width = rel[..., 2 * d: 4 * d].reshape(*rel.shape[:-1], 2, d).abs()
    size = rel[..., 4 * d:]
    gm = torch.exp(torch.mean(torch.log(torch.clamp(width, min=BOXE_EPS)), dim=-1, keepdim=True))
    width = width / torch.clamp(gm, min=BOXE_EPS)
    width = width * (1.0 + torch.nn.functional.elu(size.unsqueeze(-1)))
    if tanh:
        low = torch.tanh(center - 0.5 * width)
        up = torch.tanh(low + width)
        center = 0.5 * (low + up)
        width = up - low
        dist = torch.abs(torch.tanh(bumped) - center)
    else:
        dist = torch.abs(bumped - center)
    wp1 = 1.0 + width
    k = 0.5 * width * (wp1 - 1.0 / wp1)
    inside = dist <= 0.5 * width
    if not per_dim:
        inside = torch.all(inside, dim=-1, keepdim=True)
    final = torch.where(inside, dist / wp1, dist * wp1 - k)
    return -torch.norm(final, p=p, dim=-1).sum(-1)


def conve_net(net: Dict[str, torch.Tensor], training: bool, h: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """ConvE's query network (scoring.py:1034-1062, 1100-1110) restated with functional ops:
    BN2d -> conv 3x3 -> BN2d -> relu -> flatten -> linear -> BN1d -> relu  (dropout rates 0).
    `net` holds the state dict of the reference's `conv_layers` / `fc_layers`; h [S, d + 1] (bias last)."""
    F = torch.nn.functional
    d = r.shape[-1]
    emb_w = 4
    emb_h = d // emb_w
    x = torch.cat([h[..., :-1].reshape(-1, 1, emb_h, emb_w), r.reshape(-1, 1, emb_h, emb_w)], dim=-2)

    def bn(x: torch.Tensor, pre: str) -> torch.Tensor:
        return F.batch_norm(x, None if training else net[pre + ".running_mean"],
                            None if training else net[pre + ".running_var"], net[pre + ".weight"],
                            net[pre + ".bias"], training=training, eps=1e-5)

    x = bn(x, "conv_layers.0")
    x = F.conv2d(x, net["conv_layers.2.weight"], net["conv_layers.2.bias"])
    x = torch.relu(bn(x, "conv_layers.3"))
    x = F.linear(x.flatten(start_dim=1), net["fc_layers.0.weight"], net["fc_layers.0.bias"])
    return torch.relu(bn(x, "fc_layers.2"))


def score_triple(scorer: str, p: int, h: torch.Tensor, rel_table: torch.Tensor, rid: torch.Tensor,
                 t: torch.Tensor, net: Optional[Dict[str, torch.Tensor]] = None, training: bool = True
                 ) -> torch.Tensor:
    if scorer in BOXE_VARIANTS:
        d = h.shape[-1] // 2
        bumped = h.reshape(-1, 2, d) + t.reshape(-1, 2, d)[:, [1, 0]]
        return boxe_score(scorer, p, bumped, rel_table[rid.long()])
    if scorer == CONVE:
        q = conve_net(net, training, h, rel_table[rid.long()])
        return torch.sum(q * t[..., :-1], dim=-1) + t[..., -1]
    if scorer in AFFINE_VARIANTS:
        return -torch.norm(_affine_delta(scorer, h, rel_table[rid.long()], t), p=p, dim=-1)
    return _reduce(scorer, p, query(scorer, "t", h, rel_table[rid.long()]), t)


def score_candidates(scorer: str, p: int, sharing: bool, side: str, ent: torch.Tensor,
                     rel_table: torch.Tensor, rid: torch.Tensor, cand: torch.Tensor,
                     net: Optional[Dict[str, torch.Tensor]] = None, training: bool = True) -> torch.Tensor:
    """score_heads (side "h", ent = tails) / score_tails (side "t", ent = heads).

    cand [B, N, W].  sharing: every query vs all B*N rows -> [S, B*N];
    else query s vs cand[s] (B == S, or B == 1 broadcast) -> [S, N].
    """
    if scorer in BOXE_VARIANTS:
        d = ent.shape[-1] // 2
        c = cand.reshape(1, -1, cand.shape[-1]) if sharing else cand
        c2, e2 = c.reshape(c.shape[0], -1, 2, d), ent.reshape(-1, 1, 2, d)
        bumped = (c2 + e2[:, :, [1, 0]]) if side == "h" else (e2 + c2[:, :, [1, 0]])
        return boxe_score(scorer, p, bumped, rel_table[rid.long()][:, None, :])
    if scorer == CONVE:
        assert side == "t", "ConvE only corrupts tails"
        q = conve_net(net, training, ent, rel_table[rid.long()])[:, None, :]
        c = cand.reshape(1, -1, cand.shape[-1]) if sharing else cand
        return torch.sum(q * c[..., :-1], dim=-1) + c[..., -1]
    if scorer in AFFINE_VARIANTS:
        c = cand.reshape(1, -1, cand.shape[-1]) if sharing else cand
        e, r = ent[:, None, :], rel_table[rid.long()][:, None, :]
        delta = _affine_delta(scorer, c, r, e) if side == "h" else _affine_delta(scorer, e, r, c)
        return -torch.norm(delta, p=p, dim=-1)
    q = query(scorer, side, ent, rel_table[rid.long()])  # [S, W]
    if sharing:
        if _HALF_QUERY[0] and _is_distance(scorer) and p == 1:
            q = q + (q.half().float() - q).detach()
        flat = cand.reshape(-1, cand.shape[-1])
        return _reduce(scorer, p, q[:, None, :], flat[None, :, :])
    return _reduce(scorer, p, q[:, None, :], cand)


# ------------------------------------------------------------------- losses --
def negative_weights(neg: torch.Tensor, adversarial: bool, scale: float) -> torch.Tensor:
    if adversarial:
        return torch.softmax(scale * neg, dim=-1).detach()
    return torch.full_like(neg, 1.0 / neg.shape[-1])


def loss_value(kind: str, pos: torch.Tensor, neg: torch.Tensor, w: torch.Tensor, margin: float = 0.0,
               adversarial: bool = False, adversarial_scale: float = 1.0, loss_scale: float = 1.0,
               n_entity: int = 0) -> torch.Tensor:
    """kind in {"logsigmoid", "margin", "ssce"}; always fp32 inputs; summed."""
    if kind == "logsigmoid":
        a = negative_weights(neg, adversarial, adversarial_scale)
        per = torch.nn.functional.logsigmoid(pos + margin) + (
            a * torch.nn.functional.logsigmoid(-neg - margin)).sum(-1)
        return loss_scale * (-0.5) * (w * per).sum()
    if kind == "margin":
        a = negative_weights(neg, adversarial, adversarial_scale)
        per = (a * torch.relu(neg - pos[:, None] + margin)).sum(-1)
        return loss_scale * (w * per).sum()
    if kind == "ssce":
        shift = float(np.log(n_entity - 1) - np.log(neg.shape[1]))
        logits = torch.cat([pos[:, None], neg + shift], dim=-1)
        per = torch.logsumexp(logits, dim=-1) - pos
        return loss_scale * (w * per).sum()
    raise ValueError(kind)


# ------------------------------------------------------------ full BESS step --
class StepSpec:
    """Static description of a BESS micro-batch."""

    def __init__(self, scorer: str, p: int, sharing: bool, scheme: str, flat: bool,
                 local_sampling: bool = False, augment: bool = False, triple_based: bool = False,
                 net: Optional[Dict[str, torch.Tensor]] = None) -> None:
        self.scorer, self.p, self.sharing = scorer, p, sharing
        self.net = net  # ConvE: state dict of the query network (evaluated in train mode)
        self.scheme, self.flat = scheme, flat
        self.local_sampling, self.augment, self.triple_based = local_sampling, augment, triple_based


def _rows(table: torch.Tensor, shard: int, idx: torch.Tensor) -> torch.Tensor:
    return table[shard][idx.long()]


def embedding_moving_scores(spec: StepSpec, table: torch.Tensor, rel_table: torch.Tensor, head: torch.Tensor,
                            relation: torch.Tensor, tail: torch.Tensor, negative: torch.Tensor,
                            replica: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Scores of one replica for one micro-batch, EmbeddingMoving scheme.

    Whole-step tensors (what the sampler emits for one step, all shards):
      head, relation [n(replica), n(block), ppp]; tail [n(source), n(replica), ppp];
      negative [n(source), n(replica), B, K]; table [n, M, W].
    """
    n = table.shape[0]
    r = replica
    ppp = head.shape[-1]
    W = table.shape[-1]
    h = _rows(table, r, head[r]).reshape(-1, W)  # [S, W]
    rid = relation[r].reshape(-1)
    # what the all-to-all delivers to replica r: block j from shard j
    t = torch.stack([_rows(table, j, tail[j, r]) for j in range(n)]).reshape(-1, W)
    if spec.local_sampling:
        neg = torch.stack([_rows(table, r, negative[r, j]) for j in range(n)])
    else:
        neg = torch.stack([_rows(table, j, negative[j, r]) for j in range(n)])
    # [n, B, K, W] -> [B, n*K, W]
    neg = neg.transpose(0, 1).reshape(neg.shape[1], -1, W)
    pos = score_triple(spec.scorer, spec.p, h, rel_table, rid, t, net=spec.net)

    def corrupt(side: str, sel: Optional[torch.Tensor], cand: torch.Tensor) -> torch.Tensor:
        ent = t if side == "h" else h
        own = h if side == "h" else t  # positives of the corrupted side
        rr = rid
        if sel is not None:
            ent, own, rr = ent[sel], own[sel], rid[sel]
        if spec.augment:
            cand = torch.cat([own.reshape(cand.shape[0], -1, W), cand], dim=1)
        return score_candidates(spec.scorer, spec.p, spec.sharing, side, ent, rel_table, rr, cand, net=spec.net)

    if spec.scheme in ("h", "t"):
        return pos, corrupt(spec.scheme, None, neg)
    cut = ppp // 2
    slot = torch.arange(n * ppp).reshape(n, ppp)
    sel_h, sel_t = slot[:, :cut].reshape(-1), slot[:, cut:].reshape(-1)
    if spec.flat:
        neg_h, neg_t = neg[0:1], neg[1:2]
    else:
        per = neg.reshape(n, ppp, -1, W)
        neg_h = per[:, :cut].reshape(n * cut, -1, W)
        neg_t = per[:, cut:].reshape(n * (ppp - cut), -1, W)
    sh = corrupt("h", sel_h, neg_h)
    st = corrupt("t", sel_t, neg_t)
    both = torch.cat([sh.reshape(n, cut, -1), st.reshape(n, ppp - cut, -1)], dim=1)
    return pos, both.reshape(n * ppp, -1)


def score_moving_scores(spec: StepSpec, table: torch.Tensor, rel_table: torch.Tensor, head: torch.Tensor,
                        relation: torch.Tensor, tail: torch.Tensor, negative: torch.Tensor,
                        replica: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Scores of one replica, ScoreMoving scheme: for every shard j, the
    replica's queries are scored against the negatives stored on j."""
    n = table.shape[0]
    r = replica
    ppp = head.shape[-1]
    W = table.shape[-1]
    cut = ppp // 2
    h = _rows(table, r, head[r]).reshape(-1, W)
    rid = relation[r].reshape(-1)
    t = torch.stack([_rows(table, j, tail[j, r]) for j in range(n)]).reshape(-1, W)
    pos = score_triple(spec.scorer, spec.p, h, rel_table, rid, t, net=spec.net)
    slot = torch.arange(n * ppp).reshape(n, ppp)
    q_conve = None
    if spec.scorer == CONVE:
        # the shard that scores sees the gathered queries of *all* replicas in one batch
        # (bess.py:519-560): the batch-norm statistics of the query network are taken over all of them
        h_all = torch.cat([_rows(table, rr, head[rr]).reshape(-1, W) for rr in range(n)])
        rid_all = torch.cat([relation[rr].reshape(-1) for rr in range(n)])
        q_conve = conve_net(spec.net, True, h_all, rel_table[rid_all.long()])[r * n * ppp:(r + 1) * n * ppp]
    blocks = []
    for j in range(n):
        loc = negative[j]  # [n(dest), B, K] rows of shard j
        if spec.triple_based and spec.flat:
            loc = loc[0:1]
        if spec.sharing:
            pool = {  # every query sees all the negatives shard j holds for its side
                "h": loc if spec.scheme != "ht" else loc[:, 0:1],
                "t": loc if spec.scheme != "ht" else loc[:, 1:2],
            }
            if spec.scheme == "ht" and not spec.flat:
                per = loc.reshape(loc.shape[0], n, ppp, -1)
                pool = {"h": per[:, :, :cut], "t": per[:, :, cut:]}
        else:
            mine = loc[r]  # [S, K]: the lists of replica r's triples
            per = mine.reshape(n, ppp, -1)
            pool = {"h": mine if spec.scheme != "ht" else per[:, :cut].reshape(n * cut, -1),
                    "t": mine if spec.scheme != "ht" else per[:, cut:].reshape(n * (ppp - cut), -1)}

        def corrupt(side: str, sel: Optional[torch.Tensor]) -> torch.Tensor:
            ent = t if side == "h" else h
            rr = rid
            if sel is not None:
                ent, rr = ent[sel], rid[sel]
            idx = pool[side]
            if spec.sharing:
                cand = _rows(table, j, idx.reshape(-1)).reshape(1, -1, W)
            else:
                cand = _rows(table, j, idx)  # [Sg, K, W]
            if q_conve is not None:
                return torch.sum(q_conve[:, None, :] * cand[..., :-1], dim=-1) + cand[..., -1]
            return score_candidates(spec.scorer, spec.p, spec.sharing, side, ent, rel_table, rr, cand, net=spec.net)

        if spec.scheme in ("h", "t"):
            blocks.append(corrupt(spec.scheme, None))
        else:
            sh = corrupt("h", slot[:, :cut].reshape(-1))
            st = corrupt("t", slot[:, cut:].reshape(-1))
            blocks.append(torch.cat([sh.reshape(n, cut, -1), st.reshape(n, ppp - cut, -1)], dim=1).reshape(n * ppp, -1))
    return pos, torch.cat(blocks, dim=1)


def apply_masks(spec: StepSpec, neg_score: torch.Tensor, n: int, ppp: int, K: int,
                negative_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """Padding / augmentation kill (bess.py:182-245).  negative_mask [B', n, L] for this replica."""
    S, N = neg_score.shape
    cut = ppp // 2
    kill = torch.zeros(S, N, dtype=torch.bool)
    mask2d = None
    if negative_mask is not None:
        mask2d = negative_mask.reshape(negative_mask.shape[0], -1)
        if spec.flat and spec.scheme == "ht":
            mh = mask2d[0:1].expand(n * cut, -1).reshape(n, cut, -1)
            mt = mask2d[1:2].expand(n * (ppp - cut), -1).reshape(n, ppp - cut, -1)
            mask2d = torch.cat([mh, mt], dim=1).reshape(S, -1)
        mask2d = mask2d.expand(S, -1)
    if spec.augment:
        step = 1 if spec.flat else 1 + n * K
        qpos = torch.arange(S)
        if spec.scheme == "ht":
            blk, p_ = qpos // ppp, qpos % ppp
            qpos = blk * cut + p_ % cut
        cols = step * qpos
        ok = cols < N
        kill[torch.arange(S)[ok], cols[ok]] = True
        if mask2d is not None:
            kill[:, N - mask2d.shape[1]:] = ~mask2d
    elif mask2d is not None:
        kill[:, N - mask2d.shape[1]:] = ~mask2d
    return neg_score + BAD_NEGATIVE_SCORE * kill.to(neg_score.dtype)


def bess_step(spec: StepSpec, scheme_cls: str, table: torch.Tensor, rel_table: torch.Tensor,
              batch: Dict[str, torch.Tensor], loss: Optional[dict] = None
              ) -> Dict[str, List[torch.Tensor]]:
    """All replicas of one micro-batch.  `batch` holds whole-step tensors
    (head/relation/tail [n, n, ppp], negative [n, n, B, K], optional
    negative_mask [n(replica), B', n, L], triple_weight [n, S]).  Returns lists
    indexed by replica: positive_score, negative_score, loss."""
    n = table.shape[0]
    ppp = batch["head"].shape[-1]
    K = batch["negative"].shape[-1]
    fn = embedding_moving_scores if scheme_cls == "EmbeddingMoving" else score_moving_scores
    out: Dict[str, List[torch.Tensor]] = dict(positive_score=[], negative_score=[], loss=[])
    for r in range(n):
        pos, neg = fn(spec, table, rel_table, batch["head"], batch["relation"], batch["tail"],
                      batch["negative"], r)
        nm = batch["negative_mask"][r] if "negative_mask" in batch else None
        neg = apply_masks(spec, neg, n, ppp, K, nm)
        out["positive_score"].append(pos)
        out["negative_score"].append(neg)
        if loss is not None:
            w = batch["triple_weight"][r] if "triple_weight" in batch else torch.ones(1)
            out["loss"].append(loss_value(pos=pos.float(), neg=neg.float(), w=w, **loss))
    return out
