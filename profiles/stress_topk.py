#!/usr/bin/env python3
"""Stress of bess_topk_update: random row counts (one wave / four waves per row), widths, list lengths,
padded and unpadded leading dimensions, masks, many tied scores - against a stable sort."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]
import torch
from besskge import _native as nat

dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(2)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for it in range(n_iter):
    rows = int(torch.randint(1, 40, (1,), generator=gen)) if it % 3 else int(torch.randint(6000, 8000, (1,), generator=gen))
    L = int(torch.randint(1, 3000 if rows < 100 else 300, (1,), generator=gen))
    kk = int(torch.randint(1, 129, (1,), generator=gen))  # up to 64: one register per lane, beyond: two
    pad = bool(it % 2)
    bs = torch.full((rows, kk), -50000.0, device=dev)
    bi = torch.full((rows, kk), -1, dtype=torch.int32, device=dev)
    cols = []
    for w in range(2):
        sc = torch.randint(0, 30, (rows, L), generator=gen).float()
        mask = (torch.rand(rows, L, generator=gen) > 0.1) if (it % 5 == 0 and w == 1) else None
        eff = sc if mask is None else sc + (-50000.0) * (~mask).float()
        cols.append(eff)
        if pad:
            buf = torch.zeros(rows, (L + 3) // 4 * 4, device=dev)
            buf[:, :L] = sc.to(dev)
            view = buf[:, :L]
        else:
            view = sc.to(dev)
        nat.topk_update(view, bs, bi, id_base=w * L, mask=None if mask is None else mask.to(dev))
    allc = torch.cat(cols, dim=1).to(dev)
    k_eff = min(kk, 2 * L)
    order = torch.sort(allc, dim=1, descending=True, stable=True).indices[:, :k_eff]
    vals = torch.take_along_dim(allc, order, dim=1)
    ok_v = torch.equal(bs[:, :k_eff], vals)
    live = vals > -40000  # masked entries all tie at the sentinel level: their order among the initial fill is free
    ok_i = torch.equal(bi[:, :k_eff].long()[live], order[live])
    if not (ok_v and ok_i):
        print(f"iteration {it}: rows={rows} L={L} kk={kk} pad={pad}: values {ok_v} ids {ok_i}  FAIL")
        sys.exit(1)
print(f"{n_iter} problems: values and ids equal to a stable sort")
