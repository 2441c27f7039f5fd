"""Import shim for the upstream reference (golden-vector generation ONLY).

This module exists only so `make_golden.py` can run the *reference's own*
Python code (read in place from /root/reference, never copied) on CPU in the
build container and dump input/output vectors into `tests/golden/*.npz`.

It is never imported by the product, by `pytest`, by `bench.py` or by
`__graft_entry__.py`; /root/reference does not exist on the GPU box.

Recipe:
  * a bare `besskge` module object whose `__path__` points at the reference
    package dir, so the reference `__init__` (which dlopens a PopART .so that
    cannot be built here) is skipped;
  * stub `poptorch`, `ogb.linkproppred`, `poptorch_experimental_addons`:
      - `identity_loss(x, reduction)` -> x
      - `distance_matrix(a, b, p)`    -> explicit broadcast p-norm
      - `all_to_all_single_cross_replica` / `all_gather_cross_replica`
        -> thread-barrier simulations over `n` replica threads.
"""

import sys
import threading
import types
from typing import Any, Callable, Dict, List

import torch

REFERENCE_ROOT = "/root/reference"


class ReplicaGroup:
    """Barrier-based collective simulation for `n` replica threads."""

    def __init__(self, n: int) -> None:
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots: List[Any] = [None] * n
        self.local = threading.local()

    def rank(self) -> int:
        return int(self.local.rank)

    def all_to_all(self, x: torch.Tensor) -> torch.Tensor:
        r = self.rank()
        self.slots[r] = x
        self.barrier.wait()
        out = torch.stack([self.slots[j][r] for j in range(self.n)], dim=0)
        self.barrier.wait()
        return out

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        r = self.rank()
        self.slots[r] = x
        self.barrier.wait()
        out = torch.stack([self.slots[j] for j in range(self.n)], dim=0)
        self.barrier.wait()
        return out


_GROUP: Dict[str, Any] = {"g": None}


class _A2A(torch.autograd.Function):
    """all_to_all with the transposed all_to_all as backward."""

    @staticmethod
    def forward(ctx: Any, x: torch.Tensor) -> torch.Tensor:  # type: ignore
        return _GROUP["g"].all_to_all(x.detach())

    @staticmethod
    def backward(ctx: Any, g: torch.Tensor) -> torch.Tensor:  # type: ignore
        return _GROUP["g"].all_to_all(g.contiguous())


class _AG(torch.autograd.Function):
    """all_gather with reduce-scatter (sum) as backward."""

    @staticmethod
    def forward(ctx: Any, x: torch.Tensor) -> torch.Tensor:  # type: ignore
        return _GROUP["g"].all_gather(x.detach())

    @staticmethod
    def backward(ctx: Any, g: torch.Tensor) -> torch.Tensor:  # type: ignore
        grp = _GROUP["g"]
        allg = grp.all_gather(g.contiguous())  # [n_src, n, ...]
        return allg[:, grp.rank()].sum(dim=0)


def _all_to_all(x: torch.Tensor, n: int) -> torch.Tensor:
    if n == 1 or _GROUP["g"] is None:
        assert x.shape[0] == 1
        return x
    return _A2A.apply(x)


def _all_gather(x: torch.Tensor, n: int) -> torch.Tensor:
    if n == 1 or _GROUP["g"] is None:
        return x.unsqueeze(0)
    return _AG.apply(x)


def _distance_matrix(a: torch.Tensor, b: torch.Tensor, p: int) -> torch.Tensor:
    # assumed semantics of pea.distance_matrix (scoring.py:195): ||a_i - b_j||_p
    return torch.norm(a.unsqueeze(1) - b.unsqueeze(0), p=p, dim=-1)


def install() -> types.ModuleType:
    """Install the stubs and return the bare `besskge` package module."""
    if "besskge" in sys.modules and getattr(
        sys.modules["besskge"], "_is_reference_shim", False
    ):
        return sys.modules["besskge"]
    assert "besskge" not in sys.modules, "product besskge already imported"

    poptorch = types.ModuleType("poptorch")

    class Options:  # noqa: D401 - stub
        pass

    class DataLoader:  # noqa: D401 - stub
        pass

    def for_loop(count: int, body: Callable[..., Any], inputs: List[torch.Tensor]) -> List[torch.Tensor]:
        # poptorch.for_loop(n, body, inputs): feed the outputs back n times (bess.py:845)
        state = list(inputs)
        for _ in range(count):
            state = list(body(*state))
        return state

    poptorch.Options = Options  # type: ignore
    poptorch.DataLoader = DataLoader  # type: ignore
    poptorch.identity_loss = lambda x, reduction="none": x  # type: ignore
    poptorch.for_loop = for_loop  # type: ignore
    sys.modules["poptorch"] = poptorch

    ogb = types.ModuleType("ogb")
    ogb_lp = types.ModuleType("ogb.linkproppred")
    ogb.linkproppred = ogb_lp  # type: ignore
    sys.modules["ogb"] = ogb
    sys.modules["ogb.linkproppred"] = ogb_lp

    pea = types.ModuleType("poptorch_experimental_addons")
    pea_coll = types.ModuleType("poptorch_experimental_addons.collectives")
    pea.distance_matrix = _distance_matrix  # type: ignore
    pea_coll.all_to_all_single_cross_replica = _all_to_all  # type: ignore
    pea_coll.all_gather_cross_replica = _all_gather  # type: ignore
    pea.collectives = pea_coll  # type: ignore
    sys.modules["poptorch_experimental_addons"] = pea
    sys.modules["poptorch_experimental_addons.collectives"] = pea_coll

    # loss.py:239-248 passes an int32 class-index target to cross_entropy; that
    # is accepted by PopTorch but rejected by torch-CPU 2.10 ("expected Long").
    # Cast the target; values/semantics are unchanged (class 0 = the positive).
    _ce = torch.nn.functional.cross_entropy

    def _cross_entropy(input: torch.Tensor, target: torch.Tensor, *a: Any, **k: Any) -> torch.Tensor:
        if target.dtype == torch.int32:
            target = target.long()
        return _ce(input, target, *a, **k)

    torch.nn.functional.cross_entropy = _cross_entropy  # type: ignore

    # bess.py:372-391,433-443 call .view() on slices of the all_to_all result
    # (augment_negative); PopTorch views never fail, torch-CPU refuses a view of
    # a non-contiguous split.  Fall back to reshape (same values, same shape).
    _view = torch.Tensor.view

    def _view_or_reshape(self: torch.Tensor, *shape: Any, **kw: Any) -> torch.Tensor:
        try:
            return _view(self, *shape, **kw)
        except RuntimeError:
            return self.reshape(*shape)

    torch.Tensor.view = _view_or_reshape  # type: ignore

    pkg = types.ModuleType("besskge")
    pkg.__path__ = [REFERENCE_ROOT + "/besskge"]  # type: ignore
    pkg._is_reference_shim = True  # type: ignore
    sys.modules["besskge"] = pkg
    sys.dont_write_bytecode = True
    return pkg


def run_replicas(n: int, fn: Callable[[int], Any]) -> List[Any]:
    """Run `fn(rank)` on `n` threads sharing one simulated replica group."""
    if n == 1:
        _GROUP["g"] = None
        return [fn(0)]
    grp = ReplicaGroup(n)
    _GROUP["g"] = grp
    out: List[Any] = [None] * n
    err: List[Any] = [None] * n

    def work(r: int) -> None:
        grp.local.rank = r
        try:
            out[r] = fn(r)
        except BaseException as e:  # pragma: no cover
            err[r] = e
            grp.barrier.abort()

    ths = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    _GROUP["g"] = None
    for e in err:
        if e is not None:
            raise e
    return out
