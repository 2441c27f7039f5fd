"""Replica groups: who holds which shard, and the two BESS collectives.

The reference runs one PopTorch replica per shard and uses exactly two
collectives from poptorch_experimental_addons (reference `bess.py:14-19`):
`all_to_all_single_cross_replica` (equal splits along dim 0: block j goes to
replica j, result block j came from replica j) and `all_gather_cross_replica`
(stack in rank order).  Here:

* :class:`DistributedGroup` - the production layout: one process per GPU,
  rank == shard, `torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo"
  for CPU-side tests of the routing).
* :class:`SingleProcessGroup` - all `n_shard` replicas live in this process on
  one device and are stepped in lock-step (the role PopTorch's IPUModel plays
  in the reference's tests, `tests/test_bess.py:123-127`); the collectives are
  device-side block transposes.

* :class:`NativeGroup` - the same layout with the collectives issued through the
  library's own RCCL entry points (`bess_alltoall`, `bess_allgather`,
  `bess_allreduce_sum_f32`, `bess_pack_exchange`; include/besskge_hip.h) on the
  stream the kernels run on: a step is one in-order queue, nothing waits on
  c10d's separate collective stream, and the step can be hipGraph-captured.

All operate on *lists* with one tensor per local replica, so the model code is
written once.  The entity-shard gradient is never all-reduced (the reference
removes that all-reduce with a PopART pattern,
`custom_ops/remove_all_reduce_pattern.cpp:15-47`); only replicated parameters
(relation table) go through :meth:`all_reduce_sum`.
"""

from abc import ABC, abstractmethod
import sys
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist


class ReplicaGroup(ABC):
    """Set of BESS replicas hosted by this process."""

    #: total number of shards / replicas
    n_shard: int
    #: shard ids hosted here, in execution order
    local_shards: List[int]

    @abstractmethod
    def all_to_all(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        """xs[r] is [n_shard, ...]; out[r][j] = (replica j's x)[shard of r]."""

    @abstractmethod
    def all_gather(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        """out[r] = stack over all replicas j of (replica j's x): [n_shard, ...]."""

    @abstractmethod
    def all_reduce_sum(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        """Sum over all replicas (replicated parameters' gradients)."""

    def barrier(self) -> None:
        """Synchronise all replicas (no-op when they share a process)."""

    def all_agree(self, ok: bool) -> bool:
        """True iff `ok` on EVERY rank - for decisions that change which collectives a rank issues next (keep a
        recorded plan or fall back): a rank deciding alone leaves its peers waiting in an exchange it never joins.
        One process hosting every replica decides alone."""
        return bool(ok)

    # ------------------------------------------------------------------ recorded steps
    # A hipGraph that recorded a collective keeps the communicator busy: `ncclCommDestroy` waits for it
    # (round 3: a process that ended with a `Runner(use_graphs=True)` over a `NativeGroup` alive hung in
    # `Communicator.__del__`).  The group therefore knows every graph cache recorded over it and destroys the
    # graphs BEFORE the communicator goes.
    def register_graph_cache(self, cache: Dict[Any, Any]) -> None:
        """`cache` maps an input signature to `(graph, static inputs, static outputs)` (`Runner._graphs`)."""
        held = self.__dict__.setdefault("_graph_caches", [])
        if not any(c is cache for c in held):
            held.append(cache)

    def unregister_graph_cache(self, cache: Dict[Any, Any]) -> None:
        held = self.__dict__.get("_graph_caches", [])
        held[:] = [c for c in held if c is not cache]

    def release_graphs(self) -> int:
        """Destroy every hipGraph recorded over this group (their runners re-record on the next call).
        Returns how many were destroyed."""
        n = 0
        for cache in self.__dict__.get("_graph_caches", []):
            for entry in list(cache.values()):
                graph = entry[0] if isinstance(entry, tuple) else entry
                reset = getattr(graph, "reset", None)
                if reset is not None:
                    reset()
                    n += 1
            cache.clear()
        return n


class SingleProcessGroup(ReplicaGroup):
    """All replicas in one process, executed in lock-step."""

    def __init__(self, n_shard: int) -> None:
        self.n_shard = n_shard
        self.local_shards = list(range(n_shard))

    def all_to_all(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        assert len(xs) == self.n_shard
        if self.n_shard == 1:
            return [xs[0]]
        # stacked[j, r] = block r of replica j  ->  out[r][j]
        stacked = torch.stack(xs, dim=0)
        swapped = stacked.transpose(0, 1).contiguous()
        return [swapped[r] for r in range(self.n_shard)]

    def all_gather(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        assert len(xs) == self.n_shard
        stacked = torch.stack(xs, dim=0)
        return [stacked for _ in range(self.n_shard)]

    def all_reduce_sum(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        if len(xs) == 1:
            return [xs[0]]
        total = xs[0].clone()
        for x in xs[1:]:
            total += x
        return [total for _ in xs]


class DistributedGroup(ReplicaGroup):
    """One replica per process; rank r of `process_group` holds shard r.

    With the "nccl" backend (RCCL) device tensors go straight over xGMI.  With
    "gloo" (CPU tests, or several ranks sharing one GPU in a test) device
    tensors are staged through host memory.
    """

    def __init__(self, process_group: Optional[dist.ProcessGroup] = None) -> None:
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.pg = process_group
        self.n_shard = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.local_shards = [self.rank]
        self.host_staging = dist.get_backend(process_group) == "gloo"

    def _in(self, x: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        return x.cpu() if (self.host_staging and x.is_cuda) else x

    def all_to_all(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        (x,) = xs
        if x.shape[0] != self.n_shard:
            raise ValueError(f"all_to_all: leading dim {x.shape[0]} != n_shard {self.n_shard}")
        src = self._in(x)
        out = torch.empty_like(src)
        dist.all_to_all_single(out, src, group=self.pg)
        return [out.to(x.device)]

    def all_gather(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        (x,) = xs
        src = self._in(x)
        if src.dim() == 0:
            src = src.reshape(1)
        # concatenated form ([n * d0, ...]) is the one every backend accepts
        out = torch.empty((self.n_shard * src.shape[0], *src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src, group=self.pg)
        return [out.reshape(self.n_shard, *x.shape).to(x.device)]

    def all_reduce_sum(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        (x,) = xs
        src = self._in(x)
        dist.all_reduce(src, op=dist.ReduceOp.SUM, group=self.pg)
        if src is not x:
            x.copy_(src)
        return [x]

    def barrier(self) -> None:
        dist.barrier(group=self.pg)


class NativeGroup(ReplicaGroup):
    """One replica per process; collectives through the C ABI (`bess_comm_*`, RCCL) on
    PyTorch's current HIP stream.

    The 128-byte RCCL id is made by rank 0 and handed round through an existing
    `torch.distributed` process group of any backend (used for nothing else but
    `barrier()`), or passed in (`unique_id=`; `world` / `rank` then come from the
    arguments, and torch.distributed is not needed at all).
    """

    @classmethod
    def from_communicator(cls, comm: Any, barrier: Optional[Any] = None) -> "NativeGroup":
        """The group of one rank of an existing communicator (`Communicator.init_all`: one process, several GPUs);
        `barrier`: a callable that synchronises the ranks' host threads (default: none)."""
        g = cls.__new__(cls)
        g.pg, g._has_dist = None, False
        g.n_shard, g.rank, g.device = comm.world, comm.rank, comm.device
        g.local_shards = [g.rank]
        g.comm = comm
        g._barrier = barrier
        return g

    def __init__(self, device: torch.device, process_group: Optional[dist.ProcessGroup] = None,
                 unique_id: Optional[bytes] = None, world: Optional[int] = None,
                 rank: Optional[int] = None) -> None:
        from besskge import _native as nat

        self.pg = process_group
        if unique_id is None:
            if not dist.is_initialized():
                raise RuntimeError("NativeGroup needs torch.distributed (to hand the RCCL id round) or `unique_id=`")
            world = dist.get_world_size(process_group)
            rank = dist.get_rank(process_group)
            box = [nat.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group else 0,
                                       group=process_group, device=torch.device("cpu")
                                       if dist.get_backend(process_group) == "gloo" else device)
            unique_id = box[0]
            self._has_dist = True
        else:
            if world is None or rank is None:
                raise ValueError("NativeGroup(unique_id=...) also needs world= and rank=")
            self._has_dist = dist.is_initialized()
        self.n_shard = int(world)
        self.rank = int(rank)
        self.local_shards = [self.rank]
        self.device = device
        self.comm = nat.Communicator(self.n_shard, self.rank, unique_id, device)

    def all_to_all(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        (x,) = xs
        return [self.comm.all_to_all(x.contiguous())]

    def all_gather(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        (x,) = xs
        return [self.comm.all_gather(x.contiguous())]

    def all_reduce_sum(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        (x,) = xs
        if x.dtype != torch.float32:
            y = self.comm.all_reduce_sum_(x.float().contiguous())
            x.copy_(y)
            return [x]
        if not x.is_contiguous():
            y = self.comm.all_reduce_sum_(x.contiguous())
            x.copy_(y)
            return [x]
        return [self.comm.all_reduce_sum_(x)]

    def pack_exchange(self, table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """K1 + C1 in one native call: rows `table[idx]` (idx [n_shard, L]) -> received [n_shard, L, W]."""
        return self.comm.pack_exchange(table, idx)[1]

    def all_agree(self, ok: bool) -> bool:
        if self.n_shard == 1:
            return bool(ok)
        flag = torch.tensor([0.0 if ok else 1.0], dtype=torch.float32, device=self.device)
        return float(self.comm.all_reduce_sum_(flag).item()) == 0.0

    def barrier(self) -> None:
        torch.cuda.current_stream(self.device).synchronize()
        if self._has_dist:
            dist.barrier(group=self.pg)
        elif getattr(self, "_barrier", None) is not None:
            self._barrier()

    def close(self) -> None:
        """Destroy the recorded steps that use the communicator, then the communicator (in that order:
        `ncclCommDestroy` waits for hipGraphs that recorded its collectives)."""
        if self.release_graphs():
            torch.cuda.synchronize(self.device)
        self.comm.graphs_released()
        self.comm.close()

    def __del__(self, _finalizing: Any = sys.is_finalizing) -> None:  # pragma: no cover - collection order
        if _finalizing():
            return  # the process is going: its memory and the communicator go with it, nothing to wait for
        try:
            self.close()
        except Exception:
            pass


class MultiDeviceGroup:
    """ONE host process driving n GPUs - the reference's own runtime contract (one process feeds `[bps * n, ...]`
    tensors and reads stacked outputs: `/root/reference/tests/test_bess.py:122-150`, `besskge/pipeline.py:129-144`;
    SURVEY section 7 "Process model").  `bess_comm_init_all` (= ncclCommInitAll) makes the n communicators of the
    clique; rank r lives on `devices[r]` and is driven by its own host thread (`besskge.runtime.MultiDeviceRunner`),
    whose library calls release the GIL - with `Options.use_plans` a step is one such call per device.

    `ranks[r]` is the `NativeGroup` of rank r (collectives through the C ABI on the kernels' stream of that device).
    """

    def __init__(self, devices: Any) -> None:
        import threading

        from besskge import _native as nat

        self.devices = [torch.device(d) if not isinstance(d, torch.device) else d for d in devices]
        if len({(d.type, d.index) for d in self.devices}) != len(self.devices):
            raise ValueError("MultiDeviceGroup: one rank per GPU (RCCL refuses a clique with a device twice)")
        self.n_shard = len(self.devices)
        self.local_shards = list(range(self.n_shard))
        self._host_barrier = threading.Barrier(self.n_shard)
        comms = nat.Communicator.init_all(self.devices)
        self.ranks = [NativeGroup.from_communicator(c, barrier=self._host_barrier.wait) for c in comms]

    def close(self) -> None:
        for g in self.ranks:
            g.close()
