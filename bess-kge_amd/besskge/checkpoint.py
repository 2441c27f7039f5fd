"""Per-shard streaming checkpoints of the embedding tables and their optimiser state.

The reference has no trainer-level checkpointing: a trained table goes back in through
`entity_initializer=<tensor>` (reference embedding.py:135-163).  That round trip is kept
(`besskge.embedding`); this module adds what a 128 GB shard (BASELINE config 5) needs on top of it:
every process writes and reads only the shards it hosts, and rows move between HBM and the file in
bounded chunks through one pinned staging buffer - nothing of the size of a table is ever held in
host memory.

Layout of a checkpoint directory (one per job, shared by all ranks):

    entity_shard<k>.npy            rows of shard k, [max_entity_per_shard, W], table dtype
    entity_shard<k>.state<i>.npy   optimiser state i of that table (fp32; state pools when paged)
    entity_shard<k>.slots.npy      row -> state-row map (paged state only)
    entity_shard<k>.meta.json      step count, paging capacity / counter
    relation.npy (+ .state<i>, .meta.json)   replicated: written by the process hosting shard 0
    dense.pt                       remaining (dense) parameters and buffers of the scoring function, and the
                                   optimiser state of those parameters (ConvE's network; small)
"""

import json
import os
from pathlib import Path
from typing import Any, Dict, Union

import numpy as np
import torch

#: bytes moved per copy; one pinned buffer of this size is the only host memory used
CHUNK_BYTES = 256 << 20

_NP = {torch.float32: np.float32, torch.float16: np.float16, torch.int32: np.int32, torch.int64: np.int64}


def _staging(t: torch.Tensor, rows: int) -> torch.Tensor:
    pin = t.is_cuda
    return torch.empty((rows,) + tuple(t.shape[1:]), dtype=t.dtype, pin_memory=pin)


def _rows_per_chunk(t: torch.Tensor, chunk_bytes: int) -> int:
    per_row = max(1, t[0].numel() * t.element_size()) if t.shape[0] else 1
    return max(1, min(max(1, t.shape[0]), chunk_bytes // per_row))


def save_rows(t: torch.Tensor, path: Union[str, Path], chunk_bytes: int = CHUNK_BYTES) -> None:
    """Stream a (device) tensor, row chunk by row chunk, into a `.npy` file (readable with `np.load`)."""
    if t.dtype not in _NP:
        raise TypeError(f"save_rows: dtype {t.dtype} not supported")
    t = t.detach()
    if t.dim() == 0:
        t = t.reshape(1)
    if not t.is_contiguous():
        raise ValueError("save_rows: tensor must be contiguous")
    tmp = f"{path}.tmp{os.getpid()}"
    out = np.lib.format.open_memmap(tmp, mode="w+", dtype=_NP[t.dtype], shape=tuple(t.shape))
    rows = _rows_per_chunk(t, chunk_bytes)
    stage = _staging(t, rows) if t.shape[0] else None
    for r0 in range(0, t.shape[0], rows):
        r1 = min(r0 + rows, t.shape[0])
        stage[: r1 - r0].copy_(t[r0:r1])  # synchronous for pinned destinations: safe to read right after
        out[r0:r1] = stage[: r1 - r0].numpy()
    out.flush()
    del out
    os.replace(tmp, path)  # a reader never sees a half-written file


def load_rows(path: Union[str, Path], into: torch.Tensor, chunk_bytes: int = CHUNK_BYTES) -> torch.Tensor:
    """Stream a `.npy` file written by `save_rows` into `into` (shape and dtype must match)."""
    src = np.load(path, mmap_mode="r", allow_pickle=False)
    dst = into.detach()
    if dst.dim() == 0:
        dst = dst.reshape(1)
    if tuple(src.shape) != tuple(dst.shape) or _NP.get(dst.dtype) != src.dtype.type:
        raise ValueError(f"{path}: holds {src.dtype} {tuple(src.shape)}, the tensor is {dst.dtype} {tuple(dst.shape)}")
    rows = _rows_per_chunk(dst, chunk_bytes)
    stage = _staging(dst, rows) if dst.shape[0] else None
    for r0 in range(0, dst.shape[0], rows):
        r1 = min(r0 + rows, dst.shape[0])
        stage[: r1 - r0].numpy()[...] = src[r0:r1]
        dst[r0:r1].copy_(stage[: r1 - r0])
        if dst.is_cuda:
            torch.cuda.current_stream(dst.device).synchronize()  # the staging buffer is reused
    return into


def _write_json(path: str, obj: Dict[str, Any]) -> None:
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as f:
        json.dump(obj, f)
    os.replace(tmp, path)


def _step_of(st: Dict[str, Any]) -> int:
    """Optimiser steps taken so far.  Under hipGraph replay only the device-side count advances (the host
    count is restored after the capture): the larger of the two is the truth."""
    step = int(st["step"])
    if "step_dev" in st:
        step = max(step, int(st["step_dev"].item()))
    return step


def _save_table(model: Any, table: torch.Tensor, stem: Path, chunk_bytes: int) -> None:
    save_rows(table, f"{stem}.npy", chunk_bytes)
    st = getattr(model, "_optimizer_state", {}).get(table.data_ptr())
    meta: Dict[str, Any] = dict(step=0, n_state=0, paged=False)
    if st is not None:
        meta.update(step=_step_of(st), n_state=len(st["s"]), paged="slot_map" in st)
        if "kind" in st:
            meta["kind"] = int(st["kind"])  # BESS_OPT_*: a state tensor means different things to different optimisers
        for i, s in enumerate(st["s"]):
            save_rows(s, f"{stem}.state{i}.npy", chunk_bytes)
        if "slot_map" in st:
            save_rows(st["slot_map"], f"{stem}.slots.npy", chunk_bytes)
            meta.update(capacity=int(st["capacity"]), counter=int(st["slot_counter"].item()))
        if "step_dev" in st:
            meta["step_dev"] = _step_of(st)
    _write_json(f"{stem}.meta.json", meta)


def _stale_graphs(model: Any) -> None:
    """State tensors were replaced: hipGraphs recorded by a Runner hold the old addresses.  Runners compare
    this counter with the one they recorded under and re-capture (`Runner._call_with_graphs`)."""
    model.__dict__["_state_generation"] = model.__dict__.get("_state_generation", 0) + 1


def _load_table(model: Any, table: torch.Tensor, stem: Path, chunk_bytes: int) -> None:
    load_rows(f"{stem}.npy", table, chunk_bytes)
    with open(f"{stem}.meta.json") as f:
        meta = json.load(f)
    states = getattr(model, "_optimizer_state", None)
    have = states.get(table.data_ptr()) if states is not None else None
    dev = table.device
    paged = bool(meta["paged"])
    n_state, step = int(meta["n_state"]), int(meta["step"])
    shape = (int(meta["capacity"]), table.shape[1]) if paged else tuple(table.shape)
    if have is not None and n_state > 0 and "kind" in have and "kind" in meta and int(have["kind"]) != int(meta["kind"]):
        raise ValueError(
            f"{stem}: the checkpoint's optimiser state was written by optimiser kind {meta['kind']}, the live state "
            f"belongs to kind {have['kind']} (0 SGD, 1 Adagrad, 2 Adam): its tensors mean something else there")
    fresh = n_state == 0 and step == 0  # a checkpoint without optimiser history: the live state starts over
    if have is not None and ("slot_map" in have) == paged and (len(have["s"]) == n_state or fresh) \
            and all(tuple(x.shape) == shape for x in have["s"]) and (not paged or have["capacity"] == int(meta["capacity"])):
        # same layout as the live state: read into the existing tensors - a hipGraph recorded on them (tables,
        # moments, slot map, device-side step count) stays valid.  (Fewer state tensors in the file than live -
        # a momentum checkpoint under Adam - is NOT the same layout: zeroed moments under the checkpoint's step
        # count would skip Adam's bias correction, first updates ~30x too large.)
        for i, x in enumerate(have["s"]):
            if i < n_state:
                load_rows(f"{stem}.state{i}.npy", x, chunk_bytes)
            else:
                x.zero_()
        if paged:
            load_rows(f"{stem}.slots.npy", have["slot_map"], chunk_bytes)
            have["slot_counter"].fill_(int(meta["counter"]))
        have["step"] = step
        if "step_dev" in have:
            have["step_dev"].fill_(step)
        elif "step_dev" in meta:
            have["step_dev"] = torch.full((1,), step, dtype=torch.int32, device=dev)
            _stale_graphs(model)
        return
    if have is not None:
        states.pop(table.data_ptr(), None)
        _stale_graphs(model)
    if n_state == 0 and step == 0:
        return
    if states is None:
        model._optimizer_state = states = {}
    st: Dict[str, Any] = dict(step=step, s=[])
    if paged:
        st["capacity"] = int(meta["capacity"])
        st["slot_map"] = load_rows(f"{stem}.slots.npy", torch.empty((table.shape[0],), dtype=torch.int32, device=dev),
                                   chunk_bytes)
        st["slot_counter"] = torch.full((1,), int(meta["counter"]), dtype=torch.int32, device=dev)
    for i in range(n_state):
        st["s"].append(load_rows(f"{stem}.state{i}.npy", torch.empty(shape, dtype=torch.float32, device=dev), chunk_bytes))
    if "step_dev" in meta:
        st["step_dev"] = torch.full((1,), step, dtype=torch.int32, device=dev)
    if "kind" in meta:
        st["kind"] = int(meta["kind"])
    states[table.data_ptr()] = st
    _stale_graphs(model)


def _dense_state(model: Any) -> Dict[str, torch.Tensor]:
    fn = model.score_fn
    skip = {"entity_embedding", "relation_embedding"}
    return {k: v for k, v in fn.state_dict().items() if k not in skip}


def save_checkpoint(model: Any, directory: Union[str, Path], chunk_bytes: int = CHUNK_BYTES) -> None:
    """Write the shards hosted by this process (tables + optimiser state) under `directory`.

    Every rank calls it with the same directory; files of different ranks do not overlap.  The
    relation table and the dense parameters are replicated: the process hosting shard 0 writes them."""
    directory = Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    group = model._group()
    for dev in {model._local_table(s).device for s in group.local_shards}:
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
    for shard in group.local_shards:
        _save_table(model, model._local_table(shard), directory / f"entity_shard{shard}", chunk_bytes)
    if 0 in group.local_shards:
        _save_table(model, model.score_fn.relation_embedding.data, directory / "relation", chunk_bytes)
        states = getattr(model, "_optimizer_state", {})
        opt = {}
        for name, p in model.score_fn.named_parameters():
            st = states.get(p.data_ptr())
            if st is not None and name not in ("entity_embedding", "relation_embedding"):
                opt[name] = dict(step=_step_of(st), s=[x.detach().cpu() for x in st["s"]],
                                 step_dev=_step_of(st) if "step_dev" in st else -1)
        tmp = directory / f"dense.pt.tmp{os.getpid()}"
        torch.save(dict(tensors={k: v.detach().cpu() for k, v in _dense_state(model).items()}, optimizer=opt), tmp)
        os.replace(tmp, directory / "dense.pt")
    # every rank's files are in place before anyone returns (and before a rank could start loading the
    # replicated files the host of shard 0 writes)
    if len(group.local_shards) != group.n_shard and hasattr(group, "barrier"):
        group.barrier()


def load_checkpoint(model: Any, directory: Union[str, Path], chunk_bytes: int = CHUNK_BYTES) -> None:
    """Read the shards hosted by this process back in place and restore optimiser state, step counts and paging.
    The tables keep their addresses.  Optimiser state of the same layout as the live one (same number of state
    tensors, same paging) is read into the existing tensors too, so a hipGraph a Runner recorded earlier stays
    valid; where the layout differs the state objects are replaced and the model's `_state_generation` is bumped
    - Runners drop the graphs they recorded under an older generation and capture again on their next call."""
    directory = Path(directory)
    group = model._group()
    for shard in group.local_shards:
        _load_table(model, model._local_table(shard), directory / f"entity_shard{shard}", chunk_bytes)
    _load_table(model, model.score_fn.relation_embedding.data, directory / "relation", chunk_bytes)
    blob = torch.load(directory / "dense.pt", map_location="cpu", weights_only=True)
    dense, own = blob["tensors"], _dense_state(model)
    if set(dense) != set(own):
        raise ValueError(f"dense.pt holds {sorted(dense)}, the scoring function has {sorted(own)}")
    with torch.no_grad():
        for k, v in dense.items():
            own[k].copy_(v)
    params = dict(model.score_fn.named_parameters())
    for name, st in blob["optimizer"].items():
        p = params[name]
        if not hasattr(model, "_optimizer_state"):
            model._optimizer_state = {}
        have = model._optimizer_state.get(p.data_ptr())
        if have is not None and len(have["s"]) == len(st["s"]) and all(
                a.shape == b.shape for a, b in zip(have["s"], st["s"])) and (("step_dev" in have) == (st["step_dev"] >= 0)):
            for a, b in zip(have["s"], st["s"]):
                a.copy_(b)
            have["step"] = int(st["step"])
            if "step_dev" in have:
                have["step_dev"].fill_(int(st["step_dev"]))
            continue
        new: Dict[str, Any] = dict(step=int(st["step"]), s=[x.to(p.device) for x in st["s"]])
        if st["step_dev"] >= 0:
            new["step_dev"] = torch.full((1,), int(st["step_dev"]), dtype=torch.int32, device=p.device)
        model._optimizer_state[p.data_ptr()] = new
        _stale_graphs(model)
