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


def _save_table(model: Any, table: torch.Tensor, stem: Path, chunk_bytes: int) -> None:
    save_rows(table, f"{stem}.npy", chunk_bytes)
    st = getattr(model, "_optimizer_state", {}).get(table.data_ptr())
    meta: Dict[str, Any] = dict(step=0, n_state=0, paged=False)
    if st is not None:
        meta.update(step=int(st["step"]), n_state=len(st["s"]), paged="slot_map" in st)
        for i, s in enumerate(st["s"]):
            save_rows(s, f"{stem}.state{i}.npy", chunk_bytes)
        if "slot_map" in st:
            save_rows(st["slot_map"], f"{stem}.slots.npy", chunk_bytes)
            meta.update(capacity=int(st["capacity"]), counter=int(st["slot_counter"].item()))
        if "step_dev" in st:
            meta["step_dev"] = int(st["step_dev"].item())
    with open(f"{stem}.meta.json", "w") as f:
        json.dump(meta, f)


def _load_table(model: Any, table: torch.Tensor, stem: Path, chunk_bytes: int) -> None:
    load_rows(f"{stem}.npy", table, chunk_bytes)
    with open(f"{stem}.meta.json") as f:
        meta = json.load(f)
    states = getattr(model, "_optimizer_state", None)
    if states is not None:
        states.pop(table.data_ptr(), None)
    if meta["n_state"] == 0 and meta["step"] == 0:
        return
    if states is None:
        model._optimizer_state = states = {}
    st: Dict[str, Any] = dict(step=int(meta["step"]), s=[])
    dev = table.device
    if meta["paged"]:
        st["capacity"] = int(meta["capacity"])
        st["slot_map"] = load_rows(f"{stem}.slots.npy", torch.empty((table.shape[0],), dtype=torch.int32, device=dev),
                                   chunk_bytes)
        st["slot_counter"] = torch.full((1,), int(meta["counter"]), dtype=torch.int32, device=dev)
    shape = (st["capacity"], table.shape[1]) if meta["paged"] else tuple(table.shape)
    for i in range(int(meta["n_state"])):
        st["s"].append(load_rows(f"{stem}.state{i}.npy", torch.empty(shape, dtype=torch.float32, device=dev), chunk_bytes))
    if "step_dev" in meta:
        st["step_dev"] = torch.full((1,), int(meta["step_dev"]), dtype=torch.int32, device=dev)
    states[table.data_ptr()] = st


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
                opt[name] = dict(step=int(st["step"]), s=[x.detach().cpu() for x in st["s"]],
                                 step_dev=int(st["step_dev"].item()) if "step_dev" in st else -1)
        torch.save(dict(tensors={k: v.detach().cpu() for k, v in _dense_state(model).items()}, optimizer=opt),
                   directory / "dense.pt")


def load_checkpoint(model: Any, directory: Union[str, Path], chunk_bytes: int = CHUNK_BYTES) -> None:
    """Read the shards hosted by this process back in place (the tables keep their addresses, so a
    hipGraph recorded on them stays valid) and restore optimiser state, step counts and paging."""
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
        new: Dict[str, Any] = dict(step=int(st["step"]), s=[x.to(p.device) for x in st["s"]])
        if st["step_dev"] >= 0:
            new["step_dev"] = torch.full((1,), int(st["step_dev"]), dtype=torch.int32, device=p.device)
        model._optimizer_state[p.data_ptr()] = new
