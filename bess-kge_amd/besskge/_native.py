"""ctypes binding of `libbesskge_hip.so` (C ABI: include/besskge_hip.h).

This is the only place the Python side touches the native library.  Tensors
are validated here (device, dtype, contiguity, shapes) so that a kernel never
sees operand shapes its grid does not assume; then raw device pointers and
PyTorch's current HIP stream are handed over.  PyTorch is used for device
memory and streams only.

There is deliberately no CPU fallback: every wrapper raises if a tensor is not
on a CUDA/HIP device or if the library is missing.
"""

import contextlib
import ctypes
import pathlib
import sys
from typing import Any, List, Optional, Sequence, Tuple

import torch

LIB_NAME = "libbesskge_hip.so"
ABI_VERSION = 3

TRANSE, ROTATE, DISTMULT, COMPLEX, AFFINE, BOXE = 0, 1, 2, 3, 4, 5
F32, F16 = 0, 1
CORRUPT_HEAD, CORRUPT_TAIL = 0, 1
LOSS_LOGSIGMOID, LOSS_MARGIN, LOSS_SSCE = 0, 1, 2
BAD_NEGATIVE_SCORE = -50000.0

_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_f32p = ctypes.POINTER(ctypes.c_float)
_c_u8p = ctypes.POINTER(ctypes.c_uint8)
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_RETURNS_I64 = ("bess_neg_score_shared_workspace", "bess_neg_score_shared_bwd_workspace",
                "bess_neg_score_shared_fwd_counts_workspace", "bess_neg_score_shared_fwd_pairs_workspace")  # every other entry returns an int status
_i32 = ctypes.c_int32
_f32 = ctypes.c_float


class ModelDesc(ctypes.Structure):
    """struct bess_model_desc"""

    _fields_ = [
        ("scorer", _i32),
        ("norm_p", _i32),
        ("dtype", _i32),
        ("width", _i32),
        ("rel_width", _i32),
        ("reserved", _i32 * 3),
    ]


class LossDesc(ctypes.Structure):
    """struct bess_loss_desc"""

    _fields_ = [
        ("kind", _i32),
        ("adversarial", _i32),
        ("margin", _f32),
        ("adversarial_scale", _f32),
        ("loss_scale", _f32),
        ("ssce_shift", _f32),
        ("reserved", _i32 * 2),
    ]


class OptDesc(ctypes.Structure):
    """struct bess_opt_desc"""

    _fields_ = [
        ("kind", _i32),
        ("step", _i32),
        ("lr", _f32),
        ("momentum", _f32),
        ("beta1", _f32),
        ("beta2", _f32),
        ("eps", _f32),
        ("weight_decay", _f32),
        ("step_ptr", _i64),
        ("slot_map", _i64),
    ]


OPT_SGD, OPT_ADAGRAD, OPT_ADAM = 0, 1, 2
FLAG_FP32_MATH = 1  # BESS_FLAG_FP32_MATH (ModelDesc.reserved[0] of the four native scorers)
FLAG_PREZEROED = 2  # BESS_FLAG_PREZEROED: the targets of bess_neg_score_shared_bwd are zero on entry
FLAG_DNEG_BY_ROW = 4  # BESS_FLAG_DNEG_BY_ROW: bess_neg_score_pertriple_bwd stores d_neg at row neg_idx[k] of a row-space matrix


class KillDesc(ctypes.Structure):
    """struct bess_kill_desc: K7 (mask / augment kill) applied with the scores"""

    _fields_ = [
        ("diag_step", _i32),
        ("ht", _i32),
        ("ppp", _i32),
        ("reserved", _i32),
        ("mask", _vp),
        ("mask_rows", _i64),
        ("mask_cols", _i64),
    ]


class Pcg64State(ctypes.Structure):
    """struct bess_pcg64_state"""

    _fields_ = [
        ("state_hi", ctypes.c_uint64),
        ("state_lo", ctypes.c_uint64),
        ("inc_hi", ctypes.c_uint64),
        ("inc_lo", ctypes.c_uint64),
        ("has_uint32", ctypes.c_uint32),
        ("uinteger", ctypes.c_uint32),
    ]


_PG = ctypes.POINTER(Pcg64State)

_MD = ctypes.POINTER(ModelDesc)
_LD = ctypes.POINTER(LossDesc)

# name -> argtypes; every function returns int
SIGNATURES = {
    "bess_version": [],
    "bess_last_error": [ctypes.c_char_p, ctypes.c_size_t],
    "bess_gather_rows": [_i32, _i32, _vp, _vp, _i64, _vp, _vp],
    "bess_score_triple_fwd": [_MD, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "bess_score_triple_bwd": [_MD, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "bess_query_fwd": [_MD, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "bess_query_bwd": [_MD, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "bess_query_triple_fwd": [_MD, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "bess_query_triple_bwd": [_MD, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "bess_sparse_sgd_lists": [_i32, _i32, _vp, _i32, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64), _f32, _vp],
    "bess_sparse_sgd_lists_axpy": [_i32, _i32, _vp, _i32, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64), _f32,
                                   _vp, _vp, _i64, _f32, _vp],
    "bess_neg_score_pertriple_fwd": [_MD, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp],
    "bess_neg_score_pertriple_bwd": [_MD, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp],
    "bess_neg_score_shared_fwd": [_MD, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp],
    "bess_neg_score_shared_workspace": [_MD, _i64, _i64],
    "bess_neg_score_shared_fwd_ws": [_MD, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp],
    "bess_neg_score_shared_fwd_pruned": [_MD, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp],
    "bess_neg_score_shared_fwd_counts_workspace": [_MD, _i64, _i64],
    "bess_neg_score_shared_fwd_counts": [_MD, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp, _i64, _vp],
    "bess_neg_score_shared_fwd_pairs_workspace": [_MD, _i64, _i64],
    "bess_neg_score_shared_fwd_pairs": [_MD, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp],
    "bess_topk_update_flagged": [_vp, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _vp, _i32, _vp],
    "bess_neg_score_shared_fwd_masked": [_MD, _vp, _i64, _vp, _vp, _i64, _vp, _i64, ctypes.POINTER(KillDesc), _vp, _i64, _vp],
    "bess_neg_score_shared_bwd": [_MD, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp],
    "bess_neg_score_shared_bwd_workspace": [_MD, _i64, _i64],
    "bess_neg_score_shared_bwd_ws": [_MD, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp],
    "bess_mask_scores": [_vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _i64, _i64, _vp],
    "bess_loss_fwd_bwd": [_LD, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp],
    "bess_loss_fwd_bwd_norm": [_LD, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "bess_loss_fwd_bwd_one_launch": [_LD, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "bess_scatter_add_rows": [_vp, _i32, _vp, _vp, _i64, _f32, _vp],
    "bess_sparse_sgd": [_i32, _i32, _vp, _vp, _vp, _i64, _f32, _vp],
    "bess_dense_sgd": [_i32, _vp, _vp, _i64, _f32, _vp],
    "bess_segment_index_workspace": [_i64, ctypes.POINTER(ctypes.c_size_t)],
    "bess_build_segment_index": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, ctypes.c_size_t, _vp],
    "bess_pad_segments": [_vp, _vp, _i64, _vp, _i32, _vp],
    "bess_step_prologue": [_i32, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_uint32),
                           ctypes.POINTER(_i64), _i32, ctypes.POINTER(_vp), ctypes.POINTER(_i64), _i32, _vp, _vp, _vp, _vp,
                           _vp, _i64, _vp],
    "bess_neg_pertriple_grad_segments": [_MD, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _f32,
                                         _vp, _i64, _vp, _vp, _vp],
    "bess_apply_segments_sgd": [_i32, _i32, _vp, _vp, _vp, _i64, _vp, _f32, _vp],
    "bess_segment_sum_rows": [_i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "bess_topk_update": [_vp, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _i32, _vp],
    "bess_ranks_from_scores": [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _vp],
    "bess_ranks_from_indices": [_vp, _vp, _i64, _i64, _i32, _vp, _vp],
    "bess_apply_segments_opt": [ctypes.POINTER(OptDesc), _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "bess_neg_pertriple_step_segments": [_MD, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64,
                                         _vp, _i64, _vp, _vp, ctypes.POINTER(OptDesc), _vp, _vp, _vp, _vp, _vp],
    "bess_map_extra_rows": [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp],
    "bess_assign_state_rows": [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp],
    "bess_coalesced_update": [ctypes.POINTER(OptDesc), _i32, _i32, _vp, _i32, ctypes.POINTER(_vp), ctypes.POINTER(_i64),
                              _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "bess_coalesced_update_axpy": [ctypes.POINTER(OptDesc), _i32, _i32, _vp, _i32, ctypes.POINTER(_vp), ctypes.POINTER(_i64),
                                   _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _vp],
    "bess_neg_pertriple_items": [_MD, _i64, _i64, ctypes.POINTER(ctypes.c_int32)],
    "bess_neg_score_pertriple_fwd_dq": [_MD, _LD, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp],
    "bess_neg_score_pertriple_fwd_dq_masked": [_MD, _LD, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp,
                                               _vp, _vp, _vp],
    "bess_neg_score_pertriple_fwd_partials": [_MD, _LD, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp],
    "bess_combine_dq_partials": [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp],
    "bess_normalize_rows": [_i32, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp],
    "bess_normalize_rows_bwd": [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp],
    "bess_sample_negatives": [_PG, _vp, _i64, _i32, _i32, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "bess_sample_bucket_indices": [_PG, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp],
    "bess_lookup_triples": [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp],
    "bess_gather_candidate_lists": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp],
    "bess_comm_unique_id": [_c_u8p],
    "bess_comm_init_rank": [_i32, _i32, _c_u8p, ctypes.POINTER(_vp)],
    "bess_comm_init_all": [_i32, _c_i32p, ctypes.POINTER(_vp)],
    "bess_comm_destroy": [_vp],
    "bess_comm_info": [_vp, _c_i32p, _c_i32p, _c_i32p],
    "bess_alltoall": [_vp, _vp, _vp, _i64, _vp],
    "bess_allgather": [_vp, _vp, _vp, _i64, _vp],
    "bess_allreduce_sum_f32": [_vp, _vp, _vp, _i64, _vp],
    "bess_pack_exchange": [_vp, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp],
    "bess_graph_node_counts": [_vp, _c_i32p, _i32],
    "bess_plan_create": [ctypes.POINTER(_vp)],
    "bess_plan_destroy": [_vp],
    "bess_plan_knows": [ctypes.c_char_p],
    "bess_plan_length": [_vp],
    "bess_plan_add_call": [_vp, ctypes.c_char_p, _i32, _c_u8p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(_vp),
                           ctypes.POINTER(_i64)],
    "bess_plan_run": [_vp, _vp],
    "bess_direct_update": [ctypes.POINTER(OptDesc), _i32, _i32, _vp, _i32, ctypes.POINTER(_vp), ctypes.POINTER(_i64), _vp, _vp,
                           _vp, _vp, _vp, _vp, _vp, _i64, _f32, _vp],
    "bess_neg_score_shared_bwd_parts_plan": [_MD, _i64, _i64, _c_i32p, _c_i32p],
    "bess_neg_score_shared_bwd_parts": [_MD, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp],
    "bess_query_triple_bwd_parts": [_MD, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _i64, _vp,
                                    _vp, _vp, _vp, _vp, _vp],
    "bess_pertriple_tail_supported": [_MD, _i64],
    "bess_pertriple_tail": [_MD, _LD, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _i64, _i64, _vp,
                            _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "bess_query_triple_fwd_jobs": [_MD, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, ctypes.POINTER(_vp),
                                   ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(_i64), _vp],
    "bess_neg_score_shared_fwd_loss": [_MD, _vp, _i64, _vp, _vp, _i64, _vp, _i64, ctypes.POINTER(KillDesc), _LD, _vp, _vp, _i64,
                                       _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp],
}
COMM_ID_BYTES = 128
TICKET_INTS = 544  # BESS_TICKET_INTS
ECOMM_BASE = 10000

_lib: Optional[ctypes.CDLL] = None

# Optional per-launch timing of selected entry points with HIP events recorded
# on the stream the kernel is launched on (bench.py's roofline leg).
_timing: Optional[list] = None
_timed_names: Tuple[str, ...] = ()


def start_kernel_timing(names: Sequence[str]) -> None:
    """Record a HIP event pair around every call of the named entry points."""
    global _timing, _timed_names
    _timing, _timed_names = [], tuple(names)


def stop_kernel_timing() -> dict:
    """{entry point: [milliseconds per launch]} since start_kernel_timing()."""
    global _timing
    rec, _timing = _timing or [], None
    torch.cuda.synchronize()
    out: dict = {}
    for name, a, b in rec:
        out.setdefault(name, []).append(a.elapsed_time(b))
    return out


class _Timed:
    def __init__(self, name: str, dev: torch.device) -> None:
        self.on = _timing is not None and name in _timed_names
        self.name, self.dev = name, dev

    def __enter__(self) -> None:
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream(self.dev))

    def __exit__(self, *exc: Any) -> None:
        if self.on:
            self.b.record(torch.cuda.current_stream(self.dev))
            _timing.append((self.name, self.a, self.b))  # type: ignore


def library_path() -> pathlib.Path:
    """In-tree location of the shared library (built by `__graft_entry__.build`)."""
    return pathlib.Path(__file__).parent.absolute() / LIB_NAME


def load() -> ctypes.CDLL:
    """dlopen the HIP library; ImportError if it is missing or ABI-incompatible.

    Same role and failure mode as the reference's `load_custom_ops_so`
    (reference `besskge/__init__.py:10-37`).
    """
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise ImportError(
            f"Cannot find the HIP extension library {LIB_NAME} - tried {[str(path)]}."
            " Build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (or `make -C bess-kge_amd/csrc`)."
        )
    lib = ctypes.CDLL(str(path))
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{path} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64 if name in _RETURNS_I64 else ctypes.c_int
    if lib.bess_version() != ABI_VERSION:
        raise ImportError(
            f"{path}: ABI version {lib.bess_version()} != expected {ABI_VERSION}"
        )
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        buf = ctypes.create_string_buffer(512)
        load().bess_last_error(buf, 512)
        kind = "invalid argument" if rc < 0 else (f"ncclResult {rc - ECOMM_BASE}" if rc >= ECOMM_BASE else f"hipError {rc}")
        raise RuntimeError(f"besskge native call {what} failed ({kind}): {buf.value.decode()}")


# --------------------------------------------------------------------------- #
# tensor checks
def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f"embedding tables must be float32 or float16, got {t.dtype}")


def _dev(t: torch.Tensor, name: str) -> torch.device:
    if not t.is_cuda:
        raise RuntimeError(
            f"besskge: `{name}` is on {t.device}; the BESS hot path runs on"
            " HIP devices only (there is no CPU fallback)"
        )
    return t.device


def _same_device(ts: Sequence[Tuple[str, Optional[torch.Tensor]]]) -> torch.device:
    dev = None
    for name, t in ts:
        if t is None:
            continue
        d = _dev(t, name)
        if dev is None:
            dev = d
        elif d != dev:
            raise RuntimeError(f"besskge: `{name}` is on {d}, expected {dev}")
    assert dev is not None
    return dev


def _rows(base: torch.Tensor, name: str, width: int) -> None:
    if base.dim() != 2 or base.shape[1] != width or not base.is_contiguous():
        raise ValueError(
            f"`{name}` must be a contiguous [rows, {width}] tensor, got"
            f" shape {tuple(base.shape)} stride {base.stride()}"
        )


def _idx(idx: Optional[torch.Tensor], name: str, n: Optional[int] = None) -> int:
    if idx is None:
        return 0
    if idx.dtype != torch.int32 or idx.dim() != 1 or not idx.is_contiguous():
        raise ValueError(
            f"`{name}` must be a contiguous 1-D int32 tensor, got {idx.dtype} {tuple(idx.shape)}"
        )
    if n is not None and idx.numel() != n:
        raise ValueError(f"`{name}` has {idx.numel()} entries, expected {n}")
    return idx.data_ptr()


def _f32(t: torch.Tensor, name: str) -> None:
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"`{name}` must be contiguous float32, got {t.dtype}")


def _row_count(base: torch.Tensor, idx: Optional[torch.Tensor]) -> int:
    return int(idx.numel()) if idx is not None else int(base.shape[0])


def _stream(dev: torch.device) -> int:
    """Raw handle of PyTorch's current HIP stream on `dev` (no Stream object is built: this runs per launch)."""
    return torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())


_NO_SWITCH = contextlib.nullcontext()


def _on(dev: torch.device) -> Any:
    """`torch.cuda.device(dev)` only when `dev` is not the current device already (the usual case: one
    process per GPU) - the context manager costs more host time than the launch it wraps."""
    if dev.index is None or torch.cuda.current_device() == dev.index:
        return _NO_SWITCH
    return torch.cuda.device(dev)


class RowSource:
    """`rows[i] = base[idx[i]]` (idx None: identity).  `base` is a contiguous
    [*, W] table-dtype tensor: a shard, a receive buffer or plain embeddings."""

    __slots__ = ("base", "idx")

    def __init__(self, base: torch.Tensor, idx: Optional[torch.Tensor] = None) -> None:
        self.base = base
        self.idx = idx

    def __len__(self) -> int:
        return _row_count(self.base, self.idx)


def copy_desc(d: ModelDesc) -> ModelDesc:
    c = ModelDesc()
    ctypes.memmove(ctypes.byref(c), ctypes.byref(d), ctypes.sizeof(ModelDesc))
    return c


def make_desc(scorer: int, norm_p: int, table: torch.Tensor, rel_width: int) -> ModelDesc:
    d = ModelDesc()
    d.scorer = scorer
    d.norm_p = norm_p
    d.dtype = _dtype_code(table)
    d.width = int(table.shape[-1])
    d.rel_width = rel_width
    return d


# --------------------------------------------------------------------------- #
# wrappers (one per C entry point)
def gather_rows(table: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    dev = _same_device([("table", table), ("idx", idx), ("out", out)])
    W = int(table.shape[1])
    _rows(table, "table", W)
    n = idx.numel()
    ip = _idx(idx, "idx")
    if out is None:
        out = torch.empty((n, W), dtype=table.dtype, device=dev)
    else:
        _rows(out, "out", W)
        if out.shape[0] != n or out.dtype != table.dtype:
            raise ValueError("gather_rows: `out` does not match idx / table dtype")
    with _on(dev), _Timed("bess_gather_rows", dev):
        rc = load().bess_gather_rows(_dtype_code(table), W, table.data_ptr(), ip, n, out.data_ptr(), _stream(dev))
    _check(rc, "bess_gather_rows")
    return out


def _triple_operands(d: ModelDesc, head: RowSource, tail: RowSource, rel_table: torch.Tensor, rel_idx: torch.Tensor):
    dev = _same_device(
        [("head", head.base), ("head_idx", head.idx), ("tail", tail.base), ("tail_idx", tail.idx),
         ("relation_embedding", rel_table), ("relation", rel_idx)]
    )
    _rows(head.base, "head rows", d.width)
    _rows(tail.base, "tail rows", d.width)
    _rows(rel_table, "relation_embedding", d.rel_width)
    for t, nm in ((head.base, "head rows"), (tail.base, "tail rows"), (rel_table, "relation_embedding")):
        if _dtype_code(t) != d.dtype:
            raise TypeError(f"`{nm}` dtype {t.dtype} does not match the model descriptor")
    n = len(head)
    if len(tail) != n:
        raise ValueError(f"{n} head rows but {len(tail)} tail rows")
    _idx(rel_idx, "relation", n)
    return dev, n


def score_triple_fwd(d: ModelDesc, head: RowSource, tail: RowSource, rel_table: torch.Tensor,
                     rel_idx: torch.Tensor) -> torch.Tensor:
    dev, n = _triple_operands(d, head, tail, rel_table, rel_idx)
    out = torch.empty((n,), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = load().bess_score_triple_fwd(
            ctypes.byref(d), head.base.data_ptr(), _idx(head.idx, "head_idx"), tail.base.data_ptr(),
            _idx(tail.idx, "tail_idx"), rel_table.data_ptr(), rel_idx.data_ptr(), n, out.data_ptr(), _stream(dev))
    _check(rc, "bess_score_triple_fwd")
    return out


def score_triple_bwd(d: ModelDesc, head: RowSource, tail: RowSource, rel_table: torch.Tensor,
                     rel_idx: torch.Tensor, d_out: torch.Tensor, d_rel_table: torch.Tensor
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (d_head [n, W], d_tail [n, W]); accumulates into d_rel_table."""
    dev, n = _triple_operands(d, head, tail, rel_table, rel_idx)
    _same_device([("d_out", d_out), ("d_rel_table", d_rel_table), ("x", head.base)])
    _f32(d_out, "d_out")
    _f32(d_rel_table, "d_rel_table")
    if d_out.numel() != n or tuple(d_rel_table.shape) != tuple(rel_table.shape):
        raise ValueError("score_triple_bwd: gradient shapes do not match")
    dh = torch.empty((n, d.width), dtype=torch.float32, device=dev)
    dt = torch.empty((n, d.width), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = load().bess_score_triple_bwd(
            ctypes.byref(d), head.base.data_ptr(), _idx(head.idx, "head_idx"), tail.base.data_ptr(),
            _idx(tail.idx, "tail_idx"), rel_table.data_ptr(), rel_idx.data_ptr(), n, d_out.data_ptr(),
            dh.data_ptr(), dt.data_ptr(), d_rel_table.data_ptr(), _stream(dev))
    _check(rc, "bess_score_triple_bwd")
    return dh, dt


def _job_arrays(jobs: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], int]], who: str):
    """ctypes arrays (dst, src, value, words) of copy / fill jobs over 4-byte elements (`step_prologue`)."""
    nj = len(jobs)
    if nj > MAX_WORD_JOBS:
        raise ValueError(f"{who}: too many jobs")
    dst, src = (_vp * max(1, nj))(), (_vp * max(1, nj))()
    val, words = (ctypes.c_uint32 * max(1, nj))(), (_i64 * max(1, nj))()
    for i, (d_, s_, v_) in enumerate(jobs):
        if d_.element_size() != 4 or not d_.is_contiguous():
            raise ValueError(f"{who}: job targets must be contiguous tensors of 4-byte elements")
        if s_ is not None and (s_.element_size() != 4 or not s_.is_contiguous() or s_.numel() != d_.numel()):
            raise ValueError(f"{who}: a job's source must match its target")
        dst[i], src[i], val[i], words[i] = d_.data_ptr(), (s_.data_ptr() if s_ is not None else None), int(v_), d_.numel()
    return dst, src, val, words


def query_triple_fwd(d: ModelDesc, side: int, head: RowSource, tail: RowSource, rel_table: torch.Tensor,
                     rel_idx: torch.Tensor, jobs: Optional[Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], int]]] = None
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(query [n, W], positive score [n]) in one launch.  `jobs`: copy / fill jobs as `step_prologue` takes them, run
    by spare workgroups of the same launch (`bess_query_triple_fwd_jobs`)."""
    dev, n = _triple_operands(d, head, tail, rel_table, rel_idx)
    q = torch.empty((n, d.width), dtype=torch.float32, device=dev)
    out = torch.empty((n,), dtype=torch.float32, device=dev)
    if jobs:
        _same_device([("query rows", head.base)] + [("job dst", j[0]) for j in jobs] + [("job src", j[1]) for j in jobs])
        dst, src, val, words = _job_arrays(jobs, "query_triple_fwd")
        with _on(dev):
            rc = load().bess_query_triple_fwd_jobs(
                ctypes.byref(d), side, head.base.data_ptr(), _idx(head.idx, "head_idx"), tail.base.data_ptr(),
                _idx(tail.idx, "tail_idx"), rel_table.data_ptr(), rel_idx.data_ptr(), n, q.data_ptr(), out.data_ptr(),
                len(jobs), dst, src, val, words, _stream(dev))
        _check(rc, "bess_query_triple_fwd_jobs")
        return q, out
    with _on(dev):
        rc = load().bess_query_triple_fwd(
            ctypes.byref(d), side, head.base.data_ptr(), _idx(head.idx, "head_idx"), tail.base.data_ptr(),
            _idx(tail.idx, "tail_idx"), rel_table.data_ptr(), rel_idx.data_ptr(), n, q.data_ptr(), out.data_ptr(),
            _stream(dev))
    _check(rc, "bess_query_triple_fwd")
    return q, out


def query_triple_bwd(d: ModelDesc, side: int, head: RowSource, tail: RowSource, rel_table: torch.Tensor,
                     rel_idx: torch.Tensor, d_out: torch.Tensor, d_query: torch.Tensor, d_rel_table: torch.Tensor
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(d_head [n, W], d_tail [n, W]) of positive score and query together; accumulates into d_rel_table."""
    dev, n = _triple_operands(d, head, tail, rel_table, rel_idx)
    _same_device([("d_out", d_out), ("d_query", d_query), ("d_rel_table", d_rel_table), ("x", head.base)])
    for t, nm in ((d_out, "d_out"), (d_query, "d_query"), (d_rel_table, "d_rel_table")):
        _f32(t, nm)
    if d_out.numel() != n or tuple(d_query.shape) != (n, d.width) or tuple(d_rel_table.shape) != tuple(rel_table.shape):
        raise ValueError("query_triple_bwd: gradient shapes do not match")
    dh = torch.empty((n, d.width), dtype=torch.float32, device=dev)
    dt = torch.empty((n, d.width), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = load().bess_query_triple_bwd(
            ctypes.byref(d), side, head.base.data_ptr(), _idx(head.idx, "head_idx"), tail.base.data_ptr(),
            _idx(tail.idx, "tail_idx"), rel_table.data_ptr(), rel_idx.data_ptr(), n, d_out.data_ptr(),
            d_query.data_ptr(), dh.data_ptr(), dt.data_ptr(), d_rel_table.data_ptr(), _stream(dev))
    _check(rc, "bess_query_triple_bwd")
    return dh, dt


def _query_operands(d: ModelDesc, ent: RowSource, rel_table: torch.Tensor, rel_idx: torch.Tensor):
    dev = _same_device([("entity rows", ent.base), ("entity idx", ent.idx),
                        ("relation_embedding", rel_table), ("relation", rel_idx)])
    _rows(ent.base, "entity rows", d.width)
    _rows(rel_table, "relation_embedding", d.rel_width)
    if _dtype_code(ent.base) != d.dtype or _dtype_code(rel_table) != d.dtype:
        raise TypeError("query: operand dtype does not match the model descriptor")
    n = len(ent)
    _idx(rel_idx, "relation", n)
    return dev, n


def query_fwd(d: ModelDesc, side: int, ent: RowSource, rel_table: torch.Tensor, rel_idx: torch.Tensor) -> torch.Tensor:
    dev, n = _query_operands(d, ent, rel_table, rel_idx)
    q = torch.empty((n, query_width(d)), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = load().bess_query_fwd(ctypes.byref(d), side, ent.base.data_ptr(), _idx(ent.idx, "entity idx"),
                                   rel_table.data_ptr(), rel_idx.data_ptr(), n, q.data_ptr(), _stream(dev))
    _check(rc, "bess_query_fwd")
    return q


def query_bwd(d: ModelDesc, side: int, ent: RowSource, rel_table: torch.Tensor, rel_idx: torch.Tensor,
              d_query: torch.Tensor, d_rel_table: torch.Tensor) -> torch.Tensor:
    """Returns d_ent [n, W]; accumulates into d_rel_table."""
    dev, n = _query_operands(d, ent, rel_table, rel_idx)
    _same_device([("d_query", d_query), ("d_rel_table", d_rel_table), ("x", ent.base)])
    _f32(d_query, "d_query")
    _f32(d_rel_table, "d_rel_table")
    if tuple(d_query.shape) != (n, query_width(d)) or tuple(d_rel_table.shape) != tuple(rel_table.shape):
        raise ValueError("query_bwd: gradient shapes do not match")
    dx = torch.empty((n, d.width), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = load().bess_query_bwd(ctypes.byref(d), side, ent.base.data_ptr(), _idx(ent.idx, "entity idx"),
                                   rel_table.data_ptr(), rel_idx.data_ptr(), n, d_query.data_ptr(),
                                   dx.data_ptr(), d_rel_table.data_ptr(), _stream(dev))
    _check(rc, "bess_query_bwd")
    return dx


def query_width(d: ModelDesc) -> int:
    """Scalars per row of the query matrix the negative-scoring kernels take."""
    if d.scorer == AFFINE:  # [U | V | R]: one d-wide vector per entity part, plus the offset
        n_part = int(d.reserved[0])
        return (n_part + 1) * (d.width // n_part)
    if d.scorer == BOXE:  # [S | C | H] for each of the two candidate parts
        return 3 * d.width
    return int(d.width)


def _neg_operands(d: ModelDesc, query: torch.Tensor, neg: RowSource, n_idx: int):
    dev = _same_device([("query", query), ("negative rows", neg.base), ("negative idx", neg.idx)])
    _f32(query, "query")
    if query.dim() != 2 or query.shape[1] != query_width(d):
        raise ValueError(f"`query` must be [n_query, {query_width(d)}], got {tuple(query.shape)}")
    _rows(neg.base, "negative rows", d.width)
    if _dtype_code(neg.base) != d.dtype:
        raise TypeError("negative rows dtype does not match the model descriptor")
    if len(neg) != n_idx:
        raise ValueError(f"expected {n_idx} negative rows, got {len(neg)}")
    return dev


def _neg_idx_ptr(neg: RowSource, dev: torch.device) -> Tuple[int, Optional[torch.Tensor]]:
    """The per-triple kernel needs an explicit index array."""
    if neg.idx is not None:
        return _idx(neg.idx, "negative idx"), None
    ar = torch.arange(neg.base.shape[0], dtype=torch.int32, device=dev)
    return ar.data_ptr(), ar


def neg_score_pertriple_fwd(d: ModelDesc, query: torch.Tensor, neg: RowSource, n_neg: int,
                            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    nq = int(query.shape[0])
    dev = _neg_operands(d, query, neg, nq * n_neg)
    if out is None:
        out = torch.empty((nq, n_neg), dtype=torch.float32, device=dev)
    _f32(out, "out")
    if tuple(out.shape) != (nq, n_neg):
        raise ValueError("neg_score_pertriple_fwd: bad `out` shape")
    ip, keep = _neg_idx_ptr(neg, dev)
    with _on(dev), _Timed("bess_neg_score_pertriple_fwd", dev):
        rc = load().bess_neg_score_pertriple_fwd(ctypes.byref(d), query.data_ptr(), nq, neg.base.data_ptr(), ip,
                                                 n_neg, out.data_ptr(), n_neg, _stream(dev))
    _check(rc, "bess_neg_score_pertriple_fwd")
    del keep
    return out


def neg_score_pertriple_fwd_dq(d: ModelDesc, l: LossDesc, query: torch.Tensor, neg: RowSource, n_neg: int,
                               pos: Optional[torch.Tensor], weight: torch.Tensor, mask: Optional[torch.Tensor] = None,
                               defer: bool = False) -> Tuple[torch.Tensor, Any]:
    """Fused training forward: (scores [nq, n_neg], d loss / d query [nq, W]) in one pass over
    the negative rows.  Only valid when the loss is taken over exactly these scores; `mask` (bool
    [1 | nq, cols <= n_neg], False = masked out, over the last `cols` columns) is applied inside the pass.
    `defer`: the second item is (state_ml, state_acc) - the work items' partials, which `pertriple_tail`
    turns into d loss / d query where it uses it."""
    nq = int(query.shape[0])
    dev = _neg_operands(d, query, neg, nq * n_neg)
    _same_device([("pos", pos), ("weight", weight), ("query", query)])
    _f32(weight, "weight")
    if weight.numel() not in (1, nq):
        raise ValueError("triple weights must have 1 or n_query entries")
    if pos is not None:
        _f32(pos, "pos")
        if pos.numel() != nq:
            raise ValueError("`pos` must have n_query entries")
    items = ctypes.c_int32(0)
    _check(load().bess_neg_pertriple_items(ctypes.byref(d), nq, n_neg, ctypes.byref(items)), "bess_neg_pertriple_items")
    out = torch.empty((nq, n_neg), dtype=torch.float32, device=dev)
    dq = None if defer else torch.empty((nq, d.width), dtype=torch.float32, device=dev)
    st_ml = torch.empty((nq, items.value, 2), dtype=torch.float32, device=dev)
    st_acc = torch.empty((nq, items.value, d.width), dtype=torch.float32, device=dev)
    ip, keep = _neg_idx_ptr(neg, dev)
    mrows = mcols = 0
    if mask is not None:
        _same_device([("mask", mask), ("query", query)])
        if mask.dtype not in (torch.bool, torch.uint8) or mask.dim() != 2 or not mask.is_contiguous():
            raise ValueError("neg_score_pertriple_fwd_dq: `mask` must be a contiguous 2-D bool / uint8 tensor")
        mrows, mcols = int(mask.shape[0]), int(mask.shape[1])
    with _on(dev), _Timed("bess_neg_score_pertriple_fwd_dq", dev):
        rc = load().bess_neg_score_pertriple_fwd_dq_masked(
            ctypes.byref(d), ctypes.byref(l), query.data_ptr(), nq, neg.base.data_ptr(), ip, n_neg,
            pos.data_ptr() if pos is not None else 0, weight.data_ptr(), weight.numel(),
            mask.data_ptr() if mask is not None else 0, mrows, mcols, out.data_ptr(), n_neg,
            dq.data_ptr() if dq is not None else 0, st_ml.data_ptr(), st_acc.data_ptr(), _stream(dev))
    _check(rc, "bess_neg_score_pertriple_fwd_dq_masked")
    del keep
    return out, ((st_ml, st_acc) if defer else dq)


def pertriple_tail_supported(d: ModelDesc, n_neg: int) -> bool:
    return bool(load().bess_pertriple_tail_supported(ctypes.byref(d), int(n_neg)))


def pertriple_tail(d: ModelDesc, l: LossDesc, side: int, head: RowSource, tail: RowSource, rel_table: torch.Tensor,
                   rel_idx: torch.Tensor, partials: Tuple[torch.Tensor, torch.Tensor], pos: torch.Tensor, neg: torch.Tensor,
                   weight: torch.Tensor, d_rel_table: torch.Tensor, want_d_query: bool = False) -> Tuple[Any, ...]:
    """The per-triple remainder of a training step whose negatives went through the fused forward with `defer=True`
    (`bess_pertriple_tail`, one launch): returns (loss [], d_pos [n], d_neg [n, n_neg], d_head [n, W], d_tail [n, W])
    - and d loss / d query [n, W] as a sixth item with `want_d_query`; relation gradients are added into
    `d_rel_table`."""
    dev, n = _triple_operands(d, head, tail, rel_table, rel_idx)
    st_ml, st_acc = partials
    _same_device([("partials", st_ml), ("partials", st_acc), ("positive_score", pos), ("negative_score", neg),
                  ("triple_weight", weight), ("d_rel_table", d_rel_table), ("x", head.base)])
    for t, nm in ((st_ml, "partials"), (st_acc, "partials"), (pos, "positive_score"), (neg, "negative_score"),
                  (weight, "triple_weight"), (d_rel_table, "d_rel_table")):
        _f32(t, nm)
    N = int(neg.shape[1])
    items = int(st_ml.shape[1])
    if tuple(st_ml.shape) != (n, items, 2) or tuple(st_acc.shape) != (n, items, d.width) or pos.numel() != n \
            or neg.shape[0] != n or weight.numel() not in (1, n) or tuple(d_rel_table.shape) != tuple(rel_table.shape):
        raise ValueError("pertriple_tail: shapes do not match")
    # ONE allocation for the small pieces (every piece starts on a 16-byte boundary)
    n4 = (n + 3) // 4 * 4
    small = torch.empty((2 * n4 + 4,), dtype=torch.float32, device=dev)
    row_loss, dp, loss = small[:n], small[n4: n4 + n], small[2 * n4: 2 * n4 + 1]
    dn = torch.empty((n, N), dtype=torch.float32, device=dev)
    rows = torch.empty((3 if want_d_query else 2, n, d.width), dtype=torch.float32, device=dev)
    with _on(dev):
        stream = _stream(dev)
        counter = _counters(dev, stream, 1)
        rc = load().bess_pertriple_tail(
            ctypes.byref(d), ctypes.byref(l), side, head.base.data_ptr(), _idx(head.idx, "head_idx"),
            tail.base.data_ptr(), _idx(tail.idx, "tail_idx"), rel_table.data_ptr(), rel_idx.data_ptr(), n,
            st_ml.data_ptr(), st_acc.data_ptr(), items, pos.data_ptr(), neg.data_ptr(), N, int(neg.stride(0)),
            weight.data_ptr(), weight.numel(), row_loss.data_ptr(), loss.data_ptr(), dp.data_ptr(), dn.data_ptr(), N,
            rows[2].data_ptr() if want_d_query else 0, rows[0].data_ptr(), rows[1].data_ptr(), d_rel_table.data_ptr(),
            counter.data_ptr(), stream)
    _check(rc, "bess_pertriple_tail")
    res = (loss.reshape(()), dp, dn, rows[0], rows[1])
    return res + (rows[2],) if want_d_query else res


def row_fits_registers(d: ModelDesc) -> bool:
    """Do the per-triple kernels keep a whole row in a 16-lane group's registers (1024 f32 / 2048 f16 scalars at full
    vector width)?  Wider rows are scored in column windows, which the fused training forward cannot use."""
    W = int(d.width)
    vec = 4 if d.dtype == F32 else 8
    if W % vec:
        vec = 2 if (d.dtype == F16 and W % 2 == 0) else 1
    return W <= 256 * vec


def neg_score_pertriple_fwd_partials(d: ModelDesc, l: LossDesc, query: torch.Tensor, neg: RowSource, n_neg: int
                                     ) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
    """Training forward on a shard that holds only a part of each query's negatives (ScoreMoving): scores
    [nq, n_neg] and the partials (state_ml [nq, items, 2], state_acc [nq, items, W]) that
    `combine_dq_partials` turns into this shard's share of d loss / d query."""
    nq = int(query.shape[0])
    dev = _neg_operands(d, query, neg, nq * n_neg)
    items = ctypes.c_int32(0)
    _check(load().bess_neg_pertriple_items(ctypes.byref(d), nq, n_neg, ctypes.byref(items)), "bess_neg_pertriple_items")
    out = torch.empty((nq, n_neg), dtype=torch.float32, device=dev)
    st_ml = torch.empty((nq, items.value, 2), dtype=torch.float32, device=dev)
    st_acc = torch.empty((nq, items.value, d.width), dtype=torch.float32, device=dev)
    ip, keep = _neg_idx_ptr(neg, dev)
    with _on(dev), _Timed("bess_neg_score_pertriple_fwd_partials", dev):
        rc = load().bess_neg_score_pertriple_fwd_partials(
            ctypes.byref(d), ctypes.byref(l), query.data_ptr(), nq, neg.base.data_ptr(), ip, n_neg, out.data_ptr(),
            n_neg, st_ml.data_ptr(), st_acc.data_ptr(), _stream(dev))
    _check(rc, "bess_neg_score_pertriple_fwd_partials")
    del keep
    return out, (st_ml, st_acc)


def combine_dq_partials(state: Tuple[torch.Tensor, torch.Tensor], norm: torch.Tensor) -> torch.Tensor:
    """d_query [nq, W] of this shard's negatives from its partials and the per-query (m, L / C) of the whole
    softmax, `norm` [nq, 2] f32 (see include/besskge_hip.h)."""
    st_ml, st_acc = state
    nq, items, W = (int(x) for x in st_acc.shape)
    dev = _same_device([("state_ml", st_ml), ("state_acc", st_acc), ("norm", norm)])
    _f32(norm, "norm")
    if tuple(norm.shape) != (nq, 2) or not norm.is_contiguous():
        raise ValueError("combine_dq_partials: `norm` must be a contiguous [n_query, 2] tensor")
    dq = torch.empty((nq, W), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = load().bess_combine_dq_partials(st_ml.data_ptr(), st_acc.data_ptr(), nq, items, W, norm.data_ptr(),
                                             dq.data_ptr(), _stream(dev))
    _check(rc, "bess_combine_dq_partials")
    return dq


def neg_score_pertriple_bwd(d: ModelDesc, query: torch.Tensor, neg: RowSource, n_neg: int,
                            d_out: torch.Tensor, want_d_neg: bool = True, want_d_query: bool = True,
                            d_neg_rows: Optional[torch.Tensor] = None
                            ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Returns (d_query [nq, W] or None, d_neg [nq*n_neg, W] or None).  `want_d_query=False` (TransE / RotatE /
    DistMult / ComplEx; d_query came out of the fused forward): DistMult / ComplEx then do not read the candidate
    rows at all - their d_neg rows are coefficient x query.  `d_neg_rows` ([rows of neg.base, W] f32; the four
    native scorers): the gradient of reference k is STORED at row neg.idx[k] of it (BESS_FLAG_DNEG_BY_ROW) - only for
    lists that name a row at most once; d_neg is not returned then."""
    nq = int(query.shape[0])
    dev = _neg_operands(d, query, neg, nq * n_neg)
    _same_device([("d_out", d_out), ("query", query)])
    _f32(d_out, "d_out")
    if tuple(d_out.shape) != (nq, n_neg):
        raise ValueError("neg_score_pertriple_bwd: bad `d_out` shape")
    if not want_d_query and (int(d.scorer) > COMPLEX or not want_d_neg):
        raise ValueError("neg_score_pertriple_bwd: want_d_query=False is for the four native scorers, with d_neg")
    dq = torch.empty((nq, query_width(d)), dtype=torch.float32, device=dev) if want_d_query else None
    dn = torch.empty((nq * n_neg, d.width), dtype=torch.float32, device=dev) if want_d_neg else None
    if d_neg_rows is not None:
        if (int(d.scorer) > COMPLEX or not want_d_neg or d_neg_rows.dtype != torch.float32 or not d_neg_rows.is_contiguous()
                or tuple(d_neg_rows.shape) != (int(neg.base.shape[0]), int(d.width)) or d_neg_rows.device != dev):
            raise ValueError("neg_score_pertriple_bwd: `d_neg_rows` must be a contiguous f32 [rows of neg.base, W] "
                             "tensor (native scorers)")
        d = copy_desc(d)
        d.reserved[0] |= FLAG_DNEG_BY_ROW
        dn = d_neg_rows
    ip, keep = _neg_idx_ptr(neg, dev)
    with _on(dev), _Timed("bess_neg_score_pertriple_bwd", dev):
        rc = load().bess_neg_score_pertriple_bwd(ctypes.byref(d), query.data_ptr(), nq, neg.base.data_ptr(), ip,
                                                 n_neg, d_out.data_ptr(), n_neg, dq.data_ptr() if want_d_query else 0,
                                                 dn.data_ptr() if want_d_neg else 0, _stream(dev))
    _check(rc, "bess_neg_score_pertriple_bwd")
    del keep
    return dq, (None if d_neg_rows is not None else dn)


def normalize_rows(neg: RowSource, n_part: int, normalize: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """Gather + convert to f32 + L2-normalise every part of the rows: (hat [n, W] f32, inv [n, n_part])."""
    dev = _same_device([("rows", neg.base), ("idx", neg.idx)])
    W = int(neg.base.shape[1])
    _rows(neg.base, "rows", W)
    n = len(neg)
    hat = torch.empty((n, W), dtype=torch.float32, device=dev)
    inv = torch.empty((n, n_part), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = load().bess_normalize_rows(_dtype_code(neg.base), neg.base.data_ptr(), _idx(neg.idx, "idx"), n, W, n_part,
                                        int(normalize), hat.data_ptr(), inv.data_ptr(), _stream(dev))
    _check(rc, "bess_normalize_rows")
    return hat, inv


def normalize_rows_bwd(hat: torch.Tensor, inv: torch.Tensor, d_hat: torch.Tensor) -> torch.Tensor:
    dev = _same_device([("hat", hat), ("inv", inv), ("d_hat", d_hat)])
    for t, name in ((hat, "hat"), (inv, "inv"), (d_hat, "d_hat")):
        _f32(t, name)
    if hat.shape != d_hat.shape or inv.shape[0] != hat.shape[0]:
        raise ValueError("normalize_rows_bwd: shape mismatch")
    out = torch.empty_like(hat)
    with _on(dev):
        rc = load().bess_normalize_rows_bwd(hat.data_ptr(), inv.data_ptr(), d_hat.data_ptr(), hat.shape[0],
                                            hat.shape[1], inv.shape[1], out.data_ptr(), _stream(dev))
    _check(rc, "bess_normalize_rows_bwd")
    return out


def _affine_candidates(d: ModelDesc, neg: RowSource) -> Tuple[RowSource, torch.Tensor, torch.Tensor]:
    """The shared kernels of the affine scorers work on dense, normalised f32 candidates."""
    hat, inv = normalize_rows(neg, int(d.reserved[0]), bool(d.reserved[1] & 1))
    return RowSource(hat), hat, inv


def neg_score_shared_fwd(d: ModelDesc, query: torch.Tensor, neg: RowSource, pad_ld: bool = False,
                         kill: Optional[Tuple[int, bool, int, Optional[torch.Tensor]]] = None) -> torch.Tensor:
    """Scores [nq, n_neg].  `pad_ld`: rows of the result are 16-B aligned (leading dimension rounded
    up to 4 floats; the result is then a column slice of the buffer) - what `topk_update` streams fastest.
    `kill` = (diag_step, ht, ppp, mask [rows, cols] bool | None): K7 applied with the scores
    (`mask_scores` semantics; in the scoring kernel's epilogue where it has one)."""
    nq, n_neg = int(query.shape[0]), len(neg)
    dev = _neg_operands(d, query, neg, n_neg)
    if d.scorer == AFFINE:
        neg, _, _ = _affine_candidates(d, neg)
    ld = (n_neg + 3) // 4 * 4 if pad_ld else n_neg
    out = torch.empty((nq, ld), dtype=torch.float32, device=dev)
    lib = load()
    # scratch of the split-fp16 matrix-core path (0 for the other scorers / small shapes); from
    # torch's caching allocator, so it is stream-ordered and safe under graph capture
    ws_bytes = int(lib.bess_neg_score_shared_workspace(ctypes.byref(d), nq, n_neg))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes > 0 else None
    kd = None
    if kill is not None:
        diag_step, ht, ppp, mask = kill
        kd = KillDesc()
        kd.diag_step, kd.ht, kd.ppp = int(diag_step), int(bool(ht)), int(ppp)
        if mask is not None:
            _same_device([("negative_mask", mask), ("query", query)])
            if mask.dtype != torch.bool or mask.dim() != 2 or not mask.is_contiguous():
                raise ValueError("negative_mask must be a contiguous 2-D bool tensor")
            kd.mask, kd.mask_rows, kd.mask_cols = mask.data_ptr(), int(mask.shape[0]), int(mask.shape[1])
    with _on(dev), _Timed("bess_neg_score_shared_fwd", dev):
        if kd is None:
            rc = lib.bess_neg_score_shared_fwd_ws(ctypes.byref(d), query.data_ptr(), nq, neg.base.data_ptr(),
                                                  _idx(neg.idx, "negative idx"), n_neg, out.data_ptr(), ld,
                                                  ws.data_ptr() if ws is not None else None, ws_bytes, _stream(dev))
        else:
            rc = lib.bess_neg_score_shared_fwd_masked(ctypes.byref(d), query.data_ptr(), nq, neg.base.data_ptr(),
                                                      _idx(neg.idx, "negative idx"), n_neg, out.data_ptr(), ld,
                                                      ctypes.byref(kd), ws.data_ptr() if ws is not None else None,
                                                      ws_bytes, _stream(dev))
    _check(rc, "bess_neg_score_shared_fwd")
    return out if ld == n_neg else out[:, :n_neg]


def neg_score_shared_fwd_pruned(d: ModelDesc, query: torch.Tensor, neg: RowSource, thr: torch.Tensor
                                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(scores [nq, n_neg] with 16-B aligned rows, flags [nq, n_blocks] uint8) for the top-k passes: a row's block
    of 64 consecutive candidates is written only when one of its scores is above `thr[row]` (f32 [nq]: the row's
    current k-th best), `flags` marks the blocks that were - the rest of `scores` is uninitialised memory, which
    `topk_update(..., flags=flags)` never reads.  (`bess_neg_score_shared_fwd_pruned`: kernels without a pruning
    epilogue write and flag everything.)"""
    nq, n_neg = int(query.shape[0]), len(neg)
    dev = _neg_operands(d, query, neg, n_neg)
    _same_device([("query", query), ("thr", thr)])
    _f32(thr, "thr")
    if thr.numel() != nq:
        raise ValueError("neg_score_shared_fwd_pruned: one threshold per query")
    if d.scorer == AFFINE:
        neg, _, _ = _affine_candidates(d, neg)
    ld = (n_neg + 3) // 4 * 4
    out = torch.empty((nq, ld), dtype=torch.float32, device=dev)
    ldf = ((n_neg + 63) // 64 + 3) // 4 * 4
    flags = torch.empty((nq, ldf), dtype=torch.uint8, device=dev)
    lib = load()
    ws_bytes = int(lib.bess_neg_score_shared_workspace(ctypes.byref(d), nq, n_neg))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes > 0 else None
    with _on(dev), _Timed("bess_neg_score_shared_fwd", dev):
        rc = lib.bess_neg_score_shared_fwd_pruned(ctypes.byref(d), query.data_ptr(), nq, neg.base.data_ptr(),
                                                  _idx(neg.idx, "negative idx"), n_neg, out.data_ptr(), ld,
                                                  thr.data_ptr(), flags.data_ptr(), ldf,
                                                  ws.data_ptr() if ws is not None else None, ws_bytes, _stream(dev))
    _check(rc, "bess_neg_score_shared_fwd_pruned")
    return (out if ld == n_neg else out[:, :n_neg]), flags


def neg_score_shared_pairs(d: ModelDesc, query: torch.Tensor, neg: RowSource, like_n_query: int,
                           like_n_neg: int) -> torch.Tensor:
    """out[i] = score(query[i], candidate i of `neg`) in the arithmetic of the all-entity pass over a
    (like_n_query x like_n_neg) problem (`bess_neg_score_shared_fwd_pairs`): what `neg_score_shared_counts` compares
    against, to the last bit."""
    n = int(query.shape[0])
    if len(neg) != n or neg.idx is None:
        raise ValueError("neg_score_shared_pairs: one indexed candidate per query")
    dev = _neg_operands(d, query, neg, n)
    out = torch.empty((n,), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    lib = load()
    ws_bytes = int(lib.bess_neg_score_shared_fwd_pairs_workspace(ctypes.byref(d), int(like_n_query), int(like_n_neg)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with _on(dev), _Timed("bess_neg_score_shared_fwd_pairs", dev):
        rc = lib.bess_neg_score_shared_fwd_pairs(ctypes.byref(d), query.data_ptr(), neg.base.data_ptr(),
                                                 _idx(neg.idx, "negative idx"), n, int(like_n_query), int(like_n_neg),
                                                 out.data_ptr(), ws.data_ptr(), ws_bytes, _stream(dev))
    _check(rc, "bess_neg_score_shared_fwd_pairs")
    return out


def neg_score_shared_counts(d: ModelDesc, query: torch.Tensor, neg: RowSource, thr: torch.Tensor,
                            excl: torch.Tensor, counts: Optional[torch.Tensor] = None,
                            round_f16: bool = False) -> torch.Tensor:
    """Ranks without the score matrix (`bess_neg_score_shared_fwd_counts`): adds to `counts` [nq, 2] int32 (made
    and cleared when None) the number of candidates scoring above / exactly `thr[q]` (f32 [nq]: the score of the
    row's true completion), leaving out candidate `excl[q]` (int32 [nq]: its position in `neg`, -1: not among
    them).  `round_f16`: scores are rounded to fp16 before they are compared (the ranking of a half-precision
    model's scores).  Negative counts afterwards: an operand was outside the fp16 range of the matrix-core product - score
    that batch through `neg_score_shared_fwd` instead."""
    nq, n_neg = int(query.shape[0]), len(neg)
    dev = _neg_operands(d, query, neg, n_neg)
    _same_device([("query", query), ("thr", thr), ("excl", excl), ("counts", counts)])
    _f32(thr, "thr")
    if thr.numel() != nq or excl.numel() != nq or excl.dtype != torch.int32:
        raise ValueError("neg_score_shared_counts: one threshold and one excluded position (int32) per query")
    if not thr.is_contiguous() or not excl.is_contiguous():
        raise ValueError("neg_score_shared_counts: thr / excl must be contiguous")
    if counts is None:
        counts = torch.zeros((nq, 2), dtype=torch.int32, device=dev)
    elif tuple(counts.shape) != (nq, 2) or counts.dtype != torch.int32 or not counts.is_contiguous():
        raise ValueError("neg_score_shared_counts: counts must be a contiguous [nq, 2] int32 tensor")
    if d.scorer == AFFINE:
        neg, _, _ = _affine_candidates(d, neg)
    lib = load()
    ws_bytes = int(lib.bess_neg_score_shared_fwd_counts_workspace(ctypes.byref(d), nq, n_neg))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes > 0 else None
    with _on(dev), _Timed("bess_neg_score_shared_fwd_counts", dev):
        rc = lib.bess_neg_score_shared_fwd_counts(ctypes.byref(d), query.data_ptr(), nq, neg.base.data_ptr(),
                                                  _idx(neg.idx, "negative idx"), n_neg, thr.data_ptr(),
                                                  excl.data_ptr(), counts.data_ptr(), int(bool(round_f16)),
                                                  ws.data_ptr() if ws is not None else None, ws_bytes, _stream(dev))
    _check(rc, "bess_neg_score_shared_fwd_counts")
    return counts


def shared_bwd_buffer(d: ModelDesc, nq: int, n_neg: int, device: torch.device) -> Optional[torch.Tensor]:
    """The [nq + n_neg, W] f32 allocation that `neg_score_shared_bwd(..., prezeroed=buf)` takes for d_query and
    d_neg when a caller clears it itself (with the other fills of its step: `step_prologue`); None for scorers
    whose query and candidate rows differ in width."""
    if query_width(d) != d.width or d.scorer == AFFINE:
        return None
    return torch.empty((nq + n_neg, d.width), dtype=torch.float32, device=device)


def shared_bwd_parts_plan(d: ModelDesc, nq: int, n_neg: int) -> Tuple[int, int]:
    """(n_dq_parts, n_dneg_parts) of `neg_score_shared_bwd_parts` for this scorer and shape; (0, 0): no such form."""
    a, b = ctypes.c_int32(0), ctypes.c_int32(0)
    _check(load().bess_neg_score_shared_bwd_parts_plan(ctypes.byref(d), int(nq), int(n_neg), ctypes.byref(a), ctypes.byref(b)),
           "bess_neg_score_shared_bwd_parts_plan")
    return a.value, b.value


def neg_score_shared_bwd_parts(d: ModelDesc, query: torch.Tensor, neg: RowSource, d_out: torch.Tensor
                               ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dq_parts [n_dq, nq, W], dneg_parts [n_de, n_neg, W]): the two products of the shared-negative backward as
    partial sums, written with plain stores - no atomics, nothing to clear (`bess_neg_score_shared_bwd_parts`);
    `query_triple_bwd_parts` adds them up where it consumes them."""
    nq, n_neg = int(query.shape[0]), len(neg)
    dev = _neg_operands(d, query, neg, n_neg)
    _same_device([("d_out", d_out), ("query", query)])
    _f32(d_out, "d_out")
    if tuple(d_out.shape) != (nq, n_neg):
        raise ValueError("neg_score_shared_bwd_parts: bad score-gradient shape")
    n_dq, n_de = shared_bwd_parts_plan(d, nq, n_neg)
    if n_dq == 0:
        raise RuntimeError("neg_score_shared_bwd_parts: this scorer / shape has no partial-sum form")
    W = int(d.width)
    buf = torch.empty((n_dq * nq + n_de * n_neg, W), dtype=torch.float32, device=dev)
    dqp, dep = buf[: n_dq * nq].view(n_dq, nq, W), buf[n_dq * nq:].view(n_de, n_neg, W)
    with _on(dev), _Timed("bess_neg_score_shared_bwd", dev):
        rc = load().bess_neg_score_shared_bwd_parts(ctypes.byref(d), query.data_ptr(), nq, neg.base.data_ptr(),
                                                    _idx(neg.idx, "negative idx"), n_neg, d_out.data_ptr(), n_neg,
                                                    dqp.data_ptr(), dep.data_ptr(), _stream(dev))
    _check(rc, "bess_neg_score_shared_bwd_parts")
    return dqp, dep


def query_triple_bwd_parts(d: ModelDesc, side: int, head: RowSource, tail: RowSource, rel_table: torch.Tensor,
                           rel_idx: torch.Tensor, d_out: torch.Tensor, dq_parts: torch.Tensor, dneg_parts: torch.Tensor,
                           neg_idx: torch.Tensor, rows_acc: Tuple[torch.Tensor, torch.Tensor, torch.Tensor],
                           d_rel_table: torch.Tensor) -> None:
    """`query_triple_bwd` by row, fed by the partial sums of `neg_score_shared_bwd_parts`: the heads' and tails' gradient
    rows are added into `rows_acc[0]` / `rows_acc[1]` at the triples' row ids, the candidates' summed rows into
    `rows_acc[2]` at theirs.  rows_acc = accumulators over the row
    spaces of (head.base, tail.base, the candidates' table)."""
    dev, n = _triple_operands(d, head, tail, rel_table, rel_idx)
    acc_h, acc_t, acc_n = rows_acc
    _same_device([("d_out", d_out), ("dq_parts", dq_parts), ("dneg_parts", dneg_parts), ("neg_idx", neg_idx),
                  ("d_rel_table", d_rel_table), ("acc_h", acc_h), ("acc_t", acc_t), ("acc_n", acc_n)])
    for t, nm in ((d_out, "d_out"), (dq_parts, "dq_parts"), (dneg_parts, "dneg_parts"), (d_rel_table, "d_rel_table"),
                  (acc_h, "acc_h"), (acc_t, "acc_t"), (acc_n, "acc_n")):
        _f32(t, nm)
    W = int(d.width)
    n_neg = int(neg_idx.numel())
    if dq_parts.dim() != 3 or tuple(dq_parts.shape[1:]) != (n, W) or dneg_parts.dim() != 3 \
            or tuple(dneg_parts.shape[1:]) != (n_neg, W) or d_out.numel() != n or head.idx is None or tail.idx is None \
            or tuple(d_rel_table.shape) != tuple(rel_table.shape):
        raise ValueError("query_triple_bwd_parts: shapes do not match")
    if tuple(acc_h.shape) != (int(head.base.shape[0]), W) or tuple(acc_t.shape) != (int(tail.base.shape[0]), W) \
            or acc_n.dim() != 2 or acc_n.shape[1] != W:
        raise ValueError("query_triple_bwd_parts: accumulators must cover the row spaces of the tables")
    with _on(dev):
        rc = load().bess_query_triple_bwd_parts(
            ctypes.byref(d), side, head.base.data_ptr(), _idx(head.idx, "head_idx"), tail.base.data_ptr(),
            _idx(tail.idx, "tail_idx"), rel_table.data_ptr(), rel_idx.data_ptr(), n, d_out.data_ptr(),
            dq_parts.data_ptr(), int(dq_parts.shape[0]), dneg_parts.data_ptr(), int(dneg_parts.shape[0]), n_neg,
            _idx(neg_idx, "neg_idx", n_neg), acc_h.data_ptr(), acc_t.data_ptr(), acc_n.data_ptr(), d_rel_table.data_ptr(),
            _stream(dev))
    _check(rc, "bess_query_triple_bwd_parts")


def neg_score_shared_bwd(d: ModelDesc, query: torch.Tensor, neg: RowSource, out: torch.Tensor,
                         d_out: torch.Tensor, prezeroed: Optional[torch.Tensor] = None
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (d_query [nq, W], d_neg [n_neg, W]).  `prezeroed`: a `shared_bwd_buffer` the caller has
    already cleared on this stream - the call then writes into it and does not clear anything."""
    nq, n_neg = int(query.shape[0]), len(neg)
    dev = _neg_operands(d, query, neg, n_neg)
    _same_device([("d_out", d_out), ("out", out), ("query", query)])
    _f32(d_out, "d_out")
    _f32(out, "out")
    if tuple(d_out.shape) != (nq, n_neg) or tuple(out.shape) != (nq, n_neg):
        raise ValueError("neg_score_shared_bwd: bad score shapes")
    hat = inv = None
    if d.scorer == AFFINE:
        neg, hat, inv = _affine_candidates(d, neg)
    qw = query_width(d)
    if prezeroed is not None:
        if tuple(prezeroed.shape) != (nq + n_neg, d.width) or prezeroed.dtype != torch.float32 or qw != d.width:
            raise ValueError("neg_score_shared_bwd: `prezeroed` must come from shared_bwd_buffer()")
        dq, dn = prezeroed[:nq], prezeroed[nq:]
        d = copy_desc(d)
        d.reserved[0] |= FLAG_PREZEROED
    elif qw == d.width:  # one allocation: the library then zeroes both partial-sum targets with one memset
        both = torch.empty(((nq + n_neg), qw), dtype=torch.float32, device=dev)
        dq, dn = both[:nq], both[nq:]
    else:
        dq = torch.empty((nq, qw), dtype=torch.float32, device=dev)
        dn = torch.empty((n_neg, d.width), dtype=torch.float32, device=dev)
    lib = load()
    ws_bytes = int(lib.bess_neg_score_shared_bwd_workspace(ctypes.byref(d), nq, n_neg))  # see neg_score_shared_fwd
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes > 0 else None
    with _on(dev), _Timed("bess_neg_score_shared_bwd", dev):
        rc = lib.bess_neg_score_shared_bwd_ws(ctypes.byref(d), query.data_ptr(), nq, neg.base.data_ptr(),
                                              _idx(neg.idx, "negative idx"), n_neg, out.data_ptr(), n_neg,
                                              d_out.data_ptr(), n_neg, dq.data_ptr(), dn.data_ptr(),
                                              ws.data_ptr() if ws is not None else None, ws_bytes, _stream(dev))
    _check(rc, "bess_neg_score_shared_bwd")
    if hat is not None and (d.reserved[1] & 1):
        dn = normalize_rows_bwd(hat, inv, dn)
    return dq, dn


def mask_scores(neg: torch.Tensor, diag_step: int, ht: bool, ppp: int, mask: Optional[torch.Tensor]) -> None:
    """In place K7 (see include/besskge_hip.h)."""
    dev = _same_device([("negative_score", neg), ("negative_mask", mask)])
    _f32(neg, "negative_score")
    if neg.dim() != 2:
        raise ValueError("negative_score must be 2-D")
    S, N = int(neg.shape[0]), int(neg.shape[1])
    mp, mrows, mcols = 0, 0, 0
    if mask is not None:
        if mask.dtype != torch.bool or mask.dim() != 2 or not mask.is_contiguous():
            raise ValueError("negative_mask must be a contiguous 2-D bool tensor")
        mp, mrows, mcols = mask.data_ptr(), int(mask.shape[0]), int(mask.shape[1])
    with _on(dev):
        rc = load().bess_mask_scores(neg.data_ptr(), S, N, N, diag_step, int(ht), ppp, mp, mrows, mcols, _stream(dev))
    _check(rc, "bess_mask_scores")


_loss_counters: dict = {}  # (device, raw stream) -> int32 [TICKET_INTS], zero between calls


def loss_fwd_bwd(l: LossDesc, pos: torch.Tensor, neg: torch.Tensor, weight: torch.Tensor, want_grad: bool,
                 want_norm: bool = False) -> Tuple[Any, ...]:
    """Returns (loss [] f32, d_pos [S] | None, d_neg [S, N] | None) - and, with `want_norm`, a fourth item:
    (m, L / C) of every row's softmax, [S, 2] (what `combine_dq_partials` takes; include/besskge_hip.h)."""
    dev = _same_device([("positive_score", pos), ("negative_score", neg), ("triple_weight", weight)])
    for t, nm in ((pos, "positive_score"), (neg, "negative_score"), (weight, "triple_weight")):
        _f32(t, nm)
    S, N = int(neg.shape[0]), int(neg.shape[1])
    if pos.numel() != S or weight.numel() not in (1, S):
        raise ValueError("loss: shapes of positive_score / triple_weight do not match negative_score")
    row_loss = torch.empty((S,), dtype=torch.float32, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    dp = torch.empty((S,), dtype=torch.float32, device=dev) if want_grad else None
    dn = torch.empty((S, N), dtype=torch.float32, device=dev) if want_grad else None
    norm = torch.empty((S, 2), dtype=torch.float32, device=dev) if want_norm else None
    with _on(dev):
        stream = _stream(dev)
        # (one launch: its last workgroup sums the row terms; the counter it needs is zero between calls)
        capturing = torch.cuda.is_current_stream_capturing()
        key = (dev, "capture") if capturing else (dev, stream)
        counter = _loss_counters.get(key)
        if counter is None:
            counter = _loss_counters[key] = torch.zeros((TICKET_INTS,), dtype=torch.int32, device=dev)
            if not capturing and (dev, "capture") not in _loss_counters:
                # the counter of recorded steps exists before any recording starts (a tensor made while a
                # stream is capturing would be cleared by a fill node at every replay: one more dispatch)
                _loss_counters[(dev, "capture")] = torch.zeros((TICKET_INTS,), dtype=torch.int32, device=dev)
        rc = load().bess_loss_fwd_bwd_one_launch(ctypes.byref(l), pos.data_ptr(), neg.data_ptr(), S, N, N,
                                                 weight.data_ptr(), weight.numel(), row_loss.data_ptr(), loss.data_ptr(),
                                                 dp.data_ptr() if want_grad else 0, dn.data_ptr() if want_grad else 0, N,
                                                 norm.data_ptr() if want_norm else 0, counter.data_ptr(), stream)
    _check(rc, "bess_loss_fwd_bwd_one_launch")
    if want_norm:
        return loss.reshape(()), dp, dn, norm
    return loss.reshape(()), dp, dn


_tail_counters: dict = {}  # (device, raw stream | "capture", slots) -> int32 [slots], zero between calls


def _counters(dev: torch.device, stream: int, slots: int) -> torch.Tensor:
    """Zeroed int32 counters that kernels of one stream leave zero again (one array per stream; recorded steps
    share one that exists before any recording starts - a tensor made while a stream is capturing would be
    cleared by a fill node at every replay)."""
    slots = max(1024, 1 << (slots - 1).bit_length())  # (>= BESS_TICKET_INTS)
    capturing = torch.cuda.is_current_stream_capturing()
    key = (dev, "capture" if capturing else stream, slots)
    c = _tail_counters.get(key)
    if c is None:
        if capturing:
            raise RuntimeError("besskge: the counters of a recorded step must exist before the recording starts "
                               "(run the step once eagerly first - Runner does)")
        c = _tail_counters[key] = torch.zeros((slots,), dtype=torch.int32, device=dev)
        _tail_counters.setdefault((dev, "capture", slots), torch.zeros((slots,), dtype=torch.int32, device=dev))
    return c


def neg_score_shared_fwd_loss(d: ModelDesc, l: LossDesc, query: torch.Tensor, neg: RowSource, pos: torch.Tensor,
                              weight: torch.Tensor, kill: Optional[Tuple[int, bool, int, Optional[torch.Tensor]]] = None
                              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """K4 + K7 + K8 of a training step with shared negatives (`bess_neg_score_shared_fwd_loss`): scores [nq, n_neg]
    (K7 applied), loss [], d_pos [nq], d_neg [nq, n_neg].  One launch where the scoring kernel can finish the loss
    rows itself (packed L1 kernel, rows of up to 1024 scores), else the scoring launch + the loss launch(es)."""
    nq, n_neg = int(query.shape[0]), len(neg)
    dev = _neg_operands(d, query, neg, n_neg)
    _same_device([("query", query), ("positive_score", pos), ("triple_weight", weight)])
    _f32(pos, "positive_score")
    _f32(weight, "triple_weight")
    if pos.numel() != nq or weight.numel() not in (1, nq):
        raise ValueError("neg_score_shared_fwd_loss: shapes of positive_score / triple_weight do not match the queries")
    if d.scorer == AFFINE:
        neg, _, _ = _affine_candidates(d, neg)
    # ONE allocation: scores, score gradients, row terms, d_pos, loss (every piece starts on a 16-byte boundary;
    # dense rows: the backward kernels take contiguous [nq, n_neg] matrices)
    ld = n_neg
    mat = (nq * ld + 3) // 4 * 4
    nq4 = (nq + 3) // 4 * 4
    buf = torch.empty((2 * mat + 2 * nq4 + 4,), dtype=torch.float32, device=dev)
    out = buf[: nq * ld].view(nq, ld)
    dn = buf[mat: mat + nq * ld].view(nq, ld)
    row_loss = buf[2 * mat: 2 * mat + nq]
    dp = buf[2 * mat + nq4: 2 * mat + nq4 + nq]
    loss = buf[2 * mat + 2 * nq4: 2 * mat + 2 * nq4 + 1]
    lib = load()
    ws_bytes = int(lib.bess_neg_score_shared_workspace(ctypes.byref(d), nq, n_neg))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes > 0 else None
    kd = None
    if kill is not None:
        diag_step, ht, ppp, mask = kill
        kd = KillDesc()
        kd.diag_step, kd.ht, kd.ppp = int(diag_step), int(bool(ht)), int(ppp)
        if mask is not None:
            _same_device([("negative_mask", mask), ("query", query)])
            if mask.dtype != torch.bool or mask.dim() != 2 or not mask.is_contiguous():
                raise ValueError("negative_mask must be a contiguous 2-D bool tensor")
            kd.mask, kd.mask_rows, kd.mask_cols = mask.data_ptr(), int(mask.shape[0]), int(mask.shape[1])
    with _on(dev), _Timed("bess_neg_score_shared_fwd_loss", dev):
        stream = _stream(dev)
        counters = _counters(dev, stream, (nq + 15) // 16 + 1)
        rc = lib.bess_neg_score_shared_fwd_loss(
            ctypes.byref(d), query.data_ptr(), nq, neg.base.data_ptr(), _idx(neg.idx, "negative idx"), n_neg,
            out.data_ptr(), ld, ctypes.byref(kd) if kd is not None else None, ctypes.byref(l), pos.data_ptr(),
            weight.data_ptr(), weight.numel(), row_loss.data_ptr(), loss.data_ptr(), dp.data_ptr(), dn.data_ptr(), ld,
            counters.data_ptr(), ws.data_ptr() if ws is not None else None, ws_bytes, stream)
    _check(rc, "bess_neg_score_shared_fwd_loss")
    return out, loss.reshape(()), dp, dn


def scatter_add_rows(dst: torch.Tensor, idx: torch.Tensor, src: torch.Tensor, scale: float = 1.0) -> None:
    dev = _same_device([("dst", dst), ("idx", idx), ("src", src)])
    _f32(dst, "dst")
    _f32(src, "src")
    W = int(dst.shape[1])
    _rows(dst, "dst", W)
    _rows(src, "src", W)
    n = int(src.shape[0])
    ip = _idx(idx, "idx", n)
    with _on(dev):
        rc = load().bess_scatter_add_rows(dst.data_ptr(), W, ip, src.data_ptr(), n, scale, _stream(dev))
    _check(rc, "bess_scatter_add_rows")


def sparse_sgd(table: torch.Tensor, idx: torch.Tensor, grad: torch.Tensor, lr: float) -> None:
    dev = _same_device([("table", table), ("idx", idx), ("grad", grad)])
    W = int(table.shape[1])
    _rows(table, "table", W)
    _f32(grad, "grad")
    _rows(grad, "grad", W)
    n = int(grad.shape[0])
    ip = _idx(idx, "idx", n)
    with _on(dev), _Timed("bess_sparse_sgd", dev):
        rc = load().bess_sparse_sgd(_dtype_code(table), W, table.data_ptr(), ip, grad.data_ptr(), n, lr, _stream(dev))
    _check(rc, "bess_sparse_sgd")


def sparse_sgd_lists(table: torch.Tensor, lists: Sequence[Tuple[torch.Tensor, torch.Tensor]], lr: float,
                     axpy: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None) -> None:
    """`sparse_sgd` for several (row ids, gradient rows) lists in one launch.  `axpy` = (table2, grad2, alpha):
    `table2 += alpha * grad2` (dense, same dtype as `table`) by spare workgroups of the same launch."""
    if not 1 <= len(lists) <= MAX_ROW_LISTS:
        raise ValueError(f"sparse_sgd_lists: {len(lists)} lists (1 .. {MAX_ROW_LISTS})")
    dev = _same_device([("table", table)] + [(f"idx[{i}]", x) for i, (x, _) in enumerate(lists)]
                       + [(f"grad[{i}]", g) for i, (_, g) in enumerate(lists)])
    W = int(table.shape[1])
    _rows(table, "table", W)
    for i, (x, g) in enumerate(lists):
        _f32(g, f"grad[{i}]")
        _rows(g, f"grad[{i}]", W)
        _idx(x, f"idx[{i}]", int(g.shape[0]))
    ip = (_vp * len(lists))(*[x.data_ptr() for x, _ in lists])
    gp = (_vp * len(lists))(*[g.data_ptr() for _, g in lists])
    rows = (_i64 * len(lists))(*[int(g.shape[0]) for _, g in lists])
    x_table = x_grad = None
    x_n, x_alpha = 0, 0.0
    if axpy is not None:
        x_table, x_grad, x_alpha = axpy
        _same_device([("table", table), ("axpy table", x_table), ("axpy grad", x_grad)])
        _f32(x_grad, "axpy grad")
        if x_table.dtype != table.dtype or not x_table.is_contiguous() or x_grad.numel() != x_table.numel():
            raise ValueError("sparse_sgd_lists: the axpy table must be contiguous, of the table's dtype, and match its gradient")
        x_n = int(x_table.numel())
    with _on(dev), _Timed("bess_sparse_sgd_lists", dev):
        rc = load().bess_sparse_sgd_lists_axpy(_dtype_code(table), W, table.data_ptr(), len(lists), ip, gp, rows, lr,
                                               x_table.data_ptr() if x_n else None, x_grad.data_ptr() if x_n else None,
                                               x_n, float(x_alpha), _stream(dev))
    _check(rc, "bess_sparse_sgd_lists")


def dense_sgd(table: torch.Tensor, grad: torch.Tensor, lr: float) -> None:
    dev = _same_device([("table", table), ("grad", grad)])
    _f32(grad, "grad")
    if not table.is_contiguous() or table.numel() != grad.numel():
        raise ValueError("dense_sgd: table / grad mismatch")
    with _on(dev):
        rc = load().bess_dense_sgd(_dtype_code(table), table.data_ptr(), grad.data_ptr(), table.numel(), lr, _stream(dev))
    _check(rc, "bess_dense_sgd")


SEGMENT_CAP = 256  # BESS_SEGMENT_CAP of include/besskge_hip.h


class SegmentIndex:
    """References grouped by destination row (see bess_build_segment_index)."""

    __slots__ = ("refs", "seg_rows", "seg_offsets", "n_seg", "n_refs", "max_seg", "long_segs", "long_cap", "long_count", "long_grad")

    def __init__(self, idx: torch.Tensor, n_rows: int, width: int = 0,
                 scratch: Optional[dict] = None) -> None:
        """`width` (row width W of the table the gradients are for): lets the scratch of the long-row
        tier be prepared with the index instead of at the first reduction.  `scratch` (a dict the
        caller keeps, e.g. one per model): the zero-initialised scratch of the long-row tier is taken
        from / left in it instead of being allocated and zeroed per index (every use leaves it zero;
        share it only between steps that run one after the other on the same stream)."""
        dev = _dev(idx, "idx")
        ip = _idx(idx, "idx")
        n = int(idx.numel())
        need = ctypes.c_size_t(0)
        _check(load().bess_segment_index_workspace(n, ctypes.byref(need)), "bess_segment_index_workspace")
        ws = torch.empty((need.value,), dtype=torch.uint8, device=dev)
        self._allocate(n, n_rows, dev)
        bits = max(1, int(n_rows - 1).bit_length())
        with _on(dev), _Timed("bess_build_segment_index", dev):
            rc = load().bess_build_segment_index(ip, n, bits, self.refs.data_ptr(), self.seg_rows.data_ptr(),
                                                 self.seg_offsets.data_ptr(), self.n_seg.data_ptr(),
                                                 self.long_segs.data_ptr(), self.long_cap, ws.data_ptr(),
                                                 need.value, _stream(dev))
        _check(rc, "bess_build_segment_index")
        self._long_scratch(dev, width, scratch)

    def _allocate(self, n: int, n_rows: int, dev: torch.device) -> None:
        # ONE allocation for the index arrays (a notebook-size step builds an index per update: five
        # torch.empty calls were a measurable part of its host time)
        long_cap = n // SEGMENT_CAP + 1
        na, nb = (n + 3) & ~3, (n + 4) & ~3  # every array starts on a 16-byte boundary
        buf = torch.empty((2 * na + nb + 4 + long_cap + 1,), dtype=torch.int32, device=dev)
        self.refs = buf[:n]
        self.seg_rows = buf[na: na + n]
        self.seg_offsets = buf[2 * na: 2 * na + n + 1]
        self.n_seg = buf[2 * na + nb: 2 * na + nb + 1]  # always written by the build
        self.n_refs = n
        self.max_seg = min(n, int(n_rows))
        # rows with more than SEGMENT_CAP references (padded candidate lists, hot entities): listed
        # here, reduced by the whole device instead of one 16-lane group (include/besskge_hip.h)
        self.long_cap = long_cap
        self.long_segs = buf[2 * na + nb + 4:]

    def _long_scratch(self, dev: torch.device, width: int, scratch: Optional[dict]) -> None:
        # scratch of the long-row tier: zero before the first use, left zero / consistent by every use;
        # zeroed here, i.e. on the stream that builds the index (off the critical path of a training step)
        if scratch is not None:
            key = (dev, "long", int(width))
            have = scratch.get(key)
            if have is None or have[0].shape[0] < self.long_cap:
                have = (torch.zeros((self.long_cap,), dtype=torch.int32, device=dev),
                        torch.zeros((self.long_cap, int(width)), dtype=torch.float32, device=dev) if width else None)
                scratch[key] = have
            self.long_count, self.long_grad = have
        else:
            self.long_count = torch.zeros((self.long_cap,), dtype=torch.int32, device=dev)
            self.long_grad = torch.zeros((self.long_cap, int(width)), dtype=torch.float32, device=dev) if width else None


def step_prologue(jobs: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], int]],
                  id_lists: Sequence[torch.Tensor] = (), n_rows: int = 0, width: int = 0,
                  scratch: Optional[dict] = None) -> Optional[SegmentIndex]:
    """ONE launch in front of a notebook-size training step (`bess_step_prologue`): the copy / fill `jobs`
    (dst, src | None, fill word) over 4-byte elements - dst (and src) contiguous, same number of elements - and
    the segment index of the concatenation of the int32 row-id `id_lists` (read where they are; `n_rows` rows
    in the table they index), returned as a SegmentIndex, or None without lists."""
    if not jobs and not id_lists:
        return None
    if len(jobs) > MAX_WORD_JOBS or len(id_lists) > MAX_ROW_LISTS:
        raise ValueError("step_prologue: too many jobs / id lists")
    tensors = [("job dst", j[0]) for j in jobs] + [("job src", j[1]) for j in jobs] + [("ids", x) for x in id_lists]
    dev = _same_device(tensors)
    nj = len(jobs)
    dst, src = (_vp * max(1, nj))(), (_vp * max(1, nj))()
    val, words = (ctypes.c_uint32 * max(1, nj))(), (_i64 * max(1, nj))()
    for i, (d_, s_, v_) in enumerate(jobs):
        if d_.element_size() != 4 or not d_.is_contiguous():
            raise ValueError("step_prologue: job targets must be contiguous tensors of 4-byte elements")
        if s_ is not None and (s_.element_size() != 4 or not s_.is_contiguous() or s_.numel() != d_.numel()):
            raise ValueError("step_prologue: a job's source must match its target")
        dst[i], src[i], val[i], words[i] = d_.data_ptr(), (s_.data_ptr() if s_ is not None else None), int(v_), d_.numel()
    nl = len(id_lists)
    lp, ll = (_vp * max(1, nl))(), (_i64 * max(1, nl))()
    n_ids = 0
    for i, x in enumerate(id_lists):
        if x.dtype != torch.int32 or not x.is_contiguous():
            raise ValueError("step_prologue: id lists must be contiguous int32 tensors")
        lp[i], ll[i] = x.data_ptr(), x.numel()
        n_ids += int(x.numel())
    seg = None
    if nl:
        if not 0 < n_ids <= SMALL_INDEX_MAX:
            raise ValueError(f"step_prologue: {n_ids} row ids (1 .. {SMALL_INDEX_MAX})")
        seg = SegmentIndex.__new__(SegmentIndex)
        seg._allocate(n_ids, n_rows, dev)
    bits = max(1, int(max(1, n_rows) - 1).bit_length())
    with _on(dev), _Timed("bess_step_prologue", dev):
        rc = load().bess_step_prologue(
            nj, dst, src, val, words, nl, lp, ll, bits,
            seg.refs.data_ptr() if seg else None, seg.seg_rows.data_ptr() if seg else None,
            seg.seg_offsets.data_ptr() if seg else None, seg.n_seg.data_ptr() if seg else None,
            seg.long_segs.data_ptr() if seg else None, seg.long_cap if seg else 0, _stream(dev))
    _check(rc, "bess_step_prologue")
    if seg is not None:
        seg._long_scratch(dev, width, scratch)
    return seg


MAX_WORD_JOBS = 8  # BESS_MAX_WORD_JOBS
SMALL_INDEX_MAX = 15360  # BESS_SMALL_INDEX_MAX


class _IdentitySegments:
    """Segments of a table whose every row is updated: row s is segment s."""

    __slots__ = ("seg_rows", "n_seg", "max_seg")

    def __init__(self, n_rows: int, device: torch.device) -> None:
        self.seg_rows = torch.arange(n_rows, dtype=torch.int32, device=device)
        self.n_seg = torch.full((1,), n_rows, dtype=torch.int32, device=device)
        self.max_seg = n_rows


def identity_segments(n_rows: int, device: torch.device) -> _IdentitySegments:
    return _IdentitySegments(n_rows, device)


def neg_pertriple_grad_segments(d: ModelDesc, query: torch.Tensor, table: torch.Tensor, n_neg: int,
                                d_out: torch.Tensor, seg: SegmentIndex,
                                fused_sgd_lr: Optional[float] = None) -> Optional[torch.Tensor]:
    """grad_seg [seg.max_seg, W] f32: one gradient row per unique destination row;
    or, with fused_sgd_lr, apply `table[row] -= lr * grad` in place and return None."""
    dev = _same_device([("query", query), ("table", table), ("d_out", d_out), ("refs", seg.refs)])
    _f32(query, "query")
    _f32(d_out, "d_out")
    _rows(table, "table", d.width)
    nq = int(query.shape[0])
    if _dtype_code(table) != d.dtype or tuple(query.shape) != (nq, query_width(d)) \
            or tuple(d_out.shape) != (nq, n_neg) \
            or seg.n_refs != nq * n_neg:
        raise ValueError("neg_pertriple_grad_segments: operand shapes do not match")
    fused = fused_sgd_lr is not None
    grad = None if fused else torch.empty((seg.max_seg, d.width), dtype=torch.float32, device=dev)
    native = True  # every scorer id has a segmented reduction with a long-row tier
    if native and (seg.long_grad is None or seg.long_grad.shape[1] != d.width):
        seg.long_grad = torch.zeros((seg.long_cap, d.width), dtype=torch.float32, device=dev)
    long_grad = seg.long_grad
    with _on(dev), _Timed("bess_neg_pertriple_grad_segments", dev):
        rc = load().bess_neg_pertriple_grad_segments(ctypes.byref(d), query.data_ptr(), nq, table.data_ptr(), n_neg,
                                                     d_out.data_ptr(), n_neg, seg.refs.data_ptr(),
                                                     seg.seg_rows.data_ptr(), seg.seg_offsets.data_ptr(),
                                                     seg.n_seg.data_ptr(), seg.max_seg,
                                                     0 if fused else grad.data_ptr(),
                                                     float(fused_sgd_lr) if fused else 0.0,
                                                     seg.long_segs.data_ptr() if native else None,
                                                     seg.long_cap if native else 0,
                                                     long_grad.data_ptr() if native else None,
                                                     seg.long_count.data_ptr() if native else None, _stream(dev))
    _check(rc, "bess_neg_pertriple_grad_segments")
    return grad


def pad_segments(seg: SegmentIndex, grad_seg: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(row ids [max_seg], gradient rows [max_seg, W]) of a segment list with its unused tail neutralised (copies
    of the last real row, zero gradients): an ordinary row list of host-known length - no read-back of n_seg."""
    dev = _same_device([("grad_seg", grad_seg), ("seg_rows", seg.seg_rows)])
    _f32(grad_seg, "grad_seg")
    if grad_seg.dim() != 2 or grad_seg.shape[0] != seg.max_seg or not grad_seg.is_contiguous():
        raise ValueError("pad_segments: grad_seg must be a contiguous [seg.max_seg, W] tensor")
    with _on(dev):
        rc = load().bess_pad_segments(seg.seg_rows.data_ptr(), seg.n_seg.data_ptr(), seg.max_seg, grad_seg.data_ptr(),
                                      int(grad_seg.shape[1]), _stream(dev))
    _check(rc, "bess_pad_segments")
    return seg.seg_rows[: seg.max_seg], grad_seg


def apply_segments_sgd(table: torch.Tensor, seg: SegmentIndex, grad_seg: torch.Tensor, lr: float) -> None:
    dev = _same_device([("table", table), ("grad_seg", grad_seg), ("seg_rows", seg.seg_rows)])
    W = int(table.shape[1])
    _rows(table, "table", W)
    _f32(grad_seg, "grad_seg")
    if tuple(grad_seg.shape) != (seg.max_seg, W):
        raise ValueError("apply_segments_sgd: grad_seg shape mismatch")
    with _on(dev), _Timed("bess_apply_segments_sgd", dev):
        rc = load().bess_apply_segments_sgd(_dtype_code(table), W, table.data_ptr(), seg.seg_rows.data_ptr(),
                                            seg.n_seg.data_ptr(), seg.max_seg, grad_seg.data_ptr(), lr, _stream(dev))
    _check(rc, "bess_apply_segments_sgd")


def segment_sum_rows(src: torch.Tensor, seg: SegmentIndex) -> torch.Tensor:
    """grad_seg [seg.max_seg, W]: rows of `src` [n_refs, W] summed per destination row."""
    dev = _same_device([("src", src), ("refs", seg.refs)])
    _f32(src, "src")
    if src.dim() != 2 or src.shape[0] != seg.n_refs:
        raise ValueError("segment_sum_rows: src must be [n_refs, W]")
    W = int(src.shape[1])
    out = torch.empty((seg.max_seg, W), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = load().bess_segment_sum_rows(W, src.data_ptr(), seg.refs.data_ptr(), seg.seg_offsets.data_ptr(),
                                          seg.n_seg.data_ptr(), seg.max_seg, out.data_ptr(), _stream(dev))
    _check(rc, "bess_segment_sum_rows")
    return out


def apply_segments_opt(o: OptDesc, table: torch.Tensor, seg: SegmentIndex, grad_seg: torch.Tensor,
                       state1: Optional[torch.Tensor], state2: Optional[torch.Tensor],
                       keep: Optional[torch.Tensor] = None) -> None:
    """`keep` (int32 [seg.max_seg]): update only the segments with a non-zero entry."""
    dev = _same_device([("table", table), ("grad_seg", grad_seg), ("state1", state1), ("state2", state2)])
    W = int(table.shape[1])
    _rows(table, "table", W)
    _f32(grad_seg, "grad_seg")
    if tuple(grad_seg.shape) != (seg.max_seg, W):
        raise ValueError("apply_segments_opt: grad_seg shape mismatch")
    _state_ok(state1, table, o, "state1")
    _state_ok(state2, table, o, "state2")
    with _on(dev):
        rc = load().bess_apply_segments_opt(ctypes.byref(o), _dtype_code(table), W, table.data_ptr(),
                                            seg.seg_rows.data_ptr(), seg.n_seg.data_ptr(), seg.max_seg,
                                            grad_seg.data_ptr(), state1.data_ptr() if state1 is not None else 0,
                                            state2.data_ptr() if state2 is not None else 0,
                                            keep.data_ptr() if keep is not None else None, _stream(dev))
    _check(rc, "bess_apply_segments_opt")


MAX_ROW_LISTS = 8  # BESS_MAX_ROW_LISTS


def coalesced_update(o: Optional[OptDesc], table: torch.Tensor, seg: SegmentIndex, grads: Sequence[torch.Tensor],
                     state1: Optional[torch.Tensor] = None, state2: Optional[torch.Tensor] = None,
                     keep: Optional[torch.Tensor] = None, sum_only: bool = False,
                     axpy: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None) -> Optional[torch.Tensor]:
    """K9 + K10 of the small lists in one pass: `seg` indexes the concatenation of the lists' row
    ids, `grads[l]` is the f32 [n_l, W] gradient of list l.  Every unique row gets one update
    (one rounding).  `sum_only`: return the per-unique-row sums [seg.max_seg, W] instead.
    `axpy` = (table2, grad2, alpha): `table2 += alpha * grad2` (dense, same dtype as `table`) in the same launch."""
    dev = _same_device([("table", table), ("refs", seg.refs), ("state1", state1), ("state2", state2)]
                       + [(f"grads[{i}]", g) for i, g in enumerate(grads)])
    W = int(table.shape[1])
    _rows(table, "table", W)
    if not 1 <= len(grads) <= MAX_ROW_LISTS:
        raise ValueError(f"coalesced_update: {len(grads)} lists (1 .. {MAX_ROW_LISTS})")
    total = 0
    for i, g in enumerate(grads):
        _f32(g, f"grads[{i}]")
        if g.dim() != 2 or g.shape[1] != W:
            raise ValueError(f"coalesced_update: grads[{i}] must be [n, {W}]")
        total += int(g.shape[0])
    if total != seg.n_refs:
        raise ValueError(f"coalesced_update: the lists hold {total} rows, the index {seg.n_refs} references")
    _state_ok(state1, table, o, "state1")
    _state_ok(state2, table, o, "state2")
    ptrs = (_vp * len(grads))(*[g.data_ptr() for g in grads])
    rows = (_i64 * len(grads))(*[int(g.shape[0]) for g in grads])
    out = torch.empty((seg.max_seg, W), dtype=torch.float32, device=dev) if sum_only else None
    x_table = x_grad = None
    x_n, x_alpha = 0, 0.0
    if axpy is not None:
        x_table, x_grad, x_alpha = axpy
        _same_device([("table", table), ("axpy table", x_table), ("axpy grad", x_grad)])
        _f32(x_grad, "axpy grad")
        if x_table.dtype != table.dtype or not x_table.is_contiguous() or x_grad.numel() != x_table.numel():
            raise ValueError("coalesced_update: the axpy table must be contiguous, of the table's dtype, and match its gradient")
        x_n = int(x_table.numel())
    with _on(dev), _Timed("bess_coalesced_update", dev):
        rc = load().bess_coalesced_update_axpy(
            ctypes.byref(o) if o is not None else None, _dtype_code(table), W, table.data_ptr(), len(grads), ptrs, rows,
            seg.refs.data_ptr(), seg.seg_rows.data_ptr(), seg.seg_offsets.data_ptr(), seg.n_seg.data_ptr(), seg.max_seg,
            state1.data_ptr() if state1 is not None else None, state2.data_ptr() if state2 is not None else None,
            keep.data_ptr() if keep is not None else None, out.data_ptr() if out is not None else None,
            x_table.data_ptr() if x_n else None, x_grad.data_ptr() if x_n else None, x_n, float(x_alpha), _stream(dev))
    _check(rc, "bess_coalesced_update")
    return out


class DirectAccumulator:
    """Scratch of `bess_direct_update` for one table: `acc` [M, W] f32 (zero between steps), `claim` [M] int32 (the
    generation of the last step that updated each row), `generation` [1] int32 on the device (incremented once per
    step, before the update: `increment_job()` is the copy job for `step_prologue` that does it)."""

    __slots__ = ("acc", "claim", "generation")

    def __init__(self, table: torch.Tensor) -> None:
        dev = _dev(table, "table")
        self.acc = torch.zeros(tuple(table.shape), dtype=torch.float32, device=dev)
        self.claim = torch.zeros((int(table.shape[0]),), dtype=torch.int32, device=dev)
        self.generation = torch.ones((1,), dtype=torch.int32, device=dev)

    @staticmethod
    def bytes_for(table: torch.Tensor) -> int:
        return int(table.shape[0]) * (int(table.shape[1]) * 4 + 4)

    def increment_job(self) -> Tuple[torch.Tensor, torch.Tensor, int]:
        return (self.generation, self.generation, 1)


def direct_update(o: OptDesc, table: torch.Tensor, id_lists: Sequence[torch.Tensor], scratch: DirectAccumulator,
                  state1: Optional[torch.Tensor] = None, state2: Optional[torch.Tensor] = None,
                  axpy: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None) -> None:
    """K9 + K10 without an index (`bess_direct_update`): the gradients of the rows named by `id_lists` have been
    added into `scratch.acc` at their row ids; every such row gets ONE optimiser update with its summed gradient,
    its acc row goes back to zero.  `scratch.generation` must have been incremented for this step."""
    dev = _same_device([("table", table), ("acc", scratch.acc), ("state1", state1), ("state2", state2)]
                       + [(f"id_lists[{i}]", x) for i, x in enumerate(id_lists)])
    W = int(table.shape[1])
    _rows(table, "table", W)
    if not 1 <= len(id_lists) <= MAX_ROW_LISTS:
        raise ValueError(f"direct_update: {len(id_lists)} lists (1 .. {MAX_ROW_LISTS})")
    if tuple(scratch.acc.shape) != tuple(table.shape):
        raise ValueError("direct_update: the accumulator belongs to another table")
    for i, x in enumerate(id_lists):
        if x.dtype != torch.int32 or not x.is_contiguous():
            raise ValueError(f"direct_update: id_lists[{i}] must be a contiguous int32 tensor")
    _state_ok(state1, table, o, "state1")
    _state_ok(state2, table, o, "state2")
    ptrs = (_vp * len(id_lists))(*[x.data_ptr() for x in id_lists])
    lens = (_i64 * len(id_lists))(*[int(x.numel()) for x in id_lists])
    x_table = x_grad = None
    x_n, x_alpha = 0, 0.0
    if axpy is not None:
        x_table, x_grad, x_alpha = axpy
        _same_device([("table", table), ("axpy table", x_table), ("axpy grad", x_grad)])
        _f32(x_grad, "axpy grad")
        if x_table.dtype != table.dtype or not x_table.is_contiguous() or x_grad.numel() != x_table.numel():
            raise ValueError("direct_update: the axpy table must be contiguous, of the table's dtype, and match its gradient")
        x_n = int(x_table.numel())
    with _on(dev), _Timed("bess_direct_update", dev):
        rc = load().bess_direct_update(
            ctypes.byref(o), _dtype_code(table), W, table.data_ptr(), len(id_lists), ptrs, lens, scratch.acc.data_ptr(),
            scratch.claim.data_ptr(), scratch.generation.data_ptr(),
            state1.data_ptr() if state1 is not None else None, state2.data_ptr() if state2 is not None else None,
            x_table.data_ptr() if x_n else None, x_grad.data_ptr() if x_n else None, x_n, float(x_alpha), _stream(dev))
    _check(rc, "bess_direct_update")


def assign_state_rows(seg: Any, slot_map: torch.Tensor, counter: torch.Tensor, capacity: int,
                      keep: Optional[torch.Tensor] = None) -> None:
    """Paged optimiser state: give the unique rows of `seg` (those with keep != 0) a state row."""
    dev = _same_device([("seg_rows", seg.seg_rows), ("slot_map", slot_map), ("counter", counter), ("keep", keep)])
    if slot_map.dtype != torch.int32 or counter.dtype != torch.int32 or not slot_map.is_contiguous():
        raise ValueError("assign_state_rows: slot_map / counter must be int32")
    with _on(dev):
        rc = load().bess_assign_state_rows(seg.seg_rows.data_ptr(), seg.n_seg.data_ptr(), seg.max_seg,
                                           keep.data_ptr() if keep is not None else None, slot_map.data_ptr(),
                                           counter.data_ptr(), int(capacity), _stream(dev))
    _check(rc, "bess_assign_state_rows")


def _state_ok(st: Optional[torch.Tensor], table: torch.Tensor, o: Optional[OptDesc], name: str) -> None:
    if st is None:
        return
    _f32(st, name)
    paged = o is not None and o.slot_map != 0
    if st.dim() != 2 or st.shape[1] != table.shape[1] or (not paged and st.shape[0] != table.shape[0]):
        raise ValueError(f"{name} must be [{'capacity' if paged else table.shape[0]}, {table.shape[1]}] float32")


def map_extra_rows(seg: SegmentIndex, extra: SegmentIndex) -> Tuple[torch.Tensor, torch.Tensor]:
    """(extra_map [seg.max_seg], keep [extra.max_seg]) - see bess_neg_pertriple_step_segments."""
    dev = _same_device([("seg_rows", seg.seg_rows), ("extra_rows", extra.seg_rows)])
    xmap = torch.empty((seg.max_seg,), dtype=torch.int32, device=dev)
    keep = torch.empty((extra.max_seg,), dtype=torch.int32, device=dev)
    with _on(dev):
        rc = load().bess_map_extra_rows(seg.seg_rows.data_ptr(), seg.n_seg.data_ptr(), seg.max_seg,
                                        extra.seg_rows.data_ptr(), extra.n_seg.data_ptr(), extra.max_seg,
                                        xmap.data_ptr(), keep.data_ptr(), _stream(dev))
    _check(rc, "bess_map_extra_rows")
    return xmap, keep


def neg_pertriple_step_segments(d: ModelDesc, query: torch.Tensor, table: torch.Tensor, n_neg: int,
                                d_out: torch.Tensor, seg: SegmentIndex, o: OptDesc,
                                state1: Optional[torch.Tensor], state2: Optional[torch.Tensor],
                                extra_map: Optional[torch.Tensor] = None,
                                extra_sum: Optional[torch.Tensor] = None) -> None:
    """K9 + K10 in one pass: optimiser `o` applied to the unique rows of `seg` with their summed
    gradient (+ extra_sum[extra_map[s]]), in place (TransE / RotatE / DistMult / ComplEx)."""
    dev = _same_device([("query", query), ("table", table), ("d_out", d_out), ("refs", seg.refs),
                        ("state1", state1), ("state2", state2), ("extra_sum", extra_sum)])
    _f32(query, "query")
    _f32(d_out, "d_out")
    _rows(table, "table", d.width)
    nq = int(query.shape[0])
    if _dtype_code(table) != d.dtype or tuple(query.shape) != (nq, query_width(d)) \
            or tuple(d_out.shape) != (nq, n_neg) or seg.n_refs != nq * n_neg:
        raise ValueError("neg_pertriple_step_segments: operand shapes do not match")
    _state_ok(state1, table, o, "state1")
    _state_ok(state2, table, o, "state2")
    if (extra_map is None) != (extra_sum is None):
        raise ValueError("neg_pertriple_step_segments: extra_map and extra_sum come together")
    if extra_sum is not None:
        _f32(extra_sum, "extra_sum")
        if extra_sum.shape[1] != d.width or extra_map.dtype != torch.int32 or extra_map.numel() != seg.max_seg:
            raise ValueError("neg_pertriple_step_segments: extra_map / extra_sum shapes")
    if seg.long_grad is None or seg.long_grad.shape[1] != d.width:
        seg.long_grad = torch.zeros((seg.long_cap, d.width), dtype=torch.float32, device=dev)
    with _on(dev), _Timed("bess_neg_pertriple_step_segments", dev):
        rc = load().bess_neg_pertriple_step_segments(
            ctypes.byref(d), query.data_ptr(), nq, table.data_ptr(), n_neg, d_out.data_ptr(), n_neg,
            seg.refs.data_ptr(), seg.seg_rows.data_ptr(), seg.seg_offsets.data_ptr(), seg.n_seg.data_ptr(),
            seg.max_seg, seg.long_segs.data_ptr(), seg.long_cap, seg.long_grad.data_ptr(), seg.long_count.data_ptr(),
            ctypes.byref(o), state1.data_ptr() if state1 is not None else None,
            state2.data_ptr() if state2 is not None else None,
            extra_map.data_ptr() if extra_map is not None else None,
            extra_sum.data_ptr() if extra_sum is not None else None, _stream(dev))
    _check(rc, "bess_neg_pertriple_step_segments")


def ranks_from_scores(pos: torch.Tensor, cand: torch.Tensor, mode: int, worst_rank_infty: bool) -> torch.Tensor:
    dev = _same_device([("pos_score", pos), ("candidate_score", cand)])
    pos = pos.reshape(-1).float().contiguous()
    cand = cand.float().contiguous()
    if cand.dim() != 2 or cand.shape[0] != pos.numel():
        raise ValueError("`pos_score` and `candidate_score` need to have same size at dimension 0")
    out = torch.empty((pos.numel(),), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = load().bess_ranks_from_scores(pos.data_ptr(), cand.data_ptr(), pos.numel(), cand.shape[1], cand.shape[1],
                                           mode, int(worst_rank_infty), out.data_ptr(), _stream(dev))
    _check(rc, "bess_ranks_from_scores")
    return out


def ranks_from_indices(truth: torch.Tensor, cand: torch.Tensor, worst_rank_infty: bool) -> torch.Tensor:
    dev = _same_device([("ground_truth", truth), ("candidate_indices", cand)])
    truth = truth.reshape(-1).to(torch.int64).contiguous()
    cand = cand.to(torch.int64).contiguous()
    if cand.dim() != 2 or cand.shape[0] != truth.numel():
        raise ValueError("`ground_truth` and `candidate_indices` need to have the same size for dimension 0")
    out = torch.empty((truth.numel(),), dtype=torch.float32, device=dev)
    with _on(dev):
        rc = load().bess_ranks_from_indices(truth.data_ptr(), cand.data_ptr(), truth.numel(), cand.shape[1],
                                            int(worst_rank_infty), out.data_ptr(), _stream(dev))
    _check(rc, "bess_ranks_from_indices")
    return out


def topk_update(scores: torch.Tensor, best_score: torch.Tensor, best_id: torch.Tensor,
                ids: Optional[torch.Tensor] = None, id_base: int = 0, mask: Optional[torch.Tensor] = None,
                flags: Optional[torch.Tensor] = None) -> None:
    """Merge `scores` [rows, L] into the running (best_score, best_id) [rows, kk] lists in place.
    `flags` (from `neg_score_shared_fwd_pruned`): only the flagged blocks of 64 columns are read."""
    dev = _same_device([("scores", scores), ("best_score", best_score), ("best_id", best_id), ("ids", ids),
                        ("mask", mask), ("flags", flags)])
    if scores.dtype != torch.float32 or scores.dim() != 2 or (scores.shape[1] > 1 and scores.stride(1) != 1):
        raise ValueError("topk_update: scores must be float32 [rows, L] with contiguous rows")
    _f32(best_score, "best_score")
    if best_score.dim() != 2 or best_score.shape[0] != scores.shape[0] \
            or best_id.shape != best_score.shape or best_id.dtype != torch.int32 or not best_id.is_contiguous():
        raise ValueError("topk_update: best lists must be [rows, kk] (f32 scores, int32 ids)")
    R, L, kk = int(scores.shape[0]), int(scores.shape[1]), int(best_score.shape[1])
    ip = ir = 0
    if ids is not None:
        if ids.dtype != torch.int32 or ids.dim() != 2 or ids.shape[1] != L or ids.shape[0] not in (1, R) \
                or not ids.is_contiguous():
            raise ValueError("topk_update: ids must be a contiguous int32 [1 | rows, L] tensor")
        ip, ir = ids.data_ptr(), int(ids.shape[0])
    mp = mr = 0
    if mask is not None:
        if mask.dtype != torch.bool or mask.dim() != 2 or mask.shape[1] != L or mask.shape[0] not in (1, R) \
                or not mask.is_contiguous():
            raise ValueError("topk_update: mask must be a contiguous bool [1 | rows, L] tensor")
        mp, mr = mask.data_ptr(), int(mask.shape[0])
    if flags is not None:
        if ids is not None or mask is not None:
            raise ValueError("topk_update: flagged tiles take ids from id_base and no mask")
        if flags.dtype != torch.uint8 or flags.dim() != 2 or flags.shape[0] != R or not flags.is_contiguous() \
                or flags.shape[1] % 4 or flags.shape[1] * 64 < L:
            raise ValueError("topk_update: flags must be a contiguous uint8 [rows, 4 * ceil(L / 256)] tensor")
        with _on(dev), _Timed("bess_topk_update", dev):
            ld = int(scores.stride(0)) if R > 1 else L
            rc = load().bess_topk_update_flagged(scores.data_ptr(), R, L, max(ld, L), flags.data_ptr(),
                                                 int(flags.shape[1]), int(id_base), best_score.data_ptr(),
                                                 best_id.data_ptr(), kk, _stream(dev))
        _check(rc, "bess_topk_update_flagged")
        return
    with _on(dev), _Timed("bess_topk_update", dev):
        ld = int(scores.stride(0)) if R > 1 else L
        rc = load().bess_topk_update(scores.data_ptr(), R, L, max(ld, L), ip, ir, int(id_base), mp, mr,
                                     best_score.data_ptr(), best_id.data_ptr(), kk, _stream(dev))
    _check(rc, "bess_topk_update")


# --------------------------------------------------------------------------- #
# device-side index sampling
def _int_tensor(t: Optional[torch.Tensor], name: str, dtype: torch.dtype, numel: Optional[int] = None) -> int:
    if t is None:
        return 0
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"`{name}` must be contiguous {dtype}, got {t.dtype}")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"`{name}` has {t.numel()} elements, expected {numel}")
    return t.data_ptr()


def sample_negatives(gen: Pcg64State, jump_table: torch.Tensor, n_step: int, n_shard: int, src_begin: int,
                     src_count: int, B: int, K: int, shard_counts: torch.Tensor,
                     wanted_type: Optional[torch.Tensor] = None, type_counts: Optional[torch.Tensor] = None,
                     type_offsets: Optional[torch.Tensor] = None, local_sampling: bool = False) -> torch.Tensor:
    dev = _same_device([("jump_table", jump_table), ("shard_counts", shard_counts), ("wanted_type", wanted_type),
                        ("type_counts", type_counts), ("type_offsets", type_offsets)])
    tp = _int_tensor(jump_table, "jump_table", torch.int64, 64 * 4)
    cp = _int_tensor(shard_counts, "shard_counts", torch.int32, n_shard)
    n_type = 0
    if wanted_type is not None:
        if type_counts is None or type_offsets is None or type_counts.shape != type_offsets.shape \
                or type_counts.dim() != 2 or type_counts.shape[0] != n_shard:
            raise ValueError("type tables must both be [n_shard, n_type]")
        n_type = int(type_counts.shape[1])
    wp = _int_tensor(wanted_type, "wanted_type", torch.int32, n_step * n_shard * B)
    tcp = _int_tensor(type_counts, "type_counts", torch.int32)
    top = _int_tensor(type_offsets, "type_offsets", torch.int32)
    out = torch.empty((n_step, src_count, n_shard, B, K), dtype=torch.int32, device=dev)
    with _Timed("bess_sample_negatives", dev):
        _check(load().bess_sample_negatives(ctypes.byref(gen), tp, n_step, n_shard, src_begin, src_count, B, K, cp,
                                            wp, tcp, top, n_type, int(local_sampling), out.data_ptr(),
                                            _stream(dev)), "sample_negatives")
    return out


def sample_bucket_indices(gen: Pcg64State, jump_table: torch.Tensor, shape: Sequence[int], counts: torch.Tensor,
                          offsets: torch.Tensor) -> torch.Tensor:
    """`offsets + rng.integers(1 << 63, size=shape) % counts` with counts / offsets
    broadcast over shape[1:-1] (shape = [step, buckets..., inner])."""
    dev = _same_device([("jump_table", jump_table), ("counts", counts), ("offsets", offsets)])
    tp = _int_tensor(jump_table, "jump_table", torch.int64, 64 * 4)
    n_bucket = 1
    for s in shape[1:-1]:
        n_bucket *= int(s)
    cp = _int_tensor(counts, "counts", torch.int64, n_bucket)
    op = _int_tensor(offsets, "offsets", torch.int64, n_bucket)
    out = torch.empty(tuple(int(s) for s in shape), dtype=torch.int64, device=dev)
    _check(load().bess_sample_bucket_indices(ctypes.byref(gen), tp, out.numel(), int(shape[-1]), n_bucket, cp, op,
                                             out.data_ptr(), _stream(dev)), "sample_bucket_indices")
    return out


def lookup_triples(triples: torch.Tensor, sample_idx: torch.Tensor, swap_tail: bool,
                   want: Sequence[str] = ("head", "relation", "tail")) -> dict:
    """head / relation / tail (int32) of the sampled triples; sample_idx is
    [step, n, ppp] or [step, n, n, ppp]; the tail is block-transposed when swap_tail."""
    dev = _same_device([("triples", triples), ("sample_idx", sample_idx)])
    if triples.dtype != torch.int32 or triples.dim() != 2 or triples.shape[1] != 3 or not triples.is_contiguous():
        raise ValueError("`triples` must be a contiguous int32 [n_triple, 3] tensor")
    _int_tensor(sample_idx, "sample_idx", torch.int64)
    if sample_idx.dim() == 3:
        n_step, n1, ppp = sample_idx.shape
        n2 = 1
    elif sample_idx.dim() == 4:
        n_step, n1, n2, ppp = sample_idx.shape
    else:
        raise ValueError("`sample_idx` must be [step, n, ppp] or [step, n, n, ppp]")
    if swap_tail and sample_idx.dim() != 4:
        raise ValueError("the tail transpose needs shard-pair buckets")
    out = {}
    for k in want:
        shape = tuple(sample_idx.shape)
        if k == "tail" and swap_tail:
            shape = (n_step, n2, n1, ppp)
        out[k] = torch.empty(shape, dtype=torch.int32, device=dev)
    ptr = lambda k: out[k].data_ptr() if k in out else 0  # noqa: E731
    _check(load().bess_lookup_triples(triples.data_ptr(), triples.shape[0], sample_idx.data_ptr(), n_step, n1, n2,
                                      ppp, int(swap_tail), ptr("head"), ptr("relation"), ptr("tail"),
                                      _stream(dev)), "lookup_triples")
    return out


def gather_candidate_lists(table_h: torch.Tensor, mask_h: Optional[torch.Tensor], lookup: torch.Tensor,
                           table_t: Optional[torch.Tensor] = None, mask_t: Optional[torch.Tensor] = None,
                           per_part: int = 1, half: int = 1, mask_gather_layout: bool = False,
                           want_mask: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Fixed candidate lists of the sampled triples, laid out for the exchange (see
    bess_gather_candidate_lists): tables int32 [n_list, n_neg_shard, L] (+ bool masks), lookup int64
    [n_step, n_shard, T] -> (entities int32 [n_step, n_neg_shard, n_shard, T, L], mask bool | None)."""
    dev = _same_device([("table_h", table_h), ("mask_h", mask_h), ("lookup", lookup), ("table_t", table_t),
                        ("mask_t", mask_t)])
    for t, nm in ((table_h, "table_h"), (table_t, "table_t")):
        if t is not None and (t.dtype != torch.int32 or t.dim() != 3 or not t.is_contiguous()):
            raise ValueError(f"`{nm}` must be a contiguous int32 [n_list, n_neg_shard, L] tensor")
    for t, nm in ((mask_h, "mask_h"), (mask_t, "mask_t")):
        if t is not None and (t.dtype != torch.bool or tuple(t.shape) != tuple(table_h.shape) or not t.is_contiguous()):
            raise ValueError(f"`{nm}` must be a contiguous bool tensor of the tables' shape")
    if table_t is not None and tuple(table_t.shape) != tuple(table_h.shape):
        raise ValueError("the two candidate tables must have the same shape")
    if lookup.dtype != torch.int64 or lookup.dim() != 3 or not lookup.is_contiguous():
        raise ValueError("`lookup` must be a contiguous int64 [n_step, n_shard, T] tensor")
    n_list, n_neg_shard, L = (int(x) for x in table_h.shape)
    n_step, n_shard, T = (int(x) for x in lookup.shape)
    ent = torch.empty((n_step, n_neg_shard, n_shard, T, L), dtype=torch.int32, device=dev)
    msk = None
    if want_mask and mask_h is not None:
        shape = (n_step, n_neg_shard, n_shard, T, L) if mask_gather_layout else (n_step, n_shard, T, n_neg_shard, L)
        msk = torch.empty(shape, dtype=torch.bool, device=dev)
    with _on(dev), _Timed("bess_gather_candidate_lists", dev):
        rc = load().bess_gather_candidate_lists(
            table_h.data_ptr(), table_t.data_ptr() if table_t is not None else None,
            mask_h.data_ptr() if mask_h is not None else None, mask_t.data_ptr() if mask_t is not None else None,
            n_list, lookup.data_ptr(), n_step, n_shard, T, int(per_part), int(half), n_neg_shard, L,
            int(bool(mask_gather_layout)), ent.data_ptr(), msk.data_ptr() if msk is not None else None, _stream(dev))
    _check(rc, "bess_gather_candidate_lists")
    return ent, msk


# --------------------------------------------------------------------------- #
# collectives between shards (RCCL through the C ABI)
def comm_unique_id() -> bytes:
    """128 opaque bytes; made by one rank, handed to every rank's `Communicator`."""
    buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
    _check(load().bess_comm_unique_id(buf), "bess_comm_unique_id")
    return bytes(buf)


class Communicator:
    """`bess_comm*` of this rank (owns it: destroyed with the object).  Belongs to the HIP
    device that is current when it is made; collectives run on PyTorch's current stream
    of that device."""

    def __init__(self, world: int, rank: int, unique_id: bytes, device: torch.device) -> None:
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique id must be {COMM_ID_BYTES} bytes")
        if device.type != "cuda":
            raise RuntimeError("besskge: communicators live on HIP devices (there is no CPU fallback)")
        self.world, self.rank, self.device = int(world), int(rank), device
        self._h = _vp()
        buf = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        with torch.cuda.device(device):
            _check(load().bess_comm_init_rank(self.world, self.rank, buf, ctypes.byref(self._h)),
                   "bess_comm_init_rank")

    @classmethod
    def init_all(cls, devices: Sequence[torch.device]) -> List["Communicator"]:
        """One process driving several GPUs (`bess_comm_init_all` = ncclCommInitAll): communicator r of the
        clique lives on `devices[r]`.  Collectives of the clique's ranks must be issued from different host
        threads or inside one RCCL group; with one device it is a self-contained one-rank communicator."""
        n = len(devices)
        if n < 1 or any(d.type != "cuda" for d in devices):
            raise RuntimeError("besskge: communicators live on HIP devices (there is no CPU fallback)")
        ids = (ctypes.c_int32 * n)(*[d.index if d.index is not None else torch.cuda.current_device() for d in devices])
        handles = (_vp * n)()
        _check(load().bess_comm_init_all(n, ids, handles), "bess_comm_init_all")
        out = []
        for r, d in enumerate(devices):
            c = cls.__new__(cls)
            c.world, c.rank, c.device = n, r, torch.device("cuda", int(ids[r]))
            c._h = _vp(handles[r])
            out.append(c)
        return out

    def info(self) -> Tuple[int, int, int]:
        """(world, rank, device index) as the library holds them (`bess_comm_info`)."""
        w, r, d = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _check(load().bess_comm_info(self._h, ctypes.byref(w), ctypes.byref(r), ctypes.byref(d)), "bess_comm_info")
        return w.value, r.value, d.value

    def close(self) -> None:
        """`bess_comm_destroy`.  `ncclCommDestroy` waits for hipGraphs that recorded collectives of this
        communicator, so they must be destroyed first: a communicator that has seen one of its collectives
        recorded refuses to be destroyed until `graphs_released()` says they are gone (`NativeGroup.close()`
        does both, in order) - an error instead of a process that never ends."""
        if getattr(self, "_h", None) is not None and self._h.value:
            if getattr(self, "_recorded", False):
                raise RuntimeError(
                    "besskge: this communicator's collectives were recorded into a hipGraph; destroy the graphs "
                    "(NativeGroup.close() / ReplicaGroup.release_graphs()) and call graphs_released() first - "
                    "ncclCommDestroy would wait for them forever")
            torch.cuda.synchronize(self.device)
            h, self._h = self._h, _vp()
            _check(load().bess_comm_destroy(h), "bess_comm_destroy")

    def graphs_released(self) -> None:
        """The caller has destroyed every hipGraph that recorded this communicator's collectives."""
        self._recorded = False

    def _note_capture(self) -> None:
        if not getattr(self, "_recorded", False) and torch.cuda.is_current_stream_capturing():
            self._recorded = True

    def __del__(self, _finalizing: Any = sys.is_finalizing) -> None:  # pragma: no cover - shutdown order
        # Never `ncclCommDestroy` from a finaliser that cannot know the graphs are gone: at interpreter shutdown
        # (objects die in no useful order) or when a collective was recorded and nobody released the graphs, the
        # handle is left to the process teardown.
        if _finalizing() or getattr(self, "_recorded", False):
            return
        try:
            self.close()
        except Exception:
            pass

    def _buf(self, t: torch.Tensor, name: str) -> int:
        if not t.is_cuda or t.device != self.device:
            raise RuntimeError(f"besskge: `{name}` is on {t.device}, the communicator on {self.device}")
        if not t.is_contiguous():
            raise ValueError(f"`{name}` must be contiguous")
        return t.data_ptr()

    def all_to_all(self, send: torch.Tensor, recv: Optional[torch.Tensor] = None) -> torch.Tensor:
        """send [world, ...]: block p goes to rank p; returns recv (block p came from rank p)."""
        if send.dim() < 1 or send.shape[0] != self.world:
            raise ValueError(f"all_to_all: leading dim {tuple(send.shape)[:1]} != world {self.world}")
        if recv is None:
            recv = torch.empty_like(send)
        elif recv.shape != send.shape or recv.dtype != send.dtype:
            raise ValueError("all_to_all: recv does not match send")
        per_peer = send[0].numel() * send.element_size()
        self._note_capture()
        with _on(self.device), _Timed("bess_alltoall", self.device):
            rc = load().bess_alltoall(self._h, self._buf(send, "send"), self._buf(recv, "recv"), per_peer,
                                      _stream(self.device))
        _check(rc, "bess_alltoall")
        return recv

    def all_gather(self, send: torch.Tensor) -> torch.Tensor:
        """recv [world, *send.shape] in rank order."""
        recv = torch.empty((self.world, *send.shape), dtype=send.dtype, device=send.device)
        self._note_capture()
        with _on(self.device), _Timed("bess_allgather", self.device):
            rc = load().bess_allgather(self._h, self._buf(send, "send"), self._buf(recv, "recv"),
                                       send.numel() * send.element_size(), _stream(self.device))
        _check(rc, "bess_allgather")
        return recv

    def all_reduce_sum_(self, x: torch.Tensor) -> torch.Tensor:
        """In-place sum over ranks of a float32 tensor."""
        _f32(x, "x")
        self._note_capture()
        with _on(self.device), _Timed("bess_allreduce_sum_f32", self.device):
            p = self._buf(x, "x")
            rc = load().bess_allreduce_sum_f32(self._h, p, p, x.numel(), _stream(self.device))
        _check(rc, "bess_allreduce_sum_f32")
        return x

    def pack_exchange(self, table: torch.Tensor, idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """K1 + C1: idx [world, L] rows of `table` -> (send, recv), both [world, L, W]."""
        W = int(table.shape[1])
        _rows(table, "table", W)
        if idx.dtype != torch.int32 or idx.dim() != 2 or idx.shape[0] != self.world or not idx.is_contiguous():
            raise ValueError("pack_exchange: idx must be a contiguous int32 [world, L] tensor")
        L = int(idx.shape[1])
        send = torch.empty((self.world, L, W), dtype=table.dtype, device=table.device)
        recv = torch.empty_like(send)
        self._note_capture()
        with _on(self.device), _Timed("bess_pack_exchange", self.device):
            rc = load().bess_pack_exchange(self._h, _dtype_code(table), W, self._buf(table, "table"),
                                           self._buf(idx, "idx"), L, send.data_ptr(), recv.data_ptr(),
                                           _stream(self.device))
        _check(rc, "bess_pack_exchange")
        return send, recv


# --------------------------------------------------------------------------- #
# recorded steps
GRAPH_NODE_KINDS = ("kernel", "memcpy", "memset", "host", "graph", "empty", "wait_event", "event_record",
                    "ext_sem_signal", "ext_sem_wait", "mem_alloc", "mem_free", "memcpy_from_symbol",
                    "memcpy_to_symbol", "batch_mem_op")


def graph_node_counts(graph: Any) -> dict:
    """Node types of a recorded step: `graph` is a `torch.cuda.CUDAGraph(keep_graph=True)` (its hipGraph_t is
    `raw_cuda_graph()`) or a raw `hipGraph_t` address.  {"kernel": n, "memset": n, ...}, zero counts left out."""
    raw = graph if isinstance(graph, int) else graph.raw_cuda_graph()
    counts = (ctypes.c_int32 * len(GRAPH_NODE_KINDS))()
    _check(load().bess_graph_node_counts(ctypes.c_void_p(raw), counts, len(GRAPH_NODE_KINDS)),
           "bess_graph_node_counts")
    return {k: int(c) for k, c in zip(GRAPH_NODE_KINDS, counts) if c}


# --------------------------------------------------------------------------- #
# step plans (csrc/plan.hip): a step as a recorded list of the library's own calls, replayed from C
PLAN_ARG_INT, PLAN_ARG_FLOAT, PLAN_ARG_PTR, PLAN_ARG_BLOB, PLAN_ARG_STREAM = 0, 1, 2, 3, 4
_INT_TYPES = (ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int)


class Plan:
    """`bess_plan*`: the calls of one recorded step (`record_plan`); `run()` issues them again on PyTorch's current
    stream of `device` - no Python between the calls, and none of the GIL while they are issued (ctypes releases it
    for the duration of a foreign call), so a process that drives several GPUs runs one plan per device from one
    host thread each.  The plan names device buffers by address: the recorder's caller keeps them alive and in place
    (`Runner`: a private memory pool + static input buffers)."""

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self._h = _vp()
        _check(load().bess_plan_create(ctypes.byref(self._h)), "bess_plan_create")
        self.names: List[str] = []

    def __len__(self) -> int:
        return int(_real_lib().bess_plan_length(self._h))

    def _add(self, name: str, argtypes: Sequence[Any], args: Sequence[Any]) -> None:
        n = len(args)
        kinds = (ctypes.c_uint8 * max(1, n))()
        values = (ctypes.c_uint64 * max(1, n))()
        blobs = (_vp * max(1, n))()
        sizes = (_i64 * max(1, n))()
        keep = []  # the ctypes objects whose bytes are copied must outlive the call
        for k, (tp, a) in enumerate(zip(argtypes, args)):
            if k == n - 1:
                kinds[k] = PLAN_ARG_STREAM
            elif tp in _INT_TYPES:
                kinds[k], values[k] = PLAN_ARG_INT, int(a) & 0xFFFFFFFFFFFFFFFF
            elif tp is ctypes.c_float or tp is ctypes.c_double:
                kinds[k] = PLAN_ARG_FLOAT
                values[k] = ctypes.c_uint64.from_buffer_copy(ctypes.c_double(float(a))).value
            elif tp is _vp:
                kinds[k] = PLAN_ARG_PTR
                values[k] = int(a.value or 0) if isinstance(a, ctypes.c_void_p) else int(a or 0)
            else:  # POINTER(...): NULL, byref(struct), or a ctypes array - host bytes the call reads
                obj = getattr(a, "_obj", a)
                if a is None:
                    kinds[k], values[k] = PLAN_ARG_PTR, 0
                elif isinstance(obj, (ctypes.Structure, ctypes.Array, ctypes._SimpleCData)):
                    kinds[k] = PLAN_ARG_BLOB
                    blobs[k], sizes[k] = ctypes.addressof(obj), ctypes.sizeof(obj)
                    keep.append(obj)
                else:
                    raise TypeError(f"record_plan: {name} argument {k}: cannot record a {type(a).__name__}")
        _check(_real_lib().bess_plan_add_call(self._h, name.encode(), n, kinds, values, blobs, sizes), "bess_plan_add_call")
        self.names.append(name)

    def run(self) -> None:
        with _on(self.device):
            rc = _real_lib().bess_plan_run(self._h, _stream(self.device))
        _check(rc, "bess_plan_run")

    def run_on(self, stream: int) -> None:
        """`bess_plan_run` on a raw hipStream_t (worker threads of a multi-device process: the device must be the
        thread's current one)."""
        _check(_real_lib().bess_plan_run(self._h, stream), "bess_plan_run")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            h, self._h = self._h, _vp()
            _real_lib().bess_plan_destroy(h)

    def __del__(self, _finalizing: Any = sys.is_finalizing) -> None:  # pragma: no cover
        if not _finalizing():
            try:
                self.close()
            except Exception:
                pass


class _RecordingLib:
    """Stands in for the library while a plan is recorded: every call goes through to it; the ones that enqueue
    work are noted in the plan with their argument values."""

    def __init__(self, lib: ctypes.CDLL, plan: Plan) -> None:
        self._lib, self._plan = lib, plan
        self._wrapped: dict = {}

    def __getattr__(self, name: str) -> Any:
        fn = getattr(self._lib, name)
        if not name.startswith("bess_") or not self._lib.bess_plan_knows(name.encode()):
            return fn
        w = self._wrapped.get(name)
        if w is None:
            plan, argtypes = self._plan, fn.argtypes

            def w(*args: Any, _fn: Any = fn, _name: str = name) -> int:
                rc = _fn(*args)
                if rc == 0:
                    plan._add(_name, argtypes, args)
                return rc

            self._wrapped[name] = w
        return w


def _real_lib() -> ctypes.CDLL:
    lib = load()
    return lib._lib if isinstance(lib, _RecordingLib) else lib


@contextlib.contextmanager
def record_plan(device: torch.device):
    """Run a step inside this context: it executes as usual, and every library call that enqueues work is noted in
    the `Plan` the context yields (`plan.run()` issues them again).  The step must be made of library calls only -
    work enqueued by anything else (a torch operator) is not part of the plan - and must not allocate device
    memory outside a pool the caller keeps (see `Runner._call_with_plans`)."""
    global _lib
    real = load()
    if isinstance(real, _RecordingLib):
        raise RuntimeError("record_plan: already recording")
    plan = Plan(device)
    _lib = _RecordingLib(real, plan)  # type: ignore[assignment]
    try:
        yield plan
    finally:
        _lib = real
