"""The C ABI library loads (no GPU needed) and exports exactly what
include/besskge_hip.h declares; the ctypes binding covers every entry point;
invalid arguments are rejected with an error code + message instead of a crash
(no kernel is launched in these tests)."""

import ctypes
import os
import re

import pytest

from conftest import REPO


def header_functions():
    text = open(os.path.join(REPO, "include", "besskge_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(bess_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from besskge import _native

    lib = _native.load()
    names = header_functions()
    assert len(names) >= 16
    for name in names:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_native.SIGNATURES) == names, "ctypes binding and header disagree"
    assert lib.bess_version() == _native.ABI_VERSION == 3


def test_integration_doc_matches_the_library():
    """INTEGRATION.md quotes the number of entry points, the ABI version its stub asserts and the link line."""
    from besskge import _native

    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    m = re.search(r"C ABI \((\d+) entry points", doc)
    assert m and int(m.group(1)) == len(header_functions()), (m and m.group(1), len(header_functions()))
    m = re.search(r"bess_version\(\) == (\d+)", doc)
    assert m and int(m.group(1)) == _native.ABI_VERSION
    makefile = open(os.path.join(REPO, "bess-kge_amd", "csrc", "Makefile")).read()
    assert ("-lrccl" in makefile) == ("-lrccl" in doc)


def test_workspace_query_needs_no_gpu():
    """bess_neg_score_shared_workspace is host arithmetic: bilinear scorers with >= 256 output
    tiles of 128 x 128 ask for (S + min(N, 65536)) lines (rows rounded up to whole 256 / 128-row
    tiles) of ceil(W / 32) * 128 bytes (+ 256 for the range flag), the rest 0."""
    from besskge import _native

    lib = _native.load()
    d = _native.ModelDesc()
    d.scorer, d.norm_p, d.dtype, d.width, d.rel_width = _native.DISTMULT, 0, 0, 500, 500
    flag = 256  # tail of the scratch: the range flag of the fallback to the fp32 kernels
    assert lib.bess_neg_score_shared_workspace(ctypes.byref(d), 4096, 4096) == (4096 + 4096) * 16 * 128 + flag
    assert lib.bess_neg_score_shared_workspace(ctypes.byref(d), 4000, 5000) == (4096 + 5120) * 16 * 128 + flag  # whole tiles
    assert lib.bess_neg_score_shared_workspace(ctypes.byref(d), 4096, 1 << 20) == (4096 + 65536) * 16 * 128 + flag
    assert lib.bess_neg_score_shared_workspace(ctypes.byref(d), 512, 768) == 0  # too few tiles
    d.reserved[0] = _native.FLAG_FP32_MATH  # the exact fp32 kernels are asked for: no scratch
    assert lib.bess_neg_score_shared_workspace(ctypes.byref(d), 4096, 4096) == 0
    assert lib.bess_neg_score_shared_bwd_workspace(ctypes.byref(d), 4096, 4096) == 0
    d.reserved[0] = 0
    assert lib.bess_neg_score_shared_bwd_workspace(ctypes.byref(d), 4096, 4096) > 0
    d.scorer, d.norm_p = _native.TRANSE, 1
    assert lib.bess_neg_score_shared_workspace(ctypes.byref(d), 4096, 4096) == 0  # not a dot product


def test_struct_layouts_match_header():
    from besskge import _native

    assert ctypes.sizeof(_native.ModelDesc) == 32
    assert ctypes.sizeof(_native.LossDesc) == 32
    assert _native.ModelDesc.width.offset == 12 and _native.ModelDesc.rel_width.offset == 16
    assert _native.LossDesc.margin.offset == 8 and _native.LossDesc.ssce_shift.offset == 20


def last_error(lib):
    buf = ctypes.create_string_buffer(256)
    lib.bess_last_error(buf, 256)
    return buf.value.decode()


def test_invalid_arguments_return_error_codes():
    from besskge import _native

    lib = _native.load()
    d = _native.ModelDesc()
    d.scorer, d.norm_p, d.dtype, d.width, d.rel_width = 9, 1, 0, 8, 8
    assert lib.bess_query_fwd(ctypes.byref(d), 0, 0, 0, 0, 0, 0, 0, 0) == -1
    assert "unknown scorer" in last_error(lib)
    d.scorer, d.norm_p = _native.TRANSE, 0  # (any p >= 1 is a norm: TransE / RotatE take it)
    assert lib.bess_score_triple_fwd(ctypes.byref(d), 0, 0, 0, 0, 0, 0, 0, 0, 0) == -1
    assert "norm" in last_error(lib)
    d.scorer, d.norm_p, d.reserved[0] = _native.AFFINE, 0, 1  # the affine family and BoxE too (round 4: any p >= 1)
    assert lib.bess_query_fwd(ctypes.byref(d), 0, 0, 0, 0, 0, 0, 0, 0) == -1
    assert "norm" in last_error(lib)
    d.norm_p = 3
    assert lib.bess_query_fwd(ctypes.byref(d), 0, 0, 0, 0, 0, 0, 0, 0) == 0  # (n_query = 0: nothing to do, p = 3 accepted)
    d.reserved[0] = 0
    d.scorer, d.norm_p, d.width, d.rel_width = _native.ROTATE, 1, 8, 8  # RotatE needs Wr = W/2
    assert lib.bess_query_fwd(ctypes.byref(d), 0, 0, 0, 0, 0, 0, 0, 0) == -1
    assert "rel_width" in last_error(lib)
    d.rel_width = 4
    assert lib.bess_query_fwd(ctypes.byref(d), 7, 0, 0, 0, 0, 0, 0, 0) == -1  # bad side
    # NULL pointers with non-zero sizes are refused before any launch
    assert lib.bess_gather_rows(0, 8, 0, 0, 5, 0, 0) == -1
    assert lib.bess_gather_rows(5, 8, 0, 0, 0, 0, 0) == -1  # bad dtype
    # empty problems are fine without touching the device
    assert lib.bess_gather_rows(0, 8, 0, 0, 0, 0, 0) == 0
    assert lib.bess_scatter_add_rows(0, 8, 0, 0, 0, 1.0, 0) == 0
    l = _native.LossDesc()
    l.kind = 7
    assert lib.bess_loss_fwd_bwd(ctypes.byref(l), 0, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, 1, 0) == -1


def test_communicator_entry_points_reject_bad_arguments():
    """The RCCL entry points of SURVEY 8(b) (bess_comm_*, bess_alltoall, bess_allgather,
    bess_allreduce_sum_f32, bess_pack_exchange) are exported and validate before touching RCCL."""
    from besskge import _native

    lib = _native.load()
    for name in ("bess_comm_unique_id", "bess_comm_init_rank", "bess_comm_init_all", "bess_comm_destroy",
                 "bess_comm_info", "bess_alltoall", "bess_allgather", "bess_allreduce_sum_f32", "bess_pack_exchange"):
        assert name in _native.SIGNATURES and hasattr(lib, name)
    uid = _native.comm_unique_id()  # host-side: no device needed
    assert len(uid) == _native.COMM_ID_BYTES == 128 and uid != _native.comm_unique_id()
    h = ctypes.c_void_p()
    buf = (ctypes.c_uint8 * 128).from_buffer_copy(uid)
    assert lib.bess_comm_init_rank(2, 2, buf, ctypes.byref(h)) == -1 and "rank 2 of 2" in last_error(lib)
    assert lib.bess_comm_init_rank(0, 0, buf, ctypes.byref(h)) == -1
    assert lib.bess_comm_init_rank(1, 0, None, ctypes.byref(h)) == -1 and h.value is None
    assert lib.bess_alltoall(None, 0, 0, 16, 0) == -1 and "NULL communicator" in last_error(lib)
    assert lib.bess_allgather(None, 0, 0, 16, 0) == -1
    assert lib.bess_allreduce_sum_f32(None, 0, 0, 4, 0) == -1
    assert lib.bess_pack_exchange(None, 0, 8, 0, 0, 1, 0, 0, 0) == -1
    assert lib.bess_comm_info(None, None, None, None) == -1
    assert lib.bess_comm_destroy(None) == 0  # like free(NULL)


def test_native_group_needs_a_hip_device():
    from besskge import _native

    import torch

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.Communicator(1, 0, _native.comm_unique_id(), torch.device("cpu"))


def test_import_fails_loudly_without_the_library(tmp_path, monkeypatch):
    from besskge import _native

    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "library_path", lambda: tmp_path / "libbesskge_hip.so")
    with pytest.raises(ImportError, match="Cannot find the HIP extension library"):
        _native.load()
