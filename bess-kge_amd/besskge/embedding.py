"""Allocation / initialisation / re-sharding of the embedding tables.

SURVEY.md section 8a row a5 (reference `besskge/embedding.py:15-290`).

Layout contract (what the HIP kernels assume): the entity table is
`[n_shard, max_entity_per_shard, W]`, row-major, one contiguous `[M, W]` slice
per GPU; RotatE / ComplEx rows are `[re(d) | im(d)]`; the relation table
`[n_rel or 2*n_rel, Wr]` is replicated.  Tables are created in fp32
(reference `embedding.py:144,156,173-188`); `.half()` afterwards is supported.

Extension for tables that do not fit host memory (BASELINE config 5): every
function takes an optional `device`; with `shards=[k, ...]` only those slices
are allocated (`[len(shards), M, W]`), so each rank can create its 128 GB slice
directly in HBM.
"""

from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import torch

from besskge.sharding import Sharding

_Init = Union[torch.Tensor, List[Callable[..., torch.Tensor]]]


# --------------------------------------------------------------------------- #
# initialisers (in place on a freshly allocated slice, returning it)
def init_uniform_norm(embedding_table: torch.Tensor) -> torch.Tensor:
    """U[0, 1) entries, then every row scaled to unit L2 norm."""
    return torch.nn.functional.normalize(torch.nn.init.uniform_(embedding_table), dim=-1)


def init_xavier_norm(embedding_table: torch.Tensor, gain: float = 1.0) -> torch.Tensor:
    """Xavier-normal with fan_in = 0, fan_out = row size."""
    return torch.nn.init.normal_(
        embedding_table, std=gain * np.sqrt(2.0 / embedding_table.shape[-1])
    )


def init_uniform_rotation(embedding_table: torch.Tensor) -> torch.Tensor:
    """Phases uniform in [0, 2 pi) (RotatE relation table)."""
    return torch.rand_like(embedding_table) * 2 * np.pi


def init_KGE_uniform(
    embedding_table: torch.Tensor, b: float = 1.0, divide_by_embedding_size: bool = True
) -> torch.Tensor:
    """U(-b, b), with b divided by the row size by default."""
    bound = b / embedding_table.shape[-1] if divide_by_embedding_size else b
    return torch.nn.init.uniform_(embedding_table, -bound, bound)


def init_KGE_normal(
    embedding_table: torch.Tensor, std: float = 1.0, divide_by_embedding_size: bool = True
) -> torch.Tensor:
    """N(0, std), with std divided by the row size by default."""
    s = std / embedding_table.shape[-1] if divide_by_embedding_size else std
    return torch.nn.init.normal_(embedding_table, std=s)


# --------------------------------------------------------------------------- #
#: rows filled per pass when a table is created directly in a narrower dtype (fp32 scratch of this many scalars)
_INIT_CHUNK_SCALARS = 1 << 28


def _from_initializers(
    lead_shape: Sequence[int],
    initializer: List[Callable[..., torch.Tensor]],
    row_size: Optional[List[int]],
    device: Optional[torch.device],
    dtype: Optional[torch.dtype] = None,
) -> torch.Tensor:
    if not row_size:
        raise ValueError(
            "If not providing an embedding table, row_size needs to be specified"
        )
    if len(initializer) != len(row_size):
        raise ValueError("Different number of embedding splits and initializers provided")
    if dtype is not None and dtype != torch.float32:
        # the table is born in its final dtype (a 160 MB - 128 GB shard must not exist twice, once as
        # fp32): rows are initialised in fp32 blocks and converted on the way in.  Seed parity with the fp32
        # path (`table.half()` of the table the same seed gives in fp32) holds for tables of up to
        # _INIT_CHUNK_SCALARS scalars with a single initialiser: beyond one block, or with several column
        # pieces, the random stream is consumed block by block and piece by piece inside a block, not
        # piece by piece over the whole table as the reference does (embedding.py:173-188)
        W = int(sum(row_size))
        out = torch.empty(size=(*lead_shape, W), dtype=dtype, device=device)
        flat = out.view(-1, W)
        step = max(1, _INIT_CHUNK_SCALARS // W)
        for r0 in range(0, flat.shape[0], step):
            r1 = min(flat.shape[0], r0 + step)
            c0 = 0
            for w, init in zip(row_size, initializer):
                flat[r0:r1, c0:c0 + w] = init(torch.empty(size=(r1 - r0, w), dtype=torch.float32, device=device))
                c0 += w
        return out
    # one allocation; every initialiser fills its own column range.  The random
    # stream is consumed exactly as if the slices were created one after the
    # other and concatenated (reference embedding.py:173-188).
    pieces = [
        init(torch.empty(size=(*lead_shape, w), dtype=torch.float32, device=device))
        for w, init in zip(row_size, initializer)
    ]
    return pieces[0] if len(pieces) == 1 else torch.concat(pieces, dim=-1)


def initialize_entity_embedding(
    sharding: Sharding,
    initializer: _Init,
    row_size: Optional[List[int]] = None,
    device: Optional[torch.device] = None,
    shards: Optional[Sequence[int]] = None,
    dtype: Optional[torch.dtype] = None,
) -> torch.nn.Parameter:
    """Entity table `[n_shard, max_entity_per_shard, sum(row_size)]`.

    :param sharding: entity sharding.
    :param initializer: a table - sharded `[n_shard, M, W]` or unsharded
        `[n_entity, W]` (re-indexed by `shard_and_idx_to_entity`; padding rows
        copy the last entity) - or one initialising function per entry of
        `row_size`.
    :param row_size: widths of the pieces each row is made of.
    :param device: where to allocate (default: CPU, like the reference).
    :param shards: allocate only these shard slices (extension, see module doc).
    :param dtype: create the table in this dtype (extension; default float32 like the reference,
        to be narrowed with `.half()`).  With initialising functions the rows are drawn in fp32
        blocks and converted while the table is filled, so no fp32 copy of the shard ever exists.
    """
    n, M = sharding.n_shard, sharding.max_entity_per_shard
    keep = None if shards is None else list(shards)
    if isinstance(initializer, torch.Tensor):
        if initializer.dim() == 3:
            if tuple(initializer.shape[:2]) != (n, M):
                raise ValueError(
                    "Shape of sharded table provided for initialization"
                    " is not compatible with sharding"
                )
            table = initializer if keep is None else initializer[keep]
        elif initializer.dim() == 2:
            if initializer.shape[0] != sharding.n_entity:
                raise ValueError(
                    "Number of rows of table provided for initialization"
                    " different from number of entities."
                )
            ids = np.minimum(sharding.shard_and_idx_to_entity, sharding.n_entity - 1)
            if keep is not None:
                ids = ids[keep]
            table = initializer[torch.from_numpy(ids)]
        else:
            raise ValueError("Table for initialization needs to be 2- or 3-dimensional")
        table = table.to(dtype=dtype or torch.float32, device=device)
        if row_size:
            assert (
                sum(row_size) == table.shape[-1]
            ), "Initialization tensor and row_size provided are incompatible"
    else:
        n_alloc = n if keep is None else len(keep)
        table = _from_initializers((n_alloc, M), initializer, row_size, device, dtype)
    return torch.nn.Parameter(table)


def initialize_relation_embedding(
    n_relation_type: int,
    inverse_relations: bool,
    initializer: _Init,
    row_size: Optional[List[int]] = None,
    device: Optional[torch.device] = None,
    dtype: Optional[torch.dtype] = None,
) -> torch.nn.Parameter:
    """Relation table `[n_relation_type (x2 with inverse relations), Wr]`.

    The inverse of relation `i` is row `i + n_relation_type`.
    """
    if isinstance(initializer, torch.Tensor):
        if initializer.dim() != 2:
            raise ValueError("Table for initialization needs to be 2-dimensional")
        table = initializer.to(dtype=dtype or torch.float32, device=device)
        if row_size:
            assert (
                sum(row_size) == table.shape[-1]
            ), "Initialization tensor and row_size provided are incompatible"
    else:
        n_rows = n_relation_type * (2 if inverse_relations else 1)
        table = _from_initializers((n_rows,), initializer, row_size, device, dtype)
    return torch.nn.Parameter(table)


def refactor_embedding_sharding(
    entity_embedding: torch.nn.Parameter,
    old_sharding: Sharding,
    new_sharding: Sharding,
) -> torch.nn.Parameter:
    """Re-shard a `[n_old, M_old, W]` table to `[n_new, M_new, W]`."""
    flat = entity_embedding.detach()[
        torch.from_numpy(old_sharding.entity_to_shard),
        torch.from_numpy(old_sharding.entity_to_idx),
    ]
    return initialize_entity_embedding(initializer=flat, sharding=new_sharding)
