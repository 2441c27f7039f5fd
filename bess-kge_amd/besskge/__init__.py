"""besskge for AMD Instinct MI355X (gfx950).

A from-scratch, MI355X-native implementation of the BESS hot path of
graphcore-research/bess-kge - per-shard embedding gather, all-to-all of
tail/negative embeddings, fused score + loss for TransE / RotatE / DistMult /
ComplEx, sparse gradient scatter back into the shard - behind the reference's
own Python API (`besskge.bess`, `besskge.scoring`, `besskge.embedding`,
`besskge.sharding`, `besskge.batch_sampler`, `besskge.negative_sampler`,
`besskge.loss`).

Like the reference (which dlopens its PopART custom-op library at import,
reference `besskge/__init__.py:10-37`), importing the package loads the native
library - here `libbesskge_hip.so`, hand-written HIP kernels behind a C ABI
(`include/besskge_hip.h`) - and fails with ImportError if it is missing.
"""

from . import _native

_native.load()


def load_custom_ops_so() -> None:
    """Name kept from the reference (`besskge/__init__.py:10-34`): (re)load the native library."""
    _native.load()


from . import (  # noqa: E402,F401
    batch_sampler,
    bess,
    collectives,
    dataset,
    device_sampler,
    embedding,
    loss,
    metric,
    negative_sampler,
    pipeline,
    query,
    runtime,
    scoring,
    sharding,
    utils,
)

__all__ = [
    "batch_sampler",
    "bess",
    "collectives",
    "dataset",
    "device_sampler",
    "embedding",
    "loss",
    "metric",
    "negative_sampler",
    "pipeline",
    "query",
    "runtime",
    "scoring",
    "sharding",
    "utils",
]
