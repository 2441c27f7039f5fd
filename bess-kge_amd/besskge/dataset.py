"""Knowledge-graph dataset container (plain host plumbing).

Mirrors the fields of the reference dataclass (`besskge/dataset.py:23-81`) and
its `from_triples` splitter (`dataset.py:83-145`).  The downloaders for
OGB / YAGO / OpenBioLink are out of scope (host I/O,
no network on the build or GPU boxes).
"""

import dataclasses
import pickle
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
from numpy.typing import NDArray


@dataclasses.dataclass
class KGDataset:
    """A set of (head, relation, tail) triples split into named parts."""

    #: number of entities
    n_entity: int
    #: number of relation types
    n_relation_type: int
    #: {part: int[n_triple, 3]} columns (h, r, t), global IDs
    triples: Dict[str, NDArray[np.int32]]
    #: {part: int[n_triple]} position of each triple in the source array
    original_triple_ids: Dict[str, NDArray[np.int32]]
    #: optional labels
    entity_dict: Optional[List[str]] = None
    relation_dict: Optional[List[str]] = None
    #: {type label: first entity ID of the type}; IDs are clustered by type
    type_offsets: Optional[Dict[str, int]] = None
    #: {part: int[n_triple or 1, n_neg]} candidate corrupted heads / tails
    neg_heads: Optional[Dict[str, NDArray[np.int32]]] = None
    neg_tails: Optional[Dict[str, NDArray[np.int32]]] = None

    @property
    def ht_types(self) -> Optional[Dict[str, NDArray[np.int32]]]:
        """{part: int[n_triple, 2]} type IDs of each triple's head and tail."""
        if not self.type_offsets:
            return None
        bounds = np.fromiter(self.type_offsets.values(), dtype=np.int32)
        return {
            part: np.digitize(trip[:, [0, 2]], bounds) - 1
            for part, trip in self.triples.items()
        }

    @classmethod
    def from_triples(
        cls,
        data: NDArray[np.int32],
        split: Tuple[float, float, float] = (0.7, 0.15, 0.15),
        seed: int = 1234,
        entity_dict: Optional[List[str]] = None,
        relation_dict: Optional[List[str]] = None,
        type_offsets: Optional[Dict[str, int]] = None,
    ) -> "KGDataset":
        """Random train/valid/test split of an (n, 3) array of ID triples."""
        n = data.shape[0]
        n_train = int(n * split[0])
        n_valid = int(n * split[1])
        order = np.random.default_rng(seed=seed).permutation(np.arange(n))
        parts = dict(
            zip(
                ("train", "valid", "test"),
                np.split(order, (n_train, n_train + n_valid), axis=0),
            )
        )
        return cls(
            n_entity=data[:, [0, 2]].max() + 1,
            n_relation_type=data[:, 1].max() + 1,
            entity_dict=entity_dict,
            relation_dict=relation_dict,
            type_offsets=type_offsets,
            triples={k: data[v] for k, v in parts.items()},
            original_triple_ids=parts,
        )

    def save(self, out_file: Path) -> None:
        """Pickle the dataset."""
        with open(out_file, "wb") as f:
            pickle.dump(self, f)

    @classmethod
    def load(cls, path: Path) -> "KGDataset":
        """Load a dataset written by :meth:`save`."""
        with open(path, "rb") as f:
            obj = pickle.load(f)
        if not isinstance(obj, cls):
            raise ValueError(f"{path} does not hold a {cls.__name__}")
        return obj
