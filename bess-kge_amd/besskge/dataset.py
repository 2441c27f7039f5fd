"""Knowledge-graph dataset container (plain host plumbing).

Mirrors the fields of the reference dataclass (`besskge/dataset.py:23-81`), its
`from_triples` splitter (`dataset.py:83-145`) and `from_dataframe` label
encoder (`dataset.py:147-239`).  The downloaders for OGB / YAGO / OpenBioLink
(`dataset.py:241-459`) need packages and a network that neither the build nor
the GPU box has: they raise with a pointer to `from_triples` / `from_dataframe`.
"""

import dataclasses
import pickle
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

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

    @classmethod
    def from_dataframe(
        cls,
        df: Any,
        head_column: Union[int, str],
        relation_column: Union[int, str],
        tail_column: Union[int, str],
        entity_types: Optional[Any] = None,
        split: Tuple[float, float, float] = (0.7, 0.15, 0.15),
        seed: int = 1234,
    ) -> "KGDataset":
        """Dataset from a pandas DataFrame (or {part: DataFrame}) of labelled triples.

        Entities and relations get IDs in order of first appearance; with
        `entity_types` (label -> type label) the entity IDs are re-assigned so
        that every type is a contiguous block (types in label order), and
        `type_offsets` is set.
        A single DataFrame is split at random like :meth:`from_triples`.
        """
        import pandas as pd

        parts = {"all": df} if isinstance(df, pd.DataFrame) else dict(df)
        ent_labels = pd.unique(pd.concat(
            [pd.concat([p[head_column], p[tail_column]]) for p in parts.values()]))
        rel_labels = pd.unique(pd.concat([p[relation_column] for p in parts.values()]))
        type_offsets = None
        if entity_types is not None:
            # pandas' default (unstable) sort decides the order inside a type: use the same call as
            # the reference (dataset.py:205-208) so that entity IDs come out identical
            kinds = pd.merge(pd.Series(np.arange(len(ent_labels)), index=ent_labels, name="first_id"),
                             pd.Series(entity_types, name="kind"), how="left", left_index=True,
                             right_index=True).sort_values("kind")
            ent_labels = kinds.index.to_numpy()
            counts = kinds.groupby("kind")["kind"].count()
            firsts = np.concatenate([[0], np.cumsum(counts.to_numpy())[:-1]])
            type_offsets = {str(n): int(f) for n, f in zip(counts.index, firsts)}
        ent_id = pd.Series(np.arange(len(ent_labels)), index=ent_labels)
        rel_id = pd.Series(np.arange(len(rel_labels)), index=rel_labels)
        triples = {
            name: np.stack([p[head_column].map(ent_id).to_numpy(), p[relation_column].map(rel_id).to_numpy(),
                            p[tail_column].map(ent_id).to_numpy()], axis=1).astype(np.int32)
            for name, p in parts.items()
        }
        if isinstance(df, pd.DataFrame):
            return cls.from_triples(triples["all"], split, seed, list(ent_labels), list(rel_labels), type_offsets)
        return cls(
            n_entity=len(ent_labels), n_relation_type=len(rel_labels), entity_dict=list(ent_labels),
            relation_dict=list(rel_labels), type_offsets=type_offsets, triples=triples,
            original_triple_ids={k: np.arange(v.shape[0]) for k, v in triples.items()},
        )

    @classmethod
    def _needs_download(cls, what: str) -> "KGDataset":
        raise RuntimeError(
            f"{what} downloads and unpacks its source files (reference besskge/dataset.py:241-459); this build"
            " has no downloader - load the triples yourself and use KGDataset.from_triples / from_dataframe")

    @classmethod
    def build_ogbl_biokg(cls, root: Path) -> "KGDataset":
        return cls._needs_download("build_ogbl_biokg")

    @classmethod
    def build_ogbl_wikikg2(cls, root: Path) -> "KGDataset":
        return cls._needs_download("build_ogbl_wikikg2")

    @classmethod
    def build_yago310(cls, root: Path) -> "KGDataset":
        return cls._needs_download("build_yago310")

    @classmethod
    def build_openbiolink(cls, root: Path) -> "KGDataset":
        return cls._needs_download("build_openbiolink")

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
