"""KGDataset.from_dataframe / from_triples against the reference's outputs
(tests/golden/dataset.npz): same entity / relation IDs, type blocks and split."""

import numpy as np
import pytest

from besskge.dataset import KGDataset

from conftest import load_golden


@pytest.mark.parametrize("case", ["plain", "typed", "parts"])
def test_from_dataframe_matches_reference(case):
    pd = pytest.importorskip("pandas")
    g = load_golden("dataset")
    df = pd.DataFrame(dict(head=g["h"], relation=g["r"], tail=g["t"]))
    types = dict(zip(g["ents"].tolist(), g["kinds"].tolist()))
    if case == "plain":
        ds = KGDataset.from_dataframe(df, "head", "relation", "tail")
    elif case == "typed":
        ds = KGDataset.from_dataframe(df, "head", "relation", "tail", entity_types=types, split=(0.6, 0.2, 0.2), seed=7)
    else:
        ds = KGDataset.from_dataframe({"train": df.iloc[:500], "test": df.iloc[500:]}, "head", "relation", "tail",
                                      entity_types=types)
    assert [ds.n_entity, ds.n_relation_type] == g[f"{case}_n"].tolist()
    assert ds.entity_dict == g[f"{case}_entity_dict"].tolist()
    assert ds.relation_dict == g[f"{case}_relation_dict"].tolist()
    if case == "plain":
        assert ds.type_offsets is None
    else:
        assert list(ds.type_offsets.keys()) == g[f"{case}_type_names"].tolist()
        assert list(ds.type_offsets.values()) == g[f"{case}_type_firsts"].tolist()
        # every type is a contiguous block of IDs
        firsts = list(ds.type_offsets.values()) + [ds.n_entity]
        for (name, lo), hi in zip(ds.type_offsets.items(), firsts[1:]):
            assert all(types[ds.entity_dict[i]] == name for i in range(lo, hi))
    parts = sorted(k[len(case) + 9:] for k in g.files if k.startswith(f"{case}_triples_"))
    assert sorted(ds.triples) == parts
    for part in parts:
        assert np.array_equal(ds.triples[part], g[f"{case}_triples_{part}"])
        assert np.array_equal(ds.original_triple_ids[part], g[f"{case}_ids_{part}"])


def test_downloaders_say_what_to_do(tmp_path):
    for name in ("build_ogbl_biokg", "build_ogbl_wikikg2", "build_yago310", "build_openbiolink"):
        with pytest.raises(RuntimeError, match="from_triples"):
            getattr(KGDataset, name)(tmp_path)


def test_save_load_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    tr = np.stack([rng.integers(50, size=200), rng.integers(4, size=200), rng.integers(50, size=200)], axis=1)
    ds = KGDataset.from_triples(tr, type_offsets={"a": 0, "b": 20})
    ds.save(tmp_path / "ds.pkl")
    back = KGDataset.load(tmp_path / "ds.pkl")
    assert back.n_entity == ds.n_entity and back.type_offsets == ds.type_offsets
    assert all(np.array_equal(back.triples[k], ds.triples[k]) for k in ds.triples)
    assert np.array_equal(back.ht_types["train"], np.digitize(ds.triples["train"][:, [0, 2]], [0, 20]) - 1)
