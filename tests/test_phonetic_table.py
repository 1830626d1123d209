"""``allophant_amd.phonetic.AttributeTable`` against the REAL reference ``PhoneticAttributeIndexer`` on a synthetic
Allophoible-format table (tests/golden/g6_phonetic_table.json, written by oracle/gen_phonetic_golden.py): composition
features, vocabularies, ``composition_feature_matrix``, ``phoneme_inventory`` and the embedding-table category counts."""
import json
import os

import pytest
import torch

from allophant_amd.phonetic import AttributeTable, to_iso6393

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g6_phonetic_table.json")


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN, encoding="utf-8") as f:
        return json.load(f)


def test_vocabularies_and_composition_features(golden):
    table = AttributeTable(golden["table"])
    assert table.phonemes == golden["phonemes"]
    assert table.composition_features == golden["composition_features"]
    assert "tone" not in table.composition_features
    for feature, categories in golden["feature_categories"].items():
        assert table.feature_categories(feature) == categories, feature


def test_inventories_follow_dialect_preference_and_size(golden):
    table = AttributeTable(golden["table"])
    for name, expected in golden["inventories"].items():
        codes = name.split("+")
        got = table.phoneme_inventory(codes if len(codes) > 1 else codes[0])
        assert got == expected, name
    # 'eng': the configured default dialect (10 phonemes) wins over the larger RP inventory (17)
    assert len(golden["inventories"]["eng"]) == 10
    # two-letter codes resolve like upstream's language-code standardisation
    assert table.phoneme_inventory(["es", "it"]) == golden["inventories"]["spa+ita"]
    assert to_iso6393("es") == "spa" and to_iso6393("deu") == "deu" and to_iso6393("pt-BR") == "por"
    with pytest.raises(ValueError):
        to_iso6393("q")


def test_composition_feature_matrix_is_bit_exact(golden):
    table = AttributeTable(golden["table"])
    for name, expected in golden["matrices"].items():
        inventory = ["kp", "a", "t͡s", "ŋ"] if name == "custom" else golden["inventories"][name]
        got = table.composition_feature_matrix(inventory)
        assert got.dtype == torch.int64 and got.shape == (len(inventory), len(table.composition_features))
        assert got.tolist() == expected, name
    with pytest.raises(ValueError, match="Missing phonemes"):
        table.composition_feature_matrix(["a", "not-a-phone"])


def test_category_counts_match_the_embedding_table_sizes(golden):
    table = AttributeTable(golden["table"])
    for name, expected in golden["category_counts"].items():
        # (row 0 of the embedding table is the blank embedding, acoustic_model.py:193-195; the offsets start after it)
        assert table.category_counts(golden["inventories"][name]) == expected, name


def _composition_checkpoint(golden, allophone_layer):
    from allophant_amd import checkpoint, spec as S, synthetic

    table = AttributeTable(golden["table"])
    training = golden["inventories"]["spa+ita"]
    spec = S.multitask_spec(S.tiny_encoder(1), ["syllabic", "nasal"], embedding_size=16, train_phonemes=len(training),
                            n_features=len(table.composition_features), allophone_layer=allophone_layer)
    spec["composition_categories"] = table.category_counts(training)
    if allophone_layer:
        spec["shared_phones"] = len(training)
    state = synthetic.make_state_dict(spec, seed=3)
    phonemes = training if not allophone_layer else training[:10]
    # LanguageAllophoneMappings dump (phonetic_features.py:40-44): per language, phoneme index -> shared-phone indices
    mapping = {str(lang): {str(phonemes.index(p)): [training.index(p)] for p in golden["inventories"][iso] if p in phonemes}
               for lang, iso in enumerate(["spa", "ita"])}
    indexer_state = {
        "phoneme_inventory": phonemes,
        "language_allophones": {"allophones": mapping, "languages": ["es", "it"], "shared_phones": training} if allophone_layer else None,
        "table_file": golden["table"],
    }
    return spec, state, checkpoint.make_checkpoint(spec, state, synthetic_encoder=True, indexer_state=indexer_state)


@pytest.mark.parametrize("allophone_layer", [False, True])
def test_checkpoint_layout_is_rebuilt_from_the_embedded_table(golden, allophone_layer):
    """Like upstream, the embedding-table layout of a composition checkpoint comes from ``phonetic_indexer_state`` (table
    text + training phones), not from a stored buffer."""
    from allophant_amd import checkpoint

    spec, state, ckpt = _composition_checkpoint(golden, allophone_layer)
    assert "amx_composition_categories" not in ckpt["additional"]
    rebuilt = checkpoint.spec_from_checkpoint(ckpt)
    assert rebuilt["composition_categories"] == spec["composition_categories"]
    table, training = checkpoint.indexer_from_checkpoint(ckpt)
    assert training == golden["inventories"]["spa+ita"]
    assert table.composition_feature_matrix(training).tolist() == golden["matrices"]["spa+ita"]
    # a table that does not match the stored embedding rows is rejected
    if ckpt["phonetic_indexer_state"]["language_allophones"]:
        ckpt["phonetic_indexer_state"]["language_allophones"]["shared_phones"] = ["a"]
    else:
        ckpt["phonetic_indexer_state"]["phoneme_inventory"] = ["a"]
    with pytest.raises(ValueError, match="attribute embeddings"):
        checkpoint.spec_from_checkpoint(ckpt)


def test_hypothesis_symbols_follow_the_reference_prediction_loop():
    """tokens - 1 -> inventory symbols on the IPA outputs, category strings on attribute outputs (run.py:776-806)."""
    from collections import namedtuple

    from allophant_amd.phonetic import AttributeTable, hypothesis_symbols

    Hyp = namedtuple("Hyp", "tokens timesteps score")
    with open(GOLDEN, encoding="utf-8") as f:
        table = AttributeTable(json.load(f)["table"])
    inventory = table.phonemes[:4]
    feature = table.composition_features[0]
    categories = table.feature_categories(feature)
    decoded = {
        "phoneme": [[Hyp(torch.tensor([2, 1, 3]), torch.tensor([1, 4, 6]), -1.0)], [Hyp(torch.tensor([], dtype=torch.int64), torch.tensor([]), 0.0)]],
        feature: [[Hyp(torch.tensor([1, len(categories)]), torch.tensor([1, 2]), -2.0)], [Hyp(torch.tensor([1]), torch.tensor([3]), -0.5)]],
    }
    symbols = hypothesis_symbols(decoded, inventory, table)
    assert symbols["phoneme"] == [[[inventory[1], inventory[0], inventory[2]]], [[]]]
    assert symbols[feature] == [[[categories[0], categories[-1]]], [[categories[0]]]]
    assert table.feature_values(feature, torch.tensor([0])) == [categories[0]]
    assert hypothesis_symbols({feature: decoded[feature]}, inventory)[feature][0][0] == ["0", str(len(categories) - 1)]


# ---- the indexer an allophone-layer checkpoint restores to (training-language restriction) ----
G9 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_restricted_indexer.json")


@pytest.fixture(scope="module")
def restored():
    with open(G9, encoding="utf-8") as f:
        return json.load(f)


def _restored_table(g9):
    state = g9["state"]
    return AttributeTable(state["table_file"], g9["attribute_subset"], state["phoneme_inventory"], state["language_allophones"])


def test_restored_indexer_is_restricted_to_the_training_languages(restored):
    """Against the REAL reference after its state() -> from_config(state_dict=...) round trip (golden g9, written by
    oracle/gen_restricted_golden.py): inventories only for the training languages, cut to the mapping's phonemes, phonemes
    the selected inventory lacks re-added, rows grouped by language code."""
    table = _restored_table(restored)
    for name, expected in restored["inventories"].items():
        codes = name.split("+")
        assert table.phoneme_inventory(codes if len(codes) > 1 else codes[0]) == expected, name
    assert restored["inventories"]["deu"] == []  # in the table, not a training language
    # the corpus inventory of Spanish lacked three table phonemes and brought two the table inventory does not list
    spa = restored["inventories"]["spa"]
    assert set(spa) == set(restored["corpus_inventories"]["spa"])
    unrestricted = AttributeTable(restored["state"]["table_file"])
    assert unrestricted.phoneme_inventory("spa") != spa and unrestricted.phoneme_inventory("deu")
    assert table.phonemes == restored["phonemes"] and table.feature_names == restored["feature_names"]
    assert table.shared_phones == restored["shared_phones"]


def test_training_matrix_of_a_restored_checkpoint(restored):
    """The matrix `predict(batch)` falls back to without target_feature_indices: `_dense_feature_table` of the reference's
    composition layer minus its category offsets == composition_feature_matrix(shared phones)."""
    table = _restored_table(restored)
    training = table.composition_feature_matrix(table.shared_phones)
    assert training.tolist() == restored["training_matrix"]
    counts = table.category_counts(table.shared_phones)
    offsets = [1]
    for c in counts[:-1]:
        offsets.append(offsets[-1] + c)
    assert offsets == restored["category_offsets"] and 1 + sum(counts) == restored["embedding_rows"]


def test_indexer_from_checkpoint_applies_the_restriction(restored):
    from allophant_amd import checkpoint

    ckpt = {"phonetic_indexer_state": restored["state"],
            "config": {"nn": {"projection": {"classes": [
                {"name": "phoneme", "dependencies": ["OUTPUT", "syllabic"]}, {"name": "syllabic", "dependencies": ["OUTPUT_3"]},
                {"name": "long", "dependencies": ["OUTPUT"]}, {"name": "nasal", "dependencies": ["OUTPUT"]}]}}}}
    table, training = checkpoint.indexer_from_checkpoint(ckpt)
    assert training == restored["shared_phones"] and table.feature_names == restored["feature_names"]
    assert table.phoneme_inventory(["spa", "ita"]) == restored["inventories"]["spa+ita"]
    # a checkpoint without a mapping restores an unrestricted indexer (phonetic_features.py:765-775)
    plain = dict(ckpt, phonetic_indexer_state={"phoneme_inventory": ["a", "b"], "language_allophones": None,
                                               "table_file": restored["state"]["table_file"]})
    table2, training2 = checkpoint.indexer_from_checkpoint(plain)
    assert training2 == ["a", "b"] and table2.phoneme_inventory("deu")
    # README decode loop: inventory view
    view = table.subset(restored["inventories"]["spa"])
    assert view.feature_values("phoneme", [0, 2]) == [restored["inventories"]["spa"][0], restored["inventories"]["spa"][2]]
    assert view.feature_values("syllabic", [0]) == [table.feature_categories("syllabic")[0]]
    with pytest.raises(ValueError, match="Missing phonemes"):
        table.subset(["a", "nope"])
