"""Nearest-phone inventory mappings and the macro-language fallback of ``allophant_amd.phonetic.AttributeTable`` against
the REAL reference ``PhoneticAttributeIndexer`` (tests/golden/g10_inventory_mapping.json, written by
oracle/gen_mapping_golden.py on the synthetic Allophoible-format table): ``map_target_inventory`` ("tr2tgt",
phonetic_features.py:925-971; the evaluation loop run.py:286-294), ``map_to_subset`` (:907-917),
``map_inventories_closest`` (:355-445) with complex-segment splitting and distance thresholds -- including the cases the
reference refuses (a split part that is no table phoneme -> ValueError; an attribute-subset indexer against the full
feature set -> RuntimeError from the width mismatch) --, ``map_language_inventory`` (:858-897), and a restored indexer whose
training language ``est`` has no inventory and takes the one of ``ekk`` (:1092-1136)."""
import json
import os

import pytest

from allophant_amd.phonetic import MACROLANGUAGES, AttributeTable, base_phonemes, split_complex_segment

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_inventory_mapping.json")


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN, encoding="utf-8") as f:
        return json.load(f)


def _outcome(fn):
    try:
        return {"result": fn()}
    except Exception as error:
        return {"error": type(error).__name__}


def test_map_target_inventory_and_map_to_subset(golden):
    table = AttributeTable(golden["table"], None, golden["model_phonemes"])
    assert table.phonemes == golden["model_phonemes"]
    for name, expected in golden["map_target"].items():
        target = golden["custom"] if name == "custom" else golden["inventories"][name]
        assert table.map_target_inventory(target, map_uncovered_target_phonemes=False) == expected["plain"], name
        mine = table.map_target_inventory(target)
        # every phone of the target inventory is covered after the second phase, every model phoneme is mapped
        assert set(mine) == set(golden["model_phonemes"]) and set(mine.values()) >= set(target) - set(), name
        assert mine == expected["uncovered"], name
        assert table.map_to_subset(target) == golden["map_to_subset"][name], name
    with pytest.raises(NotImplementedError):
        table.map_target_inventory(golden["custom"], missing_feature_fallback=True)


def test_attribute_subset_indexer_fails_like_the_reference(golden):
    table = AttributeTable(golden["table"], ["phoneme", "syllabic", "long", "nasal"], golden["model_phonemes"])
    cases = golden["map_target_with_attribute_subset"]
    assert _outcome(lambda: table.map_target_inventory(golden["inventories"]["deu"])) == cases["deu"]
    assert _outcome(lambda: table.map_target_inventory(golden["model_phonemes"][:6])) == cases["own"]


def test_map_inventories_closest(golden):
    table = AttributeTable(golden["table"])
    split_cases = 0
    for case in golden["closest"]:
        expected = {k: case[k] for k in ("result", "error") if k in case}
        got = _outcome(lambda: table.map_inventories_closest(case["source"], case["target"], case["split"], case["threshold"]))
        assert got == expected, (case["source"], case["split"], case["threshold"])
        if "result" in case:
            split_cases += any(len(v) > 1 for v in case["result"].values())
    assert split_cases >= 2  # complex segments really were split in some cases
    assert sum("error" in c for c in golden["closest"]) >= 1  # and the refusal path is covered


def test_map_language_inventory(golden):
    table = AttributeTable(golden["table"])
    for language, cases in golden["language"].items():
        inventories = [golden["inventories"]["ita"], ["ai", "aː", "ŋ", "kp"]]
        assert _outcome(lambda: table.map_language_inventory(inventories, language)) == cases["plain"], language
        assert _outcome(lambda: table.map_language_inventory([golden["inventories"]["eng"]], language, distance_threshold=3)) == \
            cases["threshold3"], language


def test_segment_helpers():
    assert base_phonemes("t͡s") == ["t", "s"] and base_phonemes("aː") == ["a"] and base_phonemes("kp") == ["k", "p"]
    assert split_complex_segment("ai") == ["a", "i"] and split_complex_segment("t͡s") == ["t͡", "s"]
    assert split_complex_segment("aː") == ["aː"] and split_complex_segment("ˈa") == ["ˈa"]


def test_macro_language_fallback_of_a_restored_indexer(golden):
    macro = golden["macro"]
    assert all(MACROLANGUAGES[k] == v for k, v in macro["macro_table"].items())
    state = macro["state"]
    restored = AttributeTable(state["table_file"], macro["attribute_subset"], state["phoneme_inventory"], state["language_allophones"])
    for code, expected in macro["inventories"].items():
        assert restored.phoneme_inventory(code) == expected, code
    assert restored.phoneme_inventory(["spa", "est"]) == macro["union"]
    assert restored.phoneme_inventory("et") == macro["inventories"]["est"]  # ISO 639-1 resolves like upstream's standardisation
    assert restored.shared_phones == macro["shared_phones"]
    # a training language with neither an inventory nor a macro-language sibling in the table is refused like upstream
    bad = dict(state["language_allophones"], languages=["spa", "fin"])
    assert macro["unresolvable"] == {"error": "ValueError"}
    with pytest.raises(ValueError, match="don't contain allophone data"):
        AttributeTable(state["table_file"], macro["attribute_subset"], state["phoneme_inventory"], bad)
    # and a caller-supplied macro table extends the built-in one
    custom = AttributeTable(state["table_file"], macro["attribute_subset"], state["phoneme_inventory"], bad, macrolanguages={"fin": "est"})
    assert custom.phoneme_inventory("fin") == macro["inventories"]["est"]
