"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/g10_inventory_mapping.json from the REAL reference
``PhoneticAttributeIndexer`` (allophant/phonetic_features.py, imported through oracle/ref_import.py) on the synthetic
Allophoible-format table of oracle/gen_phonetic_golden.py:

  * nearest-phone inventory mappings: ``map_target_inventory`` (the "tr2tgt" scheme, :925-971; run.py:286-294),
    ``map_to_subset`` (:907-917), ``ArticulatoryAttributes.map_inventories_closest`` (:355-445) with and without splitting
    of complex segments and with a distance threshold, ``map_language_inventory`` (:858-897);
  * the macro-language fallback of ``extract_allophone_inventories`` (:1092-1136): a training language without an inventory
    of its own takes the inventory of a table language inside the same macro language (here: ``est`` -> ``ekk``), through
    the save -> restore round trip of an allophone-layer checkpoint.

``LanguageCode.from_str`` needs the absent `langcodes` package: replaced by a stand-in that takes ISO 639-3 codes as they
are and resolves macro languages through the small table below (what `langcodes.standardize_tag(code, macro=True)` does
for these codes); ``phoneme_segmentation.base_phonemes`` / ``split_complex_segment`` are the reference's own pure-Python
functions (the Rust ``IpaSegmenter`` is not involved in these paths).
"""
import csv
import io
import json
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

ref_import.install()

import gen_phonetic_golden as G6  # noqa: E402  (make_table; installs the standardize_to_iso6393 identity)
from allophant import phonetic_features as pf  # noqa: E402
from allophant.config import FeatureSet  # noqa: E402

MACRO = {"ekk": "est", "vro": "est", "cmn": "zho", "yue": "zho", "arb": "ara", "arz": "ara"}


class _Code:
    def __init__(self, code, macro=False):
        resolved = MACRO.get(code, code) if macro else code
        self.alpha3 = self.alpha3_t = self.alpha3_b = resolved

    @classmethod
    def from_str(cls, code, standardize=False, macro=False):
        return cls(code, macro)


pf.LanguageCode = _Code


def _rename_language(text, old, new):
    rows = list(csv.reader(io.StringIO(text)))
    col = rows[0].index("ISO6393")
    for r in rows[1:]:
        if r[col] == old:
            r[col] = new
    buf = io.StringIO()
    csv.writer(buf, lineterminator="\n").writerows(rows)
    return buf.getvalue()


def _try(fn):
    try:
        return {"result": fn()}
    except Exception as error:  # recorded: the port must fail the same way
        return {"error": type(error).__name__}


def main():
    warnings.simplefilter("ignore")
    text = G6.make_table()
    full = pf.PhoneticAttributeIndexer(FeatureSet.PHOIBLE, text, allophones_from_allophoible=True)
    inv = {name: full.phoneme_inventory(name) for name in ("spa", "ita", "deu", "eng")}
    custom = ["kp", "a", "t͡s", "ŋ", "ai"]
    golden = {"table": text, "inventories": inv, "custom": custom, "map_target": {}, "map_to_subset": {}, "closest": [],
              "language": {}}
    # ---- a model indexer over all features whose phonemes are the Spanish + Italian inventory ----
    model_phonemes = full.phoneme_inventory(["spa", "ita"])
    model = pf.PhoneticAttributeIndexer(FeatureSet.PHOIBLE, text, None, model_phonemes, None, True)
    golden["model_phonemes"] = model.phonemes.tolist()
    for name, target in (("deu", inv["deu"]), ("eng", inv["eng"]), ("custom", custom)):
        golden["map_target"][name] = {
            "uncovered": model.map_target_inventory(target),
            "plain": model.map_target_inventory(target, map_uncovered_target_phonemes=False),
        }
        golden["map_to_subset"][name] = model.map_to_subset(target)
    # with an attribute subset the feature vectors of the model and of the target inventory have different widths
    subset_model = pf.PhoneticAttributeIndexer(FeatureSet.PHOIBLE, text, ["phoneme", "syllabic", "long", "nasal"], model_phonemes, None, True)
    golden["map_target_with_attribute_subset"] = {
        "deu": _try(lambda: subset_model.map_target_inventory(inv["deu"])),
        "own": _try(lambda: subset_model.map_target_inventory(model_phonemes[:6])),
    }
    # ---- map_inventories_closest ----
    attributes = full.full_attributes
    for source, target in ((inv["deu"], inv["spa"]), (custom, inv["ita"]), (inv["eng"], custom), (["ai", "kp", "aː"], ["a", "i", "t", "s", "k", "p"]), (["ai", "t͡s", "kp", "aː"], ["a", "i", "t", "s", "k", "p"])):
        for split in (False, True):
            for threshold in (None, 2, 6):
                outcome = _try(lambda: attributes.map_inventories_closest(source, target, split_non_matching_complex=split,
                                                                          distance_threshold=threshold))
                golden["closest"].append({"source": source, "target": target, "split": split, "threshold": threshold, **outcome})
    # ---- map_language_inventory ----
    for language in ("spa", "deu"):
        golden["language"][language] = {
            "plain": _try(lambda: full.map_language_inventory([inv["ita"], ["ai", "aː", "ŋ", "kp"]], language)),
            "threshold3": _try(lambda: full.map_language_inventory([inv["eng"]], language, distance_threshold=3)),
        }
    # ---- macro-language fallback through a checkpoint's indexer state ----
    text_est = _rename_language(text, "ita", "ekk")
    table_est = pf.PhoneticAttributeIndexer(FeatureSet.PHOIBLE, text_est, allophones_from_allophoible=True)
    spa, ekk = table_est.phoneme_inventory("spa"), table_est.phoneme_inventory("ekk")
    corpus = {0: spa[:-2], 1: ekk[1:]}
    languages = ["spa", "est"]  # `est` has no inventory: its macro-language sibling `ekk` stands in (:1092-1128)
    inventories = pf.LanguageInventories(corpus, languages)
    attribute_subset = ["phoneme", "syllabic", "long", "nasal"]
    training = pf.PhoneticAttributeIndexer(FeatureSet.PHOIBLE, text_est, attribute_subset, sorted(inventories.shared_inventory()),
                                           inventories, True)
    state = training.state()
    restored = pf.PhoneticAttributeIndexer(FeatureSet.PHOIBLE, state.table_file, attribute_subset, state.phoneme_inventory,
                                           state.language_allophones, True)
    golden["macro"] = {
        "macro_table": MACRO,
        "state": {
            "phoneme_inventory": state.phoneme_inventory,
            "language_allophones": {
                "allophones": {str(lang): {str(k): list(map(int, v)) for k, v in phones.items()}
                               for lang, phones in state.language_allophones.allophones.items()},
                "languages": state.language_allophones.languages,
                "shared_phones": state.language_allophones.shared_phones,
            },
            "table_file": state.table_file,
        },
        "attribute_subset": attribute_subset,
        "inventories": {code: restored.phoneme_inventory(code) for code in ("spa", "est", "ekk", "deu")},
        "union": restored.phoneme_inventory(["spa", "est"]),
        "shared_phones": restored.allophone_data.shared_phone_indexer.phonemes.tolist(),
    }
    # a language with neither an inventory nor a macro-language sibling is refused
    bad = pf.LanguageInventories({0: spa[:-2], 1: ekk[1:]}, ["spa", "fin"])
    golden["macro"]["unresolvable"] = _try(lambda: pf.PhoneticAttributeIndexer(
        FeatureSet.PHOIBLE, text_est, attribute_subset, sorted(bad.shared_inventory()), bad, True))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g10_inventory_mapping.json")
    with open(out, "w", encoding="utf-8") as f:
        json.dump(golden, f, ensure_ascii=False, indent=0)
    print("wrote", out)
    print("map_target deu", golden["map_target"]["deu"]["uncovered"])
    print("subset model", golden["map_target_with_attribute_subset"])
    print("closest[0]", golden["closest"][0])
    print("closest split", [c.get("result", c.get("error")) for c in golden["closest"] if c["source"][0] == "ai" and c["split"]][:2])
    print("errors", sum("error" in c for c in golden["closest"]), "of", len(golden["closest"]))
    print("language", {k: {kk: ("error" if "error" in vv else "ok") for kk, vv in v.items()} for k, v in golden["language"].items()})
    print("macro inventories", {k: len(v) for k, v in golden["macro"]["inventories"].items()}, golden["macro"]["unresolvable"])


if __name__ == "__main__":
    main()
