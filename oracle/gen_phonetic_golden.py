"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/g6_phonetic_table.json: a synthetic Allophoible-format table and
what the REAL reference ``PhoneticAttributeIndexer`` (allophant/phonetic_features.py, imported through
oracle/ref_import.py) derives from it -- ``composition_features``, feature vocabularies, ``composition_feature_matrix``
for several inventories, ``phoneme_inventory`` per language and the per-feature category counts that
``EmbeddingCompositionLayer.__init__`` computes (acoustic_model.py:191-207).  The real allophoible.csv is not part of
the reference snapshot (.MISSING_LARGE_BLOBS), so the table is generated here (seeded): 4 languages, one of them with
two dialects of which the reference's default_dialects.json prefers the smaller one, contour features on complex
segments, marginal phonemes and an inventory without allophone data.

``language_codes.standardize_to_iso6393`` needs the `langcodes` package (absent: stubbed), so it is replaced by the
identity and the goldens use ISO 639-3 codes.
"""
import csv
import io
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

ref_import.install()

import torch  # noqa: E402
from allophant import language_codes, phonetic_features as pf  # noqa: E402
from allophant.config import FeatureSet  # noqa: E402
from allophant.network.acoustic_model import EmbeddingCompositionLayer  # noqa: E402

language_codes.standardize_to_iso6393 = lambda code: code

FEATURES = ["tone", "stress", "syllabic", "short", "long", "consonantal", "sonorant", "continuant", "delayedRelease",
            "nasal", "labial", "round", "coronal", "dorsal", "high", "low"]
COLUMNS = ["InventoryID", "Glottocode", "ISO6393", "LanguageName", "SpecificDialect", "GlyphID", "Phoneme", "Allophones",
           "Marginal", "SegmentClass", "Source"] + FEATURES
PHONES = ["a", "e", "i", "o", "u", "p", "b", "t", "d", "k", "g", "m", "n", "s", "z", "l", "r", "t͡s", "ai", "ŋ",
          "ʃ", "aː", "kp"]


def make_table(seed=7):
    rng = random.Random(seed)
    contours = {}
    for ph in PHONES:
        vals = []
        for f in FEATURES:
            if f == "tone":
                vals.append("0")
                continue
            v = rng.choice(["+", "-", "0"])
            if ph in ("t͡s", "ai", "kp") and rng.random() < 0.5:
                v = v + "," + rng.choice(["+", "-"])
            vals.append(v)
        contours[ph] = vals
    rows = []
    inventory = 1
    # ("ita", "NoAllophones") is the largest Italian inventory but carries no allophone data: never selected (:1069)
    plan = [("spa", [("", 13)]), ("ita", [("", 15), ("NoAllophones", 19)]), ("deu", [("", 12), ("Northern", 9)]),
            ("eng", [("Western and Mid-Western US; Southern California", 10), ("RP", 17)])]
    for iso, dialects in plan:
        for dialect, n in dialects:
            chosen = rng.sample(PHONES, n)
            for j, ph in enumerate(chosen):
                allophones = ph if rng.random() < 0.6 else ph + " " + rng.choice(PHONES)
                marginal = "TRUE" if j % 6 == 5 else rng.choice(["", "FALSE"])
                if dialect == "NoAllophones":
                    allophones = ""
                rows.append([str(inventory), "glot" + iso, iso, "Language " + iso, dialect, "G%d" % len(rows), ph, allophones,
                             marginal, "vowel" if ph[0] in "aeiou" else "consonant", "src%d" % inventory] + contours[ph])
            inventory += 1
    buf = io.StringIO()
    w = csv.writer(buf, lineterminator="\n")
    w.writerow(COLUMNS)
    w.writerows(rows)
    return buf.getvalue()


def main():
    text = make_table()
    indexer = pf.PhoneticAttributeIndexer(FeatureSet.PHOIBLE, text, allophones_from_allophoible=True)
    full = indexer.full_attributes
    golden = {
        "table": text,
        "phonemes": full.phonemes.tolist(),
        "composition_features": indexer.composition_features,
        "feature_categories": {f: full.feature_categories(f) for f in indexer.composition_features},
        "matrices": {},
        "inventories": {},
        "category_counts": {},
    }
    inventories = {"spa": ["spa"], "ita": ["ita"], "spa+ita": ["spa", "ita"], "deu": ["deu"], "eng": ["eng"]}
    for name, codes in inventories.items():
        inv = indexer.phoneme_inventory(codes if len(codes) > 1 else codes[0])
        golden["inventories"][name] = inv
        golden["matrices"][name] = indexer.composition_feature_matrix(inv).tolist()
        # sizes of the embedding table a model trained on this inventory would have (acoustic_model.py:191-207)
        training = full.subset(inv, indexer.composition_features.copy())
        layer = EmbeddingCompositionLayer(8, training)
        offsets = layer._category_offsets.view(-1).tolist()
        rows_total = layer._attribute_embeddings.weight.shape[0]
        counts = [b - a for a, b in zip(offsets, offsets[1:] + [rows_total])]
        golden["category_counts"][name] = counts
    golden["matrices"]["custom"] = indexer.composition_feature_matrix(["kp", "a", "t͡s", "ŋ"]).tolist()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g6_phonetic_table.json")
    with open(out, "w", encoding="utf-8") as f:
        json.dump(golden, f, ensure_ascii=False, indent=0)
    print("wrote", out, {k: len(v) for k, v in golden["inventories"].items()})


if __name__ == "__main__":
    main()
