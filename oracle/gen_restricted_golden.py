"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/g9_restricted_indexer.json: what the REAL reference
``PhoneticAttributeIndexer`` looks like after the save -> restore round trip of an allophone-layer checkpoint
(``Estimator.restore``, allophant/estimator.py:1085-1126):

  training:  ``PhoneticAttributeIndexer.from_config(config, table, LanguageInventories)`` (phonetic_features.py:746-786,
             language inventories = the phonemes the training corpora use per language) -> ``state()`` (:111-115, 728-729)
  restore:   ``from_config(config, state_dict=state)`` -> ``cls(feature_set, state.table_file, attribute_subset,
             state.phoneme_inventory, state.language_allophones, True)``: the inventories are restricted to the training
             languages and re-filtered to the phonemes the mapping lists per language, phonemes the selected inventory
             lacks are re-added from the rest of the table (``extract_allophone_inventories`` / ``_filter_inventory``,
             :1040-1160)

Recorded: the dumped state, the restored indexer's ``phoneme_inventory`` per language / language set, its phoneme list,
the shared phones and the training feature matrix the ``EmbeddingCompositionLayer`` is built from
(acoustic_model.py:422-446, 191-217) -- i.e. what ``predict(batch)`` uses when no ``target_feature_indices`` are given.

The table is the synthetic Allophoible-format table of oracle/gen_phonetic_golden.py.  ``LanguageCode.from_str`` needs
the absent `langcodes` package: replaced by a stand-in that takes ISO 639-3 codes as they are (the goldens use 639-3).
"""
import json
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

ref_import.install()

import gen_phonetic_golden as G6  # noqa: E402  (make_table; also installs the standardize_to_iso6393 identity)
from allophant import phonetic_features as pf  # noqa: E402
from allophant.config import FeatureSet  # noqa: E402
from allophant.network.acoustic_model import EmbeddingCompositionLayer  # noqa: E402


class _Code:
    def __init__(self, code):
        self.alpha3 = self.alpha3_t = self.alpha3_b = code

    @classmethod
    def from_str(cls, code, *_args):
        return cls(code)


pf.LanguageCode = _Code


def main():
    warnings.simplefilter("ignore")
    text = G6.make_table()
    full = pf.PhoneticAttributeIndexer(FeatureSet.PHOIBLE, text, allophones_from_allophoible=True)
    spa = full.phoneme_inventory("spa")
    ita = full.phoneme_inventory("ita")
    # corpus inventories: a subset of each language's table inventory plus phonemes the selected inventory lacks (they
    # come back through `_filter_inventory`'s "remaining" branch)
    extra_spa = [p for p in G6.PHONES if p not in spa][:2]
    extra_ita = [p for p in G6.PHONES if p not in ita][:1]
    corpus = {0: spa[:-3] + extra_spa, 1: ita[1:] + extra_ita}
    languages = ["spa", "ita"]
    inventories = pf.LanguageInventories(corpus, languages)
    attribute_subset = ["phoneme", "syllabic", "long", "nasal"]
    training = pf.PhoneticAttributeIndexer(FeatureSet.PHOIBLE, text, attribute_subset, sorted(inventories.shared_inventory()),
                                           inventories, True)
    state = training.state()
    restored = pf.PhoneticAttributeIndexer(FeatureSet.PHOIBLE, state.table_file, attribute_subset, state.phoneme_inventory,
                                           state.language_allophones, True)
    shared = restored.allophone_data.shared_phone_indexer
    layer = EmbeddingCompositionLayer(8, shared)
    offsets = layer._category_offsets.view(-1)
    golden = {
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
        "corpus_inventories": {languages[k]: v for k, v in corpus.items()},
        "attribute_subset": attribute_subset,
        "phonemes": restored.phonemes.tolist(),
        "feature_names": restored.feature_names,
        "inventories": {
            "spa": restored.phoneme_inventory("spa"),
            "ita": restored.phoneme_inventory("ita"),
            "spa+ita": restored.phoneme_inventory(["spa", "ita"]),
            "deu": restored.phoneme_inventory("deu"),
        },
        "shared_phones": shared.phonemes.tolist(),
        # rows of `_dense_feature_table` minus the category offsets = composition_feature_matrix(shared phones)
        "training_matrix": (layer._dense_feature_table - offsets).tolist(),
        "training_matrix_direct": restored.composition_feature_matrix(shared.phonemes.tolist()).tolist(),
        "category_offsets": offsets.tolist(),
        "embedding_rows": int(layer._attribute_embeddings.weight.shape[0]),
    }
    assert golden["training_matrix"] == golden["training_matrix_direct"]
    assert golden["shared_phones"] == state.language_allophones.shared_phones
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g9_restricted_indexer.json")
    with open(out, "w", encoding="utf-8") as f:
        json.dump(golden, f, ensure_ascii=False, indent=0)
    print("wrote", out, {k: len(v) for k, v in golden["inventories"].items()}, "shared phones", len(golden["shared_phones"]))
    print("spa table", spa)
    print("spa restored", golden["inventories"]["spa"])
    print("spa+ita", golden["inventories"]["spa+ita"])


if __name__ == "__main__":
    main()
