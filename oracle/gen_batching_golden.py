"""TEST INFRASTRUCTURE ONLY.  tests/golden/g7_batching.json from the REAL reference: ``MaxFrameBatchSampler``
(allophant/batching.py:94-139), ``_build_batch`` (162-177) and ``RawLabeledBatch.split_by_language``
(allophant/dataset_processing.py:103-126) on seeded inputs."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

ref_import.install()

import torch  # noqa: E402
from allophant import batching as ref_batching  # noqa: E402
from allophant.dataset_processing import Batch as RefBatch, RawLabeledBatch  # noqa: E402


def main():
    g = torch.Generator().manual_seed(11)
    cases = []
    for n, budget in [(40, 200000), (25, 90000), (10, 50000), (7, 10)]:
        lengths = torch.randint(4000, 60000, (n,), generator=g)
        order = torch.randperm(n, generator=g).tolist()
        sampler = ref_batching.MaxFrameBatchSampler(order, budget, lengths)
        cases.append({"lengths": lengths.tolist(), "order": order, "max_frames": budget, "batches": [list(map(int, b)) for b in sampler]})
    # _build_batch (unlabeled) + split_by_language
    lens = [5, 9, 3, 7, 7, 2]
    langs = [0, 0, 2, 2, 2, 1]
    entries = [RefBatch(torch.arange(1, l + 1, dtype=torch.float32).unsqueeze(0) * (i + 1), torch.tensor(l), torch.tensor(lang))
               for i, (l, lang) in enumerate(zip(lens, langs))]
    # the reference collates single-entry batches whose audio is [1, L]; squeeze like its datasets do ([L] per entry)
    for e in entries:
        e.audio_features = e.audio_features.squeeze(0)
    build = ref_batching._build_batch(ref_batching.BatchType.UNLABELED)
    dense = build(entries)
    raw = RawLabeledBatch(dense.audio_features, dense.lengths, dense.language_ids, [[["x"]] * len(lens)], [str(i) for i in range(len(lens))])
    splits = [{"language": int(lang), "audio": b.audio_features.tolist(), "lengths": b.lengths.tolist(), "ids": b.language_ids.tolist()}
              for lang, b in raw.split_by_language()]
    out = {"sampler_cases": cases,
           "collate": {"lens": lens, "langs": langs, "audio": dense.audio_features.tolist(), "lengths": dense.lengths.tolist(),
                       "ids": dense.language_ids.tolist()},
           "splits": splits}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g7_batching.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, [len(c["batches"]) for c in cases], len(splits))


if __name__ == "__main__":
    main()
