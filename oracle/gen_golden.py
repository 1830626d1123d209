"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the REAL reference (/root/reference, imported via
oracle/ref_import.py) in the build container, and checks the oracle restatement against it while doing so.

    python oracle/gen_golden.py            # regenerates every fixture (needs /root/reference; ~2 min)

The fixtures are data only (inputs, weights or weight seeds, expected outputs).  The reference itself never travels.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from allophant_amd import spec as S, synthetic  # noqa: E402
from oracle import allophant_oracle as O, ref_import  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def _load_into_reference(model, state):
    ref_state = model.state_dict()
    missing = [k for k in ref_state if k not in state and "._layers." not in k.split("encoder")[-1][:9]]
    # `encoder._layers.*` aliases appear when the graph uses OUTPUT_i; they alias `layers.*`
    alias = {k: state[k.replace("encoder._layers.", "encoder.layers.")] for k in ref_state if "encoder._layers." in k}
    full = dict(state)
    full.update(alias)
    unexpected = [k for k in full if k not in ref_state]
    missing = [k for k in ref_state if k not in full]
    assert not unexpected, f"unexpected keys {unexpected[:5]}"
    assert not missing, f"missing keys {missing[:5]}"
    for k, v in ref_state.items():
        assert tuple(v.shape) == tuple(full[k].shape), (k, v.shape, full[k].shape)
    model.load_state_dict(full)


def run_case(name, spec, n, length, ragged, inventory_phones, seed, store_weights, subsample=None, store_audio=True):
    S.validate(spec)
    torch.manual_seed(0)
    composed = bool(spec.get("embedding_size"))
    train_table = None
    if composed or spec.get("allophone_layer"):
        n_train = spec.get("shared_phones") if spec.get("allophone_layer") else next(
            c["size"] for c in spec["classes"] if c["name"] == "phoneme")
        train_table = synthetic.make_inventory(spec, n_train, seed=seed + 100)
        # make sure every category occurs so that n_f == max + 1 (embedding table size, acoustic_model.py:196-207)
        for f, ncat in enumerate(spec["composition_categories"]):
            train_table[:ncat, f] = torch.arange(ncat)
    t0 = time.time()
    estimator, model = ref_import.build_reference_estimator(spec, train_table)
    state = synthetic.make_state_dict(spec, seed=seed)
    _load_into_reference(model, state)
    audio, lengths = synthetic.make_audio(n, length, seed=1234 + seed, ragged=ragged)
    tfi = synthetic.make_inventory(spec, inventory_phones, seed=seed) if composed else None
    pred = ref_import.reference_predict(estimator, audio, lengths, tfi, True)
    raw = ref_import.reference_predict(estimator, audio, lengths, tfi, False)
    t1 = time.time()

    # ---- pin the oracle against the reference ----
    offsets = synthetic.category_offsets(spec) if composed else None
    if composed:
        ref_offsets = model._projection._layers["phoneme"]._composition_layer._category_offsets.view(-1)
        assert torch.equal(ref_offsets, offsets), (ref_offsets, offsets)
    out, flen, inter = O.predict(audio, lengths, state, spec, tfi, offsets, True, keep_intermediates=True)
    assert list(out.keys()) == list(pred.outputs.keys()) == S.output_names(spec), (list(out), list(pred.outputs))
    assert torch.equal(flen, pred.lengths)
    worst = 0.0
    for k in out:
        valid = (torch.arange(out[k].shape[0]).unsqueeze(1) < flen.unsqueeze(0)).unsqueeze(-1)
        err = ((out[k] - pred.outputs[k]).abs() * valid).max().item()
        worst = max(worst, err)
    # hidden states straight from the HF module for an intermediate pin
    with torch.inference_mode():
        mask = O.mask_sequence(lengths)
        # the call of acoustic_model.py:839-847: attention_mask=None when the preprocessor has return_attention_mask=False
        hf = model._acoustic_model._model(O.zero_mean_unit_var_norm(audio, lengths, mask),
                                          mask.long() if spec.get("use_attention_mask", True) else None,
                                          output_hidden_states=True)
    if spec.get("add_adapter"):
        # the adapter is live inside the reference's HF model (its result, strided down, is `last_hidden_state`) -- and `predict` never
        # sees it: the hidden-state tuple keeps the encoder's frame count, which is also what `pred.lengths` counts
        assert model._acoustic_model._model.adapter is not None
        assert hf.last_hidden_state.shape[1] < hf.hidden_states[-1].shape[1] == int(flen.max()), (hf.last_hidden_state.shape, flen)
        assert all(v.shape[0] == int(flen.max()) for v in pred.outputs.values())
    fm = (torch.arange(flen.max()).unsqueeze(0) < flen.unsqueeze(1)).unsqueeze(-1)
    herr = max(((a - b).abs() * fm).max().item() for a, b in zip(hf.hidden_states, inter["hidden_states"]))
    cerr = ((hf.extract_features - torch.nn.functional.layer_norm(
        inter["conv_out"], (spec["conv_dim"],),
        state[synthetic.AM + "feature_projection.layer_norm.weight"],
        state[synthetic.AM + "feature_projection.layer_norm.bias"], spec["eps"])).abs()).max().item()
    print(f"[{name}] reference {t1 - t0:.1f}s; oracle-vs-reference max-abs: log-probs {worst:.2e}, hidden {herr:.2e}, "
          f"conv {cerr:.2e}; keys={list(out.keys())[:3]}..{len(out)}")
    assert worst < 2e-4 and herr < 2e-4, "oracle restatement deviates from the reference"

    # greedy decode by the reference decoder (predictions.py:194-207)
    from allophant.predictions import GreedyCTCDecoder

    dec = GreedyCTCDecoder()
    tokens = {}
    for k, v in pred.outputs.items():
        hyps = dec(v.transpose(1, 0).contiguous(), pred.lengths)
        for i, h in enumerate(hyps):
            tokens[f"tokens/{k}/{i}"] = _np(h[0].tokens)
            tokens[f"timesteps/{k}/{i}"] = _np(h[0].timesteps)
            tokens[f"score/{k}/{i}"] = np.float32(h[0].score.item())

    data = {
        "spec_json": np.frombuffer(json.dumps(spec).encode(), dtype=np.uint8),
        "seed": np.int64(seed),
        "audio": _np(audio) if store_audio else np.zeros(0, np.float32),
        "audio_args": np.array([n, length, 1234 + seed, int(ragged)], dtype=np.int64),
        "lengths": _np(lengths),
        "frame_lengths": _np(pred.lengths),
        "output_names": np.frombuffer(json.dumps(list(pred.outputs.keys())).encode(), dtype=np.uint8),
    }
    if tfi is not None:
        data["tfi"] = _np(tfi)
        data["category_offsets"] = _np(offsets)
    if store_weights:
        for k, v in state.items():
            data["w/" + k] = _np(v)
    rows = slice(None)
    for k in pred.outputs:
        data["logprobs/" + k] = _np(pred.outputs[k])
        data["logits/" + k] = _np(raw.outputs[k])
    hs = hf.hidden_states
    keep = range(len(hs)) if subsample is None else subsample
    for i in keep:
        h = hs[i]
        data[f"hidden/{i}"] = _np(h if subsample is None else h[:, :, ::8])
    conv = inter["conv_out"]
    data["conv_out"] = _np(conv if subsample is None else conv[:, :, ::8])
    data.update(tokens)
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


def integer_goldens():
    ref_import.install()
    from allophant.attribute_graph import AttributeGraph, AttributeNode
    from allophant.network import frontend
    from allophant.predictions import GreedyCTCDecoder
    from allophant import utils

    data = {}
    # G4a: frame-length arithmetic (frontend.py:192-203 applied per conv layer as acoustic_model.py:832-835)
    enc = S.xlsr_300m_encoder()
    sweep = torch.tensor(
        [400, 401, 479, 480, 719, 720, 721, 799, 800, 1039, 1040, 1279, 1280, 16000, 16001, 31999, 32000, 47999, 48000,
         80000, 159999, 160000, 160001, 479999, 480000, 959999, 960000] + list(range(400, 4000, 37)), dtype=torch.int64)
    out = sweep
    for k, s in zip(enc["conv_kernel"], enc["conv_stride"]):
        out = frontend.conv_length(k, s, use_padding=False)(out)
    data["lengths_in"] = _np(sweep)
    data["lengths_out"] = _np(out)
    assert torch.equal(out, O.downsampled_lengths(sweep, enc["conv_kernel"], enc["conv_stride"]))
    assert S.frame_lengths(sweep.tolist(), enc) == out.tolist()
    # G4b: sample masks (utils.py:45-76)
    lens = torch.tensor([5, 1, 9, 3])
    data["mask_lengths"] = _np(lens)
    data["mask"] = _np(utils.mask_sequence(lens))
    assert torch.equal(utils.mask_sequence(lens), O.mask_sequence(lens))
    # G4c: greedy CTC decode (predictions.py:194-207) on random log-probs (continuous values: no exact ties)
    g = torch.Generator().manual_seed(7)
    dec = GreedyCTCDecoder()
    cases = []
    for ci, (n, t, c) in enumerate([(3, 50, 4), (2, 200, 28), (4, 17, 201), (1, 1, 4), (2, 64, 2)]):
        # sticky random walk so that repeated symbols / blanks occur
        base = torch.randn(n, t, c, generator=g)
        base = base + 2.5 * torch.nn.functional.one_hot(
            torch.randint(0, c, (n, (t + 3) // 4), generator=g).repeat_interleave(4, 1)[:, :t], c)
        lp = torch.log_softmax(base, -1)
        ln = torch.randint(1, t + 1, (n,), generator=g)
        ln[0] = t
        hyps = dec(lp, ln)
        orc = O.greedy_ctc(lp, ln)
        data[f"ctc/{ci}/logprobs"] = _np(lp)
        data[f"ctc/{ci}/lengths"] = _np(ln)
        for i, h in enumerate(hyps):
            data[f"ctc/{ci}/tokens/{i}"] = _np(h[0].tokens)
            data[f"ctc/{ci}/timesteps/{i}"] = _np(h[0].timesteps)
            data[f"ctc/{ci}/score/{i}"] = np.float32(h[0].score.item())
            assert torch.equal(h[0].tokens, orc[i][0]) and torch.equal(h[0].timesteps, orc[i][1])
        cases.append(ci)
    data["ctc_cases"] = np.array(cases)
    # G4d: evaluation order of random DAGs (attribute_graph.py:124-199)
    rng = np.random.default_rng(3)
    graphs = []
    for gi in range(12):
        n = int(rng.integers(2, 9))
        perm = rng.permutation(n)
        classes = []
        for i in range(n):
            deps = ["OUTPUT"] if rng.random() < 0.6 else []
            # dependencies only on nodes with a smaller rank in a hidden topological order `perm`
            cands = [j for j in range(n) if perm[j] < perm[i]]
            for j in cands:
                if rng.random() < 0.4:
                    deps.append(f"c{j}")
            if not deps:
                deps = [f"OUTPUT_{int(rng.integers(0, 3))}"]
            rng.shuffle(deps)
            classes.append({"name": f"c{i}", "size": int(rng.integers(2, 5)), "dependencies": list(map(str, deps))})
        graph = AttributeGraph([AttributeNode(c["name"], c["size"], None, c["dependencies"]) for c in classes])
        order = [n_.name for n_ in graph.sort()]
        assert order == [classes[i]["name"] for i in O.topological_order(classes)], (classes, order)
        assert order == [classes[i]["name"] for i in S.evaluation_order(classes)], (classes, order)
        graphs.append({"classes": classes, "order": order})
    data["graphs_json"] = np.frombuffer(json.dumps(graphs).encode(), dtype=np.uint8)
    path = os.path.join(GOLDEN, "g4_integer.npz")
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(path, **data)
    print(f"[g4] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


def main():
    torch.set_num_threads(8)
    which = set(sys.argv[1:]) or {"g1", "g2", "g2b", "g3", "g4", "g5", "g8", "g11", "g11b", "g12", "g13", "g13b", "g14", "g15"}
    tiny = S.tiny_encoder(2)
    if "g4" in which:
        integer_goldens()
    if "g1" in which:
        # G1: tiny multitask, 3 attribute heads + composed phoneme head, allophone pass-through, ragged lengths
        spec = S.multitask_spec(tiny, ["syllabic", "long", "nasal"], embedding_size=16, train_phonemes=9, n_features=5,
                                n_values=3, allophone_layer=True)
        spec["shared_phones"] = 11
        run_case("g1_tiny_multitask", spec, n=3, length=6000, ragged=True, inventory_phones=7, seed=1, store_weights=True)
    if "g2" in which:
        # G2: tiny hierarchical graph with OUTPUT_i dependencies, blank columns dropped before the softmax
        enc = S.tiny_encoder(3)
        spec = S.multitask_spec(enc, ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5)
        spec["classes"] = [
            {"name": "phoneme", "size": 9, "dependencies": ["OUTPUT", "syllabic", "long", "OUTPUT_1"]},
            {"name": "syllabic", "size": 3, "dependencies": ["OUTPUT_0"]},
            {"name": "long", "size": 2, "dependencies": ["syllabic", "OUTPUT"]},
        ]
        spec["dependency_blanks"] = False
        run_case("g2_tiny_hierarchical", spec, n=2, length=5000, ragged=True, inventory_phones=6, seed=2,
                 store_weights=False)
    if "g2b" in which:
        enc = S.tiny_encoder(2)
        spec = S.hierarchical_spec(enc, ["syllabic", "long", "nasal"], embedding_size=16, train_phonemes=9, n_features=5,
                                   dependency_blanks=True)
        run_case("g2b_tiny_hierarchical_blanks", spec, n=2, length=4000, ragged=True, inventory_phones=12, seed=3,
                 store_weights=False)
    if "g5" in which:
        # baseline schema: single non-composed phoneme head (BASELINE config 1 plumbing), tiny shape
        spec = S.baseline_spec(S.tiny_encoder(2), phonemes=10)
        run_case("g5_tiny_baseline", spec, n=1, length=4800, ragged=False, inventory_phones=0, seed=5, store_weights=False)
    if "g8" in which:
        # G8: time-layer classifiers (ProjectingMultiheadAttention, acoustic_model.py:237-268): `long` = 3 values + blank
        # -> embed 4, 2 heads, sinusoidal positions, on cat(softmax(syllabic), OUTPUT); `nasal` = embed 3, 1 head, no
        # positions, straight on OUTPUT; the composed phoneme head depends on both (hierarchical), ragged batch
        enc = S.tiny_encoder(2)
        spec = S.multitask_spec(enc, ["syllabic", "long", "nasal"], embedding_size=16, train_phonemes=9, n_features=5)
        spec["classes"] = [
            {"name": "syllabic", "size": 3, "dependencies": ["OUTPUT"]},
            {"name": "long", "size": 3, "dependencies": ["syllabic", "OUTPUT"],
             "time_layer": {"num_heads": 2, "positional_embeddings": True}},
            {"name": "nasal", "size": 2, "dependencies": ["OUTPUT"],
             "time_layer": {"num_heads": 1, "positional_embeddings": False}},
            {"name": "phoneme", "size": 9, "dependencies": ["OUTPUT", "long", "nasal"]},
        ]
        run_case("g8_tiny_time_layer", spec, n=3, length=5200, ragged=True, inventory_phones=8, seed=8, store_weights=False)
    if "g11" in which:
        # G11: the group-norm / post-LN wav2vec 2.0 variant (wav2vec2-base / -large: feat_extract_norm="group", conv_bias=False,
        # do_stable_layer_norm=False) with a preprocessor that has return_attention_mask=False, i.e. the reference calls the
        # model with attention_mask=None (acoustic_model.py:814,842-846); tiny shape, ragged batch, hierarchical graph with an
        # OUTPUT_i dependency so that intermediate hidden states of the post-LN stack are pinned too
        enc = S.tiny_encoder(3)
        enc.update(feat_extract_norm="group", conv_bias=False, stable_layer_norm=False, use_attention_mask=False)
        spec = S.multitask_spec(enc, ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5)
        spec["classes"] = [
            {"name": "syllabic", "size": 3, "dependencies": ["OUTPUT_1"]},
            {"name": "long", "size": 2, "dependencies": ["OUTPUT"]},
            {"name": "phoneme", "size": 9, "dependencies": ["OUTPUT", "syllabic", "long"]},
        ]
        run_case("g11_tiny_groupnorm_postln", spec, n=3, length=5600, ragged=True, inventory_phones=7, seed=11,
                 store_weights=False)
    if "g11b" in which:
        # G11b: the same variant WITH the attention mask (a group-norm checkpoint whose preprocessor returns one), plus a conv
        # bias: frames beyond an utterance are zeroed / masked as keys, the GroupNorm still runs over the padded length
        enc = S.tiny_encoder(2)
        enc.update(feat_extract_norm="group", conv_bias=True, stable_layer_norm=False, use_attention_mask=True)
        spec = S.multitask_spec(enc, ["syllabic", "long", "nasal"], embedding_size=16, train_phonemes=9, n_features=5,
                                allophone_layer=True)
        spec["shared_phones"] = 11
        run_case("g11b_tiny_groupnorm_masked", spec, n=3, length=6000, ragged=True, inventory_phones=7, seed=12,
                 store_weights=False)
    if "g12" in which:
        # G12: full wav2vec2-base shape (768 / 12 layers / 12 heads / 3072) of the group-norm / post-LN variant, procedural
        # weights, 2 x 3 s ragged, attention_mask=None; sub-sampled tensors only
        spec = S.multitask_spec(S.wav2vec2_base_encoder(), allophone_layer=True)
        spec["shared_phones"] = 80
        run_case("g12_w2v2base_multitask", spec, n=2, length=48000, ragged=True, inventory_phones=27, seed=0,
                 store_weights=False, subsample=[0, 1, 6, 12], store_audio=False)
    if "g13" in which:
        # G13 (round 6): head_dim != 64.  The reference builds whatever `model_id` names (acoustic_model.py:796-826): XLS-R 1B / 2B
        # have head dimensions 80 / 120.  Tiny shape with hidden 160 / 2 heads = head_dim 80 (rows of Q / K / V padded to 128 on the
        # device), hierarchical graph with an OUTPUT_i dependency, ragged batch
        enc = S.tiny_encoder(2)
        enc.update(hidden=160, heads=2, ffn=320, pos_groups=4)
        spec = S.multitask_spec(enc, ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5, allophone_layer=True)
        spec["shared_phones"] = 11
        spec["classes"] = [
            {"name": "syllabic", "size": 3, "dependencies": ["OUTPUT_1"]},
            {"name": "long", "size": 2, "dependencies": ["OUTPUT"]},
            {"name": "phoneme", "size": 9, "dependencies": ["OUTPUT", "syllabic", "long"]},
        ]
        run_case("g13_tiny_head_dim_80", spec, n=3, length=6400, ragged=True, inventory_phones=7, seed=13, store_weights=False)
    if "g13b" in which:
        # G13b: head_dim 32 (hidden 64 / 2 heads: narrower than the 64-column rows of the device layout), post-LN variant
        enc = S.tiny_encoder(2)
        enc.update(hidden=64, heads=2, ffn=128, pos_groups=4, stable_layer_norm=False)
        spec = S.multitask_spec(enc, ["syllabic", "long", "nasal"], embedding_size=16, train_phonemes=9, n_features=5)
        run_case("g13b_tiny_head_dim_32", spec, n=2, length=5200, ragged=True, inventory_phones=6, seed=14, store_weights=False)
    if "g14" in which:
        # G14 (round 6): the WIDTH of the XLS-R 1B encoder -- hidden 1280, 16 heads of 80, positional convolution with 80 channels per
        # group -- on two layers and a narrow ffn (rows wider than 1024: the row kernels' second instance, the fold's 20 blocks per
        # row, the grouped positional convolution beyond 64 channels per group); 2 x 2 s ragged, sub-sampled tensors only
        enc = S.tiny_encoder(2)
        enc.update(hidden=1280, heads=16, ffn=2560, pos_groups=16, pos_kernel=128, conv_dim=64)
        spec = S.multitask_spec(enc, ["syllabic", "long", "nasal"], embedding_size=32, train_phonemes=12, n_features=6, allophone_layer=True)
        spec["shared_phones"] = 14
        run_case("g14_xlsr1b_width", spec, n=2, length=32000, ragged=True, inventory_phones=9, seed=14, store_weights=False,
                 subsample=[0, 1, 2], store_audio=False)
    if "g15" in which:
        # G15 (round 6): `add_adapter=True` (Wav2Vec2Adapter: three strided Conv1d + GLU layers behind the encoder; the reference builds
        # whatever the HF config names, acoustic_model.py:796-799).  The reference computes the adapter and reads nothing of it
        # (`.hidden_states` = encoder outputs, acoustic_model.py:839-853): same outputs, same frame counts as without -- the golden
        # pins that the drop-in may ignore the adapter's weights
        enc = S.tiny_encoder(2)
        enc.update(add_adapter=True, num_adapter_layers=3, adapter_kernel_size=3, adapter_stride=2)
        spec = S.multitask_spec(enc, ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5, allophone_layer=True)
        spec["shared_phones"] = 11
        run_case("g15_tiny_adapter", spec, n=3, length=6400, ragged=True, inventory_phones=7, seed=15, store_weights=False)
    if "g3" in which:
        # G3: full XLS-R shape, procedural weights (seed 0), 2 x 3 s ragged; sub-sampled tensors only
        spec = S.multitask_spec(S.xlsr_300m_encoder(), allophone_layer=True)
        spec["shared_phones"] = 80
        run_case("g3_xlsr_multitask", spec, n=2, length=48000, ragged=True, inventory_phones=27, seed=0,
                 store_weights=False, subsample=[0, 1, 12, 24], store_audio=False)


if __name__ == "__main__":
    main()
