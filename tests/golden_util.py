"""Helpers shared by the parity tests: loading the committed golden fixtures (tests/golden/*.npz, generated from the
real reference by oracle/gen_golden.py) and rebuilding their inputs."""
import json
import os

import numpy as np
import torch

from allophant_amd import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.spec = json.loads(bytes(self.z["spec_json"]).decode())
        self.seed = int(self.z["seed"])
        self.output_names = json.loads(bytes(self.z["output_names"]).decode())
        n, length, aseed, ragged = [int(x) for x in self.z["audio_args"]]
        audio, lengths = synthetic.make_audio(n, length, seed=aseed, ragged=bool(ragged))
        if self.z["audio"].size:
            stored = torch.from_numpy(self.z["audio"])
            assert torch.equal(stored, audio), "procedural audio changed (torch RNG drift?)"
        assert torch.equal(lengths, torch.from_numpy(self.z["lengths"]))
        self.audio, self.lengths = audio, lengths
        self.frame_lengths = torch.from_numpy(self.z["frame_lengths"])
        self.tfi = torch.from_numpy(self.z["tfi"]) if "tfi" in self.z.files else None
        self.category_offsets = torch.from_numpy(self.z["category_offsets"]) if "category_offsets" in self.z.files else None

    def state_dict(self):
        stored = {k[2:]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith("w/")}
        generated = synthetic.make_state_dict(self.spec, seed=self.seed)
        if stored:
            assert stored.keys() == generated.keys()
            for k in stored:
                assert torch.equal(stored[k], generated[k]), f"procedural weights changed for {k}"
            return stored
        return generated

    def logprobs(self, name):
        return torch.from_numpy(self.z["logprobs/" + name])

    def logits(self, name):
        return torch.from_numpy(self.z["logits/" + name])

    def hidden(self, i):
        return torch.from_numpy(self.z[f"hidden/{i}"])

    def hidden_indices(self):
        return sorted(int(k.split("/")[1]) for k in self.z.files if k.startswith("hidden/"))

    def conv_out(self):
        return torch.from_numpy(self.z["conv_out"])

    def tokens(self, name, i):
        return (torch.from_numpy(self.z[f"tokens/{name}/{i}"]), torch.from_numpy(self.z[f"timesteps/{name}/{i}"]),
                float(self.z[f"score/{name}/{i}"]))

    @property
    def subsampled(self):
        return self.hidden(self.hidden_indices()[0]).shape[-1] != self.spec["hidden"]


def valid_mask(frame_lengths, T):
    """[T, N, 1] mask of frames t < len[n] (padded-frame outputs are garbage-but-deterministic upstream)."""
    return (torch.arange(T).unsqueeze(1) < frame_lengths.unsqueeze(0)).unsqueeze(-1)


def max_abs_valid_tm(a, b, frame_lengths):
    """max |a - b| over valid frames for time-major [T, N, C] tensors."""
    m = valid_mask(frame_lengths, a.shape[0])
    return ((a - b).abs() * m).max().item()


def max_abs_valid_bm(a, b, frame_lengths):
    """max |a - b| over valid frames for batch-major [N, T, C] tensors."""
    m = (torch.arange(a.shape[1]).unsqueeze(0) < frame_lengths.unsqueeze(1)).unsqueeze(-1)
    return ((a - b).abs() * m).max().item()
