"""Host-side helpers that keep the reference's names and argument meaning for the integer / boolean glue around the
prediction path.  On the device these are folded into the kernels; the helpers exist for callers
that used the upstream functions directly."""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
from torch import Tensor


def mask_sequence(lengths: Tensor, max_length: Optional[int] = None, start: int = 0, inverse: bool = False,
                  batch_first: bool = True) -> Tensor:
    """Boolean ``[batch, max_length - start]`` mask of the valid positions of variable-length sequences (reference
    ``allophant/utils.py:45-76``): ``arange(start, max_length) < lengths[:, None]``, ``>=`` with ``inverse``, transposed
    when ``batch_first`` is false.  ``max_length`` defaults to ``lengths.max()`` and may truncate or pad."""
    if max_length is None:
        max_length = int(lengths.max())
    positions = torch.arange(start, max_length, device=lengths.device)
    if batch_first:
        positions, limits = positions.unsqueeze(0), lengths.unsqueeze(1)
    else:
        positions, limits = positions.unsqueeze(1), lengths.unsqueeze(0)
    return positions >= limits if inverse else positions < limits


def conv_length(kernel_size: int, stride: int = 1) -> Callable[[Tensor], Tensor]:
    """Output length of an un-padded strided convolution, ``floor((length - kernel) / stride) + 1`` (reference
    ``frontend.conv_length(kernel, stride, use_padding=False)``, ``frontend.py:192-203``, as wav2vec 2.0 uses it)."""

    def length(lengths: Tensor) -> Tensor:
        return torch.div(lengths - kernel_size, stride, rounding_mode="floor") + 1

    return length


def downsampled_lengths(lengths: Tensor, kernels: Sequence[int], strides: Sequence[int]) -> Tensor:
    """Frame counts after the conv feature extractor (reference ``Wav2Vec2AcousticModel.downsampled_lengths``,
    ``acoustic_model.py:832-835``): the seven ``conv_length`` steps of XLS-R map 160 000 samples to 499 frames."""
    for kernel, stride in zip(kernels, strides):
        lengths = conv_length(kernel, stride)(lengths)
    return lengths
