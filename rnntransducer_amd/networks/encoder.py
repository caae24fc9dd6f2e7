"""AudioTransNet — transcription network of the RNN-Transducer on the MI355X HIP path.

Mirrors the constructor and forward signature of the reference's networks/encoder.py:54-76,78-108 (same
argument names, same parameter names `rnn.*`, `out_proj.*`).  What differs is HOW: no sort / pack /
unpack / unsort and no cuDNN/MIOpen; sequences are masked per row inside the persistent HIP LSTM kernel,
which gives the same result (zero outputs on padded frames, reverse direction starting at each sequence's
last frame).  rnn_type is one of the reference's supported_rnns (lstm | gru | rnn, encoder.py:48-52).
"""
from typing import Sequence, Union

import torch
import torch.nn as nn

from ..ops import LinearFn
from .rnn import RNN_CELLS


def lengths_to_device(lengths: Union[Sequence[int], torch.Tensor], device) -> torch.Tensor:
    """The reference hands lengths as python lists (dataloader.py:20,37); the kernels want int32 on device."""
    if isinstance(lengths, torch.Tensor):
        return lengths.to(device=device, dtype=torch.int32, non_blocking=True)
    return torch.tensor(list(lengths), dtype=torch.int32, device=device)


class HipLinear(nn.Linear):
    """nn.Linear parameters (same names / init), forward on the MFMA GEMM of librnnt_hip."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return LinearFn.apply(x, self.weight, self.bias)


class AudioTransNet(nn.Module):
    supported_rnns = RNN_CELLS

    def __init__(self, input_size: int, hidden_size: int, output_size: int, num_layers: int, rnn_type: str = "lstm",
                 dropout: float = 0.2, bidirectional: bool = True):
        super().__init__()
        if rnn_type.lower() not in self.supported_rnns:
            raise NotImplementedError(f"rnn_type={rnn_type!r}: supported {sorted(self.supported_rnns)}")
        self.hidden_size = hidden_size
        self.rnn = self.supported_rnns[rnn_type.lower()](input_size, hidden_size, num_layers,
                                                         dropout=(dropout if num_layers > 1 else 0.0),
                                                         bidirectional=bidirectional)
        self.out_proj = HipLinear(2 * hidden_size if bidirectional else hidden_size, output_size)

    def forward_time_major(self, inputs: torch.Tensor, lens_dev) -> torch.Tensor:
        """(B,T,F) mel + int32 device lengths (or an ops.RaggedPlan built from the host list of lengths) -> (T,B,O) time-major
        (what the fused joint+loss consumes)."""
        x_tm = inputs.transpose(0, 1).contiguous()
        return self.out_proj(self.rnn(x_tm, lens_dev))

    def forward(self, inputs: torch.Tensor, inputs_lengths) -> torch.Tensor:
        """Reference surface (encoder.py:78): (B,T,F), lengths -> (B,T,O)."""
        lens = lengths_to_device(inputs_lengths, inputs.device)
        return self.forward_time_major(inputs, lens).transpose(0, 1).contiguous()
