"""TextPredNet — prediction network (training branch) on the MI355X HIP path.

Mirrors networks/decoder.py:57-80 (ctor) and :82-126 (forward, packed branch :102-120,124) of the reference:
Embedding(V, H, padding_idx=blank) -> LSTM -> Linear(H -> O); returns (outputs, hidden_states).
The step branch (`input_lengths=None`, carried `prev_hidden_state`, decoder.py:121-123) that the reference's search loops call
is one launch of `rnnt_hip_prednet_step` per token column (inference only: no autograd through it).
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ..ops import EmbeddingFn, LinearFn, prednet_step
from .encoder import HipLinear, lengths_to_device
from .rnn import RNN_CELLS


class HipEmbedding(nn.Embedding):
    """nn.Embedding parameters (row padding_idx zero-initialised, no gradient), gather/scatter in HIP."""

    def forward(self, idx: torch.Tensor) -> torch.Tensor:
        return EmbeddingFn.apply(self.weight, idx, self.padding_idx)


class TextPredNet(nn.Module):
    supported_rnns = RNN_CELLS

    def __init__(self, embedding_size: int, pad_token_id: int, hidden_size: int, output_size: int, num_layers: int,
                 rnn_type: str = "lstm", dropout: float = 0.2):
        super().__init__()
        if rnn_type.lower() not in self.supported_rnns:
            raise NotImplementedError(f"rnn_type={rnn_type!r}: supported {sorted(self.supported_rnns)}")
        self.hidden_size = hidden_size
        self.embedding = HipEmbedding(embedding_size, hidden_size, padding_idx=pad_token_id)
        self.rnn = self.supported_rnns[rnn_type.lower()](hidden_size, hidden_size, num_layers,
                                                         dropout=(dropout if num_layers > 1 else 0.0), bidirectional=False)
        self.out_proj = HipLinear(hidden_size, output_size)

    def forward_time_major(self, inputs: torch.Tensor, lens_dev: torch.Tensor) -> torch.Tensor:
        """(B,U1) int64 tokens + int32 device lengths -> (U1,B,O) time-major."""
        emb = self.embedding(inputs.transpose(0, 1).contiguous())  # (U1,B,H)
        return self.out_proj(self.rnn(emb, lens_dev))

    def forward(self, inputs: torch.Tensor, input_lengths=None, prev_hidden_state=None):
        """Training branch (input_lengths given): (B,U1) tokens -> ((B,U1,O), hidden_states) with hidden_states what the
        reference's packed `self.rnn(...)` returns (decoder.py:115,126): the (L,B,H) states after each sequence's own last
        token — (h_n, c_n) for LSTM, h_n otherwise — in the reference's LENGTH-SORTED batch order (it packs the batch sorted by
        descending length at decoder.py:105-111 and only un-sorts `outputs`, :116-120).
        Step branch (input_lengths None, decoder.py:121-123): (B,S) tokens fed one column at a time from `prev_hidden_state`
        (None = zeros; LSTM: a (h, c) tuple of (L,B,H) tensors, GRU / RNN: h) -> ((B,S,O), new hidden state in torch's format)."""
        if input_lengths is not None:
            if prev_hidden_state is not None:
                raise ValueError("prev_hidden_state goes with the step branch (input_lengths=None), as in decoder.py:102-123")
            lens = lengths_to_device(input_lengths, inputs.device)
            emb = self.embedding(inputs.transpose(0, 1).contiguous())
            y, h_n, c_n = self.rnn(emb, lens, want_final=True)
            lens_cpu = input_lengths.detach().cpu() if isinstance(input_lengths, torch.Tensor) else torch.tensor(list(input_lengths))
            order = torch.sort(lens_cpu, descending=True)[1].to(inputs.device)  # decoder.py:106
            h_n = h_n.index_select(1, order)
            hidden = (h_n, c_n.index_select(1, order)) if self.rnn.CELL == 0 else h_n
            return self.out_proj(y).transpose(0, 1).contiguous(), hidden
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("the step branch is inference-only: call it under torch.no_grad()")
        lstm = self.rnn.CELL == 0
        h, c = (prev_hidden_state if lstm else (prev_hidden_state, None)) if prev_hidden_state is not None else (None, None)
        outs = []
        for s_ in range(inputs.size(1)):
            h, c = prednet_step(inputs[:, s_], self.embedding.weight, self.rnn.flat_weights(), self.rnn.CELL, h, c)
            outs.append(LinearFn.apply(h[-1], self.out_proj.weight, self.out_proj.bias))
        return torch.stack(outs, dim=1), ((h, c) if lstm else h)
