"""HipLSTM / HipGRU / HipRNN: parameter containers + forward for the persistent HIP recurrences (no torch.nn.LSTM/GRU/RNN,
no MIOpen call) — the reference's `supported_rnns = {lstm, gru, rnn}` (networks/encoder.py:48-52, decoder.py:51-55).

Parameter names, shapes, gate order (i,f,g,o) and initialisation order are those of torch.nn.LSTM, which the
reference instantiates at networks/encoder.py:67-75 and networks/decoder.py:71-79, so reference checkpoints
(`...rnn.weight_ih_l0`, `...rnn.weight_hh_l0_reverse`, ...) load unchanged.
"""
import math

import torch
import torch.nn as nn

from .. import _lib
from ..ops import LstmStackFn


class HipLSTM(nn.Module):
    GATES, CELL = 4, 0  # i,f,g,o

    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1, dropout: float = 0.0,
                 bidirectional: bool = False):
        super().__init__()
        if hidden_size % 4 != 0:
            raise ValueError(f"{type(self).__name__} needs hidden_size % 4 == 0 (each workgroup owns 4-unit slices)")
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.dropout, self.bidirectional = float(dropout), bool(bidirectional)
        D = 2 if bidirectional else 1
        for layer in range(num_layers):
            in_l = input_size if layer == 0 else hidden_size * D
            for suffix in ("", "_reverse")[:D]:
                G = self.GATES
                self.register_parameter(f"weight_ih_l{layer}{suffix}", nn.Parameter(torch.empty(G * hidden_size, in_l)))
                self.register_parameter(f"weight_hh_l{layer}{suffix}", nn.Parameter(torch.empty(G * hidden_size, hidden_size)))
                self.register_parameter(f"bias_ih_l{layer}{suffix}", nn.Parameter(torch.empty(G * hidden_size)))
                self.register_parameter(f"bias_hh_l{layer}{suffix}", nn.Parameter(torch.empty(G * hidden_size)))
        self.reset_parameters()
        self._step = 0

    def reset_parameters(self) -> None:
        stdv = 1.0 / math.sqrt(self.hidden_size)
        for p in self.parameters():  # registration order == torch.nn.LSTM's, so the same seed gives the same weights
            nn.init.uniform_(p, -stdv, stdv)

    def flat_weights(self):
        out = []
        for layer in range(self.num_layers):
            for suffix in ("", "_reverse")[:2 if self.bidirectional else 1]:
                for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    out.append(getattr(self, f"{name}_l{layer}{suffix}"))
        return out

    def forward(self, x_tm: torch.Tensor, lens, want_final: bool = False):
        """x_tm (T,B,I) time-major fp32, lens (B) int32 on the same device — or an ops.RaggedPlan (lengths + valid-frame table:
        padded frames are then skipped by the big products and the recurrences) -> (T,B,D*H), zeros for t >= lens[b].
        want_final: -> (y, h_n, c_n) with the (L*D, B, H) states after each sequence's own last step (c_n empty unless LSTM)."""
        from ..ops import RaggedPlan
        p = self.dropout if (self.training and self.num_layers > 1) else 0.0
        self._step += 1
        seed = (torch.initial_seed() * 1000003 + self._step * 7919) & 0x7FFFFFFFFFFFFFFF
        B = x_tm.shape[1]
        cap = _lib.lib().rnnt_hip_lstm_max_batch(self.hidden_size, 2 if self.bidirectional else 1, self.CELL)
        if cap <= 0:
            raise _lib.RnntHipError(f"hidden_size={self.hidden_size} does not fit the persistent recurrence kernels")
        if B <= cap:
            return LstmStackFn.apply(x_tm, lens, self.hidden_size, self.num_layers, self.bidirectional, p, seed, self.CELL,
                                     want_final, *self.flat_weights())
        if isinstance(lens, RaggedPlan):   # the table indexes the whole batch: slices of it run dense
            lens = lens.lens
        # batch rows are independent: run slices of the batch back to back (autograd sums the weight gradients)
        outs = []
        for b0 in range(0, B, cap):
            outs.append(LstmStackFn.apply(x_tm[:, b0:b0 + cap].contiguous(), lens[b0:b0 + cap].contiguous(), self.hidden_size,
                                          self.num_layers, self.bidirectional, p, seed + 104729 * (b0 + 1), self.CELL,
                                          want_final, *self.flat_weights()))
        if not want_final:
            return torch.cat(outs, dim=1)
        return tuple(torch.cat([o[i] for o in outs], dim=1) for i in range(3))


class HipGRU(HipLSTM):
    """torch.nn.GRU parameters / equations (gates r,z,n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn)))."""
    GATES, CELL = 3, 1


class HipRNN(HipLSTM):
    """torch.nn.RNN (Elman) parameters; nonlinearity tanh (the reference's default) or relu."""
    GATES, CELL = 1, 2

    def __init__(self, *args, nonlinearity: str = "tanh", **kwargs):
        if nonlinearity not in ("tanh", "relu"):
            raise ValueError("nonlinearity must be tanh or relu")
        self.CELL = 2 if nonlinearity == "tanh" else 3
        super().__init__(*args, **kwargs)


RNN_CELLS = {"lstm": HipLSTM, "gru": HipGRU, "rnn": HipRNN}
