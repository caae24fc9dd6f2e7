"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's RNN-T training hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import this module; the product (rnntransducer_amd/) never does and
fails loudly when its HIP library is missing.

What is restated, and from where (paths relative to /root/reference):
  * transcription net   networks/encoder.py:54-76 (ctor), :93-103 (forward): lengths -> sort ->
                        pack_padded_sequence -> nn.LSTM(batch_first, bidirectional) -> pad_packed (zeros on
                        padding) -> unsort -> Linear(D*H -> O).
  * prediction net      networks/decoder.py:57-80 (ctor), :102-120,124 (training branch of forward):
                        Embedding(V, H, padding_idx=blank) -> packed uni-LSTM -> Linear(H -> O).
  * joint               networks/transducer.py:54-69: broadcast enc over U+1 and dec over T, concat on the
                        feature axis (enc first), GELU(approximate="tanh"), Linear(O_e + O_d -> V).
  * loss                model.py:39,57: RNNTLoss(blank, reduction="mean") on (logits, targets int32,
                        frame lengths int32, target lengths int32).  Arithmetic: oracle/rnnt_loss_ref.c.
  * batch layout        dataloader.py:19-49 (7-tuple, dtypes, padding value 0).
torch.nn.LSTM / Linear / Embedding / GELU on the CPU are *dependencies* of the reference, not reference
code, and are used here as such.  The restatement is structurally different from the
reference (functional, `enforce_sorted=False` instead of a manual sort/unsort) and is checked against
the imported reference networks by tests/golden/make_golden.py -> tests/golden/*.npz.

PARITY PIN: networks (a4-a7) pinned by fixtures generated from the imported reference; loss (a8) pinned by
the G3 known-answer vector, brute force and autograd (see rnnt_loss_ref.c header) — unpinned by the
reference's own loss package, which is not installable offline.
"""
from __future__ import annotations

import ctypes
import itertools
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import build_oracle

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_oracle.build())
        for name, real in (("rnnt_loss_ref_f64", ctypes.c_double), ("rnnt_loss_ref_f32", ctypes.c_float)):
            fn = getattr(_LIB, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 2
    return _LIB


# ----------------------------------------------------------------------------------------------------
# loss (model.py:39,57; arithmetic per SURVEY Appendix A.3)
# ----------------------------------------------------------------------------------------------------
def rnnt_loss_c(logits: np.ndarray, labels: np.ndarray, t_lens, u_lens, blank: int = 0,
                want_grad: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """Per-utterance NLL (B,) and d NLL_b / d logits (B,T,U1,V) from the C restatement.

    dtype of `logits` selects the float64 or the float32 (warp-transducer-like) variant.
    """
    assert logits.ndim == 4
    B, T, U1, V = logits.shape
    real = logits.dtype
    assert real in (np.float64, np.float32)
    logits = np.ascontiguousarray(logits)
    labels = np.ascontiguousarray(labels, dtype=np.int32).reshape(B, max(U1 - 1, 0))
    t_lens = np.ascontiguousarray(t_lens, dtype=np.int32)
    u_lens = np.ascontiguousarray(u_lens, dtype=np.int32)
    nll = np.empty(B, dtype=real)
    grad = np.empty_like(logits) if want_grad else None
    fn = _lib().rnnt_loss_ref_f64 if real == np.float64 else _lib().rnnt_loss_ref_f32
    rc = fn(logits.ctypes.data, labels.ctypes.data, t_lens.ctypes.data, u_lens.ctypes.data, B, T, U1, V,
            int(blank), nll.ctypes.data, grad.ctypes.data if want_grad else None)
    if rc != 0:
        raise ValueError(f"rnnt_loss_ref returned {rc} (bad lengths or out of memory)")
    return nll, grad


def rnnt_nll_torch(logits: torch.Tensor, labels, t_lens, u_lens, blank: int = 0) -> torch.Tensor:
    """Independent, differentiable log-space DP in torch (python loops: SMALL lattices only).

    Used to pin the C restatement through autograd and to produce golden parameter
    gradients of (loss o reference networks).  Returns per-utterance NLL (B,).
    """
    B, T, U1, V = logits.shape
    lp = torch.log_softmax(logits, dim=-1)
    out = []
    for b in range(B):
        Tb, Ub1 = int(t_lens[b]), int(u_lens[b]) + 1
        alpha = [[None] * Ub1 for _ in range(Tb)]
        for t in range(Tb):
            for u in range(Ub1):
                if t == 0 and u == 0:
                    alpha[t][u] = lp.new_zeros(())
                    continue
                terms = []
                if t > 0:
                    terms.append(alpha[t - 1][u] + lp[b, t - 1, u, blank])
                if u > 0:
                    terms.append(alpha[t][u - 1] + lp[b, t, u - 1, int(labels[b][u - 1])])
                alpha[t][u] = terms[0] if len(terms) == 1 else torch.logaddexp(terms[0], terms[1])
        out.append(-(alpha[Tb - 1][Ub1 - 1] + lp[b, Tb - 1, Ub1 - 1, blank]))
    return torch.stack(out)


def rnnt_nll_bruteforce(logits: np.ndarray, labels: Sequence[int], blank: int = 0) -> float:
    """-log sum over ALL monotone alignments of one utterance (float64; tiny lattices only).

    logits (T,U1,V).  A path interleaves U emissions (right moves) with T blanks (down moves), the last
    move being the final blank at (T-1,U).
    """
    T, U1, V = logits.shape
    U = U1 - 1
    z = logits.astype(np.float64)
    lp = z - np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1, keepdims=True)) - z.max(-1, keepdims=True)
    total = -math.inf
    # choose at which of the T+U-1 interior moves the U emissions happen
    for emits in itertools.combinations(range(T + U - 1), U):
        emits = set(emits)
        t = u = 0
        s = 0.0
        for step in range(T + U - 1):
            if step in emits:
                s += lp[t, u, labels[u]]
                u += 1
            else:
                s += lp[t, u, blank]
                t += 1
        if t != T - 1 or u != U:
            continue
        s += lp[T - 1, U, blank]
        total = np.logaddexp(total, s)
    return float(-total)


class _RNNTLossFn(torch.autograd.Function):
    """torch wrapper over the C restatement so the CPU composite can run a full training step."""

    @staticmethod
    def forward(ctx, logits, targets, t_lens, u_lens, blank):
        nll, grad = rnnt_loss_c(logits.detach().numpy(), targets.numpy(), t_lens.numpy(), u_lens.numpy(), blank)
        ctx.save_for_backward(torch.from_numpy(grad))
        return torch.from_numpy(nll)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g.view(-1, 1, 1, 1), None, None, None, None


def rnnt_loss_mean(logits, targets, t_lens, u_lens, blank: int = 0) -> torch.Tensor:
    """reduction="mean" of model.py:31,39: mean over the batch of per-utterance NLL (0-d tensor)."""
    return _RNNTLossFn.apply(logits, targets, t_lens, u_lens, blank).mean()


# ----------------------------------------------------------------------------------------------------
# networks (networks/encoder.py, networks/decoder.py, networks/transducer.py) — torch-CPU composite
# ----------------------------------------------------------------------------------------------------
PER_UTTERANCE = False  # see _per_utterance_lstm; switched on by tests at config 3 / 5 sizes only


def _per_utterance_lstm(rnn: nn.Module, x: torch.Tensor, lens: Sequence[int]) -> torch.Tensor:
    """The same function as `_packed_lstm`, computed one utterance at a time on its own valid prefix (no PackedSequence): a packed
    batch treats every sequence independently, ends it at its own length (the reverse direction starts at its own last frame) and
    pads the output with zeros — exactly what running `rnn` on x[b, :len_b] alone and zero-padding gives.  Exists because autograd
    through a PackedSequence on the CPU costs O(T^2) (every timestep's slice gradient is a full-size zero tensor): 114 s instead of
    a few seconds for config 3's T = 2000.  Equality with `_packed_lstm` (1e-12, outputs and every gradient) is pinned by
    tests/test_oracle_networks.py::test_per_utterance_lstm_equals_the_packed_one."""
    lens = [int(n) for n in lens]
    if len(set(lens)) == 1:   # equal lengths: the whole batch is one plain call on the common prefix
        y, _ = rnn(x[:, :lens[0]])
        return F.pad(y, (0, 0, 0, x.size(1) - lens[0]))
    outs = []
    for b, n in enumerate(lens):
        y, _ = rnn(x[b:b + 1, :n])
        outs.append(F.pad(y, (0, 0, 0, x.size(1) - n)))
    return torch.cat(outs, dim=0)


def _packed_lstm(rnn: nn.Module, x: torch.Tensor, lens: Sequence[int]) -> torch.Tensor:
    """encoder.py:93-102 / decoder.py:105-120: packed-sequence LSTM, zero outputs on padded frames."""
    if PER_UTTERANCE and not (rnn.training and rnn.dropout > 0):
        return _per_utterance_lstm(rnn, x, lens)
    total = x.size(1)
    packed = nn.utils.rnn.pack_padded_sequence(x, torch.as_tensor(list(lens), device="cpu"), batch_first=True,
                                               enforce_sorted=False)
    out, _ = rnn(packed)
    out, _ = nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=total)
    return out


class OracleJointNet(nn.Module):
    """Same parameter names/layouts as the reference's JointNet so state_dicts interchange."""

    class _Enc(nn.Module):
        def __init__(self, input_size, hidden_size, output_size, num_layers, rnn_type="lstm", dropout=0.2,
                     bidirectional=True):
            super().__init__()
            cell = {"lstm": nn.LSTM, "gru": nn.GRU, "rnn": nn.RNN}[rnn_type.lower()]  # encoder.py:48-52
            self.rnn = cell(input_size, hidden_size, num_layers, bias=True, batch_first=True,
                            dropout=dropout if num_layers > 1 else 0.0, bidirectional=bidirectional)
            self.out_proj = nn.Linear(hidden_size * (2 if bidirectional else 1), output_size)

        def forward(self, x, lens):
            return self.out_proj(_packed_lstm(self.rnn, x, lens))

    class _Dec(nn.Module):
        def __init__(self, embedding_size, pad_token_id, hidden_size, output_size, num_layers, rnn_type="lstm",
                     dropout=0.2):
            super().__init__()
            cell = {"lstm": nn.LSTM, "gru": nn.GRU, "rnn": nn.RNN}[rnn_type.lower()]  # decoder.py:51-55
            self.embedding = nn.Embedding(embedding_size, hidden_size, padding_idx=pad_token_id)
            self.rnn = cell(hidden_size, hidden_size, num_layers, bias=True, batch_first=True,
                            dropout=dropout if num_layers > 1 else 0.0, bidirectional=False)
            self.out_proj = nn.Linear(hidden_size, output_size)

        def forward(self, tokens, lens):
            return self.out_proj(_packed_lstm(self.rnn, self.embedding(tokens), lens))

        def forward_with_hidden(self, tokens, lens):
            """decoder.py:102-126 including its second return value: the packed RNN's final states.  The reference packs
            the batch sorted by descending length (decoder.py:105-111) and un-sorts only `outputs` (:116-120), so
            `hidden_states` stay in that sorted batch order."""
            x = self.embedding(tokens)
            lens_t = torch.as_tensor(list(lens), device="cpu")
            packed = nn.utils.rnn.pack_padded_sequence(x, lens_t, batch_first=True, enforce_sorted=False)
            out, hidden = self.rnn(packed)                      # hidden comes back in the ORIGINAL batch order here
            out, _ = nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=x.size(1))
            order = torch.sort(lens_t, descending=True)[1]      # decoder.py:106
            hidden = tuple(h[:, order] for h in hidden) if isinstance(hidden, tuple) else hidden[:, order]
            return self.out_proj(out), hidden

    def __init__(self, transnet_params: dict, prednet_params: dict, num_classes: int):
        super().__init__()
        self.encoder = self._Enc(**transnet_params)
        self.decoder = self._Dec(**prednet_params)
        self.fc = nn.Linear(transnet_params["output_size"] + prednet_params["output_size"], num_classes)

    def joint(self, enc: torch.Tensor, dec: torch.Tensor) -> torch.Tensor:
        """MATERIALISING joint exactly as transducer.py:58-69 describes (this is what the CPU path costs)."""
        T, U1 = enc.size(1), dec.size(1)
        e = enc[:, :, None, :].expand(-1, -1, U1, -1)
        d = dec[:, None, :, :].expand(-1, T, -1, -1)
        return self.fc(F.gelu(torch.cat((e, d), dim=-1), approximate="tanh"))

    def joint_separable(self, enc: torch.Tensor, dec: torch.Tensor) -> torch.Tensor:
        """The same logits without the (B,T,U+1,2*O) concat: GELU is element-wise and fc is linear, so
        fc(gelu(cat(e, d))) = gelu(e) W[:, :O_e]^T + gelu(d) W[:, O_e:]^T + b.  For float64 checks at sizes where the
        concat does not fit (config 3 / 5: 8-10 GB per copy); equality with `joint` is pinned at small sizes by
        tests/test_oracle_networks.py."""
        Oe = enc.size(-1)
        A = F.gelu(enc, approximate="tanh") @ self.fc.weight[:, :Oe].T
        C = F.gelu(dec, approximate="tanh") @ self.fc.weight[:, Oe:].T
        return A[:, :, None, :] + C[:, None, :, :] + self.fc.bias

    def forward(self, audios, audio_lens, texts, text_lens, separable: bool = False):
        enc, dec = self.encoder(audios, audio_lens), self.decoder(texts, text_lens)
        return self.joint_separable(enc, dec) if separable else self.joint(enc, dec)

    @torch.no_grad()
    def recognize_greedy(self, audios, audio_lens, blank: int, max_iters: int = 3, return_margin: bool = False,
                         visit_padded_frames: bool = False):
        """Greedy search restated from networks/transducer.py:95-145 (decoder single step: decoder.py:121-123; 1-D joint:
        transducer.py:64-69).  Per utterance: prediction net primed with [[blank]] and zero state; for every frame of the
        utterance (the reference decodes one utterance per call, so transducer.py:115's max_length is its own length;
        visit_padded_frames=True walks the padded batch length instead, as a batched reference call would) up to
        `max_iters` symbols: argmax of the joint (softmax dropped: monotone);
        blank ends the frame; a non-blank symbol is appended unless it repeats the last appended one (:132-133), and
        always advances the prediction net (:135-136).  Returns one python list per utterance (the reference stacks them,
        which only works for B == 1) and optionally the smallest top-1/top-2 logit gap seen (how robust the argmax
        decisions are to fp32 rounding)."""
        enc = self.encoder(audios, audio_lens)  # (B,T,Oe)
        dec_net = self.decoder
        out, margin = [], float("inf")
        for b in range(enc.size(0)):
            toks, last = [], blank
            emb = dec_net.embedding(torch.tensor([[blank]], dtype=torch.long))
            y, state = dec_net.rnn(emb, None)
            d = dec_net.out_proj(y).view(-1)
            for t in range(enc.size(1) if visit_padded_frames else int(audio_lens[b])):
                for _ in range(max_iters):
                    z = self.fc(F.gelu(torch.cat((enc[b, t], d)), approximate="tanh"))
                    top2 = torch.topk(z, 2).values
                    margin = min(margin, float(top2[0] - top2[1]))
                    k = int(z.argmax())
                    if k == blank:
                        break
                    if k != last:
                        toks.append(k)
                        last = k
                    y, state = dec_net.rnn(dec_net.embedding(torch.tensor([[k]], dtype=torch.long)), state)
                    d = dec_net.out_proj(y).view(-1)
            out.append(toks)
        return (out, margin) if return_margin else out


def training_loss(net: OracleJointNet, batch, blank: int = 0, separable: bool = False) -> torch.Tensor:
    """model.py:54-57 restated: unpack the 7-tuple, forward, mean RNN-T loss.  `separable`: the concat-free form of the joint
    (same logits; for float64 checks at config 3 / 5 sizes)."""
    audios, audio_lens, t_lens, texts, text_lens, targets, u_lens = batch
    logits = net(audios, audio_lens, texts, text_lens, separable=separable)
    return rnnt_loss_mean(logits, targets, t_lens, u_lens, blank)


# ----------------------------------------------------------------------------------------------------
# independent numpy LSTM — second opinion on torch.nn.LSTM's packed semantics
# ----------------------------------------------------------------------------------------------------
def lstm_layer_np(x: np.ndarray, lens: Sequence[int], w_ih, w_hh, b_ih, b_hh, reverse: bool) -> np.ndarray:
    """One direction of one layer, float64, python loops.  x (B,T,I) -> (B,T,H); zeros for t >= T_b."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    y = np.zeros((B, T, H))
    sig = lambda a: 1.0 / (1.0 + np.exp(-a))
    for b in range(B):
        h = np.zeros(H)
        c = np.zeros(H)
        order = range(lens[b] - 1, -1, -1) if reverse else range(lens[b])
        for t in order:
            g = w_ih @ x[b, t] + b_ih + w_hh @ h + b_hh
            i, f, gg, o = sig(g[:H]), sig(g[H:2 * H]), np.tanh(g[2 * H:3 * H]), sig(g[3 * H:])
            c = f * c + i * gg
            h = o * np.tanh(c)
            y[b, t] = h
    return y


def lstm_stack_np(x: np.ndarray, lens, sd: Dict[str, np.ndarray], prefix: str, num_layers: int,
                  bidirectional: bool) -> np.ndarray:
    """Multi-layer (bi)LSTM from a torch-style state dict (dropout off)."""
    cur = x.astype(np.float64)
    for k in range(num_layers):
        outs = []
        for suffix, rev in (("", False),) + ((("_reverse", True),) if bidirectional else ()):
            p = lambda n: np.asarray(sd[f"{prefix}{n}_l{k}{suffix}"], dtype=np.float64)
            outs.append(lstm_layer_np(cur, lens, p("weight_ih"), p("weight_hh"), p("bias_ih"), p("bias_hh"), rev))
        cur = np.concatenate(outs, axis=-1)
    return cur


# ----------------------------------------------------------------------------------------------------
# synthetic 7-tuple batches (dataloader.py:16-49 layout; SURVEY §8d recipe)
# ----------------------------------------------------------------------------------------------------
def make_batch(B: int, T: int, U: int, V: int, n_mels: int = 80, ragged: bool = False, seed: int = 1234,
               blank: int = 0):
    g = torch.Generator().manual_seed(seed)
    audios = torch.randn(B, T, n_mels, generator=g)
    if ragged:
        t_list = torch.randint(max(1, T // 2), T + 1, (B,), generator=g).tolist()
        t_list[0] = T
        u_list = [max(1, round(U * t / T)) for t in t_list]
        u_list[0] = U
    else:
        t_list, u_list = [T] * B, [U] * B
    targets = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32)
    for b in range(B):
        audios[b, t_list[b]:] = 0.0
        targets[b, u_list[b]:] = blank
    texts = torch.cat([torch.full((B, 1), blank, dtype=torch.int64), targets.to(torch.int64)], dim=1)
    text_lens = [u + 1 for u in u_list]
    return (audios, t_list, torch.tensor(t_list, dtype=torch.int32), texts, text_lens, targets,
            torch.tensor(u_list, dtype=torch.int32))
