"""JointNet — encoder + prediction net + joint, on the MI355X HIP path.

Mirrors networks/transducer.py:27-39 (ctor), :41-71 (joint), :73-93 (forward) of the reference: attributes
`encoder`, `decoder`, `fc` (Linear(O_e + O_d -> V), enc half first: transducer.py:64), GELU(tanh).

The reference's joint materialises (B,T,U+1,2*O) three times; here logits[b,t,u,:] = A[b,t,:] + C[b,u,:] + bias
with A = gelu(enc) W_e^T, C = gelu(dec) W_d^T (GELU is element-wise, fc is linear: SURVEY.md §0), so:
  * `loss(...)`  — fused joint + log-softmax + alpha/beta + gradient; (B,T,U+1,V) is never built;
  * `joint()/forward()` — still return the full logits tensor for callers that ask for it.
`recognize_greedy` (transducer.py:95-145) runs as ONE kernel launch (a workgroup per utterance, csrc/decode.hip) instead of
a host loop with a device sync per symbol; beam search (transducer.py:147-361) stays out of scope.
"""
import torch
import torch.nn as nn

from ..ops import JointLogitsFn, JointLossFn, greedy_decode
from .decoder import TextPredNet
from .encoder import AudioTransNet, HipLinear, lengths_to_device


class JointNet(nn.Module):
    def __init__(self, transnet_params: dict, prednet_params: dict, num_classes: int):
        super().__init__()
        self.encoder = AudioTransNet(**transnet_params)
        self.decoder = TextPredNet(**prednet_params)
        self.num_classes = num_classes
        self.enc_out = transnet_params["output_size"]
        self.dec_out = prednet_params["output_size"]
        # parameter container only: fc is applied inside the fused kernels (A/C pre-GEMMs), never as one Linear
        self.fc = HipLinear(self.enc_out + self.dec_out, num_classes)

    def joint(self, encoder_outputs: torch.Tensor, decoder_outputs: torch.Tensor) -> torch.Tensor:
        """(B,T,O_e), (B,U+1,O_d) -> logits (B,T,U+1,V) (materialising; transducer.py:54-69).  1-D inputs (one encoder frame,
        one prediction-net output: the search loops' call at transducer.py:125,309) -> logits (V,)."""
        if encoder_outputs.dim() == 1 and decoder_outputs.dim() == 1:
            from ..ops import LinearFn
            z = torch.cat((encoder_outputs, decoder_outputs)).unsqueeze(0)
            return LinearFn.apply(torch.nn.functional.gelu(z, approximate="tanh"), self.fc.weight, self.fc.bias).squeeze(0)
        if encoder_outputs.dim() != 3 or decoder_outputs.dim() != 3:
            raise ValueError("joint takes (B,T,O) with (B,U+1,O), or two 1-D vectors")
        return JointLogitsFn.apply(encoder_outputs.transpose(0, 1).contiguous(),
                                   decoder_outputs.transpose(0, 1).contiguous(), self.fc.weight, self.fc.bias)

    def forward(self, input_audios, audio_lengths, input_texts, text_lengths) -> torch.Tensor:
        dev = input_audios.device
        enc = self.encoder.forward_time_major(input_audios, lengths_to_device(audio_lengths, dev))
        dec = self.decoder.forward_time_major(input_texts, lengths_to_device(text_lengths, dev))
        return JointLogitsFn.apply(enc, dec, self.fc.weight, self.fc.bias)

    def loss(self, input_audios, tensor_audio_lengths, input_texts, targets, target_lengths, blank: int,
             reduction: str = "none", audio_lengths=None) -> torch.Tensor:
        """-log P(y|x) through the fused path (no (B,T,U+1,V) tensor): per utterance, shape (B,) (reduction "none"), or the 0-d
        "mean" / "sum" over the batch (model.py:39 builds the reference's loss with reduction="mean").
        `audio_lengths`: the python list of frame counts the reference's collate hands over next to the tensor (dataloader.py:20,49).
        With it a ragged batch is handled as the reference handles it (networks/encoder.py:93-96: sort by length, pack): rows are
        sorted by descending length so that the recurrences' sync groups are length-homogeneous, and a valid-frame table lets the
        big products and the recurrences skip the padding (ops.RaggedPlan) — all planned on the host, no device synchronisation.
        Results do not depend on it."""
        dev = input_audios.device
        t_lens = lengths_to_device(tensor_audio_lengths, dev)
        u_lens = lengths_to_device(target_lengths, dev)
        enc_lens, inv = t_lens, None
        T, B = input_audios.size(1), input_audios.size(0)
        if audio_lengths is not None and len(audio_lengths) == B and B > 1 and min(audio_lengths) < T:
            from ..ops import RaggedPlan
            host = [int(n) for n in audio_lengths]
            order = sorted(range(B), key=lambda b: (-host[b], b))
            if order != list(range(B)):   # length-sorted rows (encoder.py:94-96); the small per-utterance tensors follow, nll is un-sorted below
                perm = torch.tensor(order, dtype=torch.int64, device=dev)
                input_audios, input_texts, targets = (x.index_select(0, perm) for x in (input_audios, input_texts, targets))
                t_lens, u_lens = t_lens.index_select(0, perm), u_lens.index_select(0, perm)
                host = [host[b] for b in order]
                if reduction == "none":
                    inv = torch.empty_like(perm)
                    inv[perm] = torch.arange(B, dtype=torch.int64, device=dev)
            enc_lens = RaggedPlan(host, T, dev)
        enc = self.encoder.forward_time_major(input_audios, enc_lens)
        dec = self.decoder.forward_time_major(input_texts, u_lens + 1)  # text length = label length + 1 (dataloader.py:39-40)
        out = JointLossFn.apply(enc, dec, self.fc.weight, self.fc.bias, targets, t_lens, u_lens, blank, torch.is_grad_enabled(), reduction)
        return out if inv is None else out.index_select(0, inv)

    @torch.no_grad()
    def recognize_greedy(self, inputs: torch.Tensor, inputs_lengths, blank_token_id: int, max_iters: int = 3,
                         visit_padded_frames: bool = False):
        """Greedy search, same result as transducer.py:95-145: per frame up to `max_iters` non-blank symbols, a symbol
        equal to the previously appended one is dropped (but still advances the prediction net).  Returns a LongTensor
        (1, n) for a single utterance, as the reference's `torch.stack` does.  For B > 1 (where the reference's stack
        raises on ragged outputs) a list of B 1-D LongTensors, each equal to what the reference returns for that
        utterance decoded alone; `visit_padded_frames=True` instead walks all max(lengths) frames for every utterance,
        which is what the reference's loop bound (transducer.py:115,121) does inside a batched call."""
        if self.training:
            raise RuntimeError("recognize_greedy expects eval() mode (dropout inactive), like the reference's validation_step")
        dev = inputs.device
        t_lens = lengths_to_device(inputs_lengths, dev)
        enc = self.encoder.forward_time_major(inputs, t_lens)
        dec = self.decoder
        tokens, ntok = greedy_decode(enc, self.fc.weight, self.fc.bias, dec.embedding.weight, dec.rnn.flat_weights(),
                                     dec.rnn.CELL, dec.out_proj.weight, dec.out_proj.bias, blank_token_id, max_iters,
                                     None if visit_padded_frames else t_lens)
        n = ntok.tolist()  # the only host sync of the decode
        outs = [tokens[b, :n[b]] for b in range(tokens.shape[0])]
        return outs[0].unsqueeze(0) if len(outs) == 1 else outs
