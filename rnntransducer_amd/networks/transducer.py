"""JointNet — encoder + prediction net + joint, on the MI355X HIP path.

Mirrors networks/transducer.py:27-39 (ctor), :41-71 (joint), :73-93 (forward) of the reference: attributes
`encoder`, `decoder`, `fc` (Linear(O_e + O_d -> V), enc half first: transducer.py:64), GELU(tanh).

The reference's joint materialises (B,T,U+1,2*O) three times; here logits[b,t,u,:] = A[b,t,:] + C[b,u,:] + bias
with A = gelu(enc) W_e^T, C = gelu(dec) W_d^T (GELU is element-wise, fc is linear: SURVEY.md §0), so:
  * `loss(...)`  — fused joint + log-softmax + alpha/beta + gradient; (B,T,U+1,V) is never built;
  * `joint()/forward()` — still return the full logits tensor for callers that ask for it.
Greedy/beam decoding (transducer.py:95-361) is out of scope.
"""
import torch
import torch.nn as nn

from ..ops import JointLogitsFn, JointLossFn
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
        """(B,T,O_e), (B,U+1,O_d) -> logits (B,T,U+1,V) (materialising; transducer.py:54-69)."""
        if encoder_outputs.dim() != 3 or decoder_outputs.dim() != 3:
            raise NotImplementedError("1-D single-step joint (decoding, transducer.py:125,309) is out of scope")
        return JointLogitsFn.apply(encoder_outputs.transpose(0, 1).contiguous(),
                                   decoder_outputs.transpose(0, 1).contiguous(), self.fc.weight, self.fc.bias)

    def forward(self, input_audios, audio_lengths, input_texts, text_lengths) -> torch.Tensor:
        dev = input_audios.device
        enc = self.encoder.forward_time_major(input_audios, lengths_to_device(audio_lengths, dev))
        dec = self.decoder.forward_time_major(input_texts, lengths_to_device(text_lengths, dev))
        return JointLogitsFn.apply(enc, dec, self.fc.weight, self.fc.bias)

    def loss(self, input_audios, tensor_audio_lengths, input_texts, targets, target_lengths, blank: int) -> torch.Tensor:
        """Per-utterance -log P(y|x), shape (B,), through the fused path (no (B,T,U+1,V) tensor)."""
        dev = input_audios.device
        t_lens = lengths_to_device(tensor_audio_lengths, dev)
        u_lens = lengths_to_device(target_lengths, dev)
        enc = self.encoder.forward_time_major(input_audios, t_lens)
        dec = self.decoder.forward_time_major(input_texts, u_lens + 1)  # text length = label length + 1 (dataloader.py:39-40)
        return JointLossFn.apply(enc, dec, self.fc.weight, self.fc.bias, targets, t_lens, u_lens, blank)
