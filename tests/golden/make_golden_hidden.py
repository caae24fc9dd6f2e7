"""Generates tests/golden/h*_hidden.npz: what the REFERENCE's TextPredNet.forward (training branch, networks/decoder.py:102-126)
returns as `hidden_states` next to the outputs — the packed RNN's final states, in its length-sorted batch order.

Run ONLY in the build container (where /root/reference is mounted):  python tests/golden/make_golden_hidden.py
Import method as make_golden.py (three pyctcdecode names used only by beam search are empty placeholder modules).
The fixtures are data (token ids, lengths, state_dict, outputs, hidden states); no reference source is written anywhere.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
for name, attrs in (("pyctcdecode", ["LanguageModel"]), ("pyctcdecode.language_model", ["HotwordScorer"]),
                    ("pyctcdecode.constants", ["DEFAULT_HOTWORD_WEIGHT"])):
    mod = types.ModuleType(name)
    for a in attrs:
        setattr(mod, a, None)
    sys.modules[name] = mod

from networks import TextPredNet  # noqa: E402  (the reference's)


def run(tag, prednet, lens, seed):
    torch.manual_seed(seed)
    net = TextPredNet(**prednet).double().eval()
    B, U1, V = len(lens), max(lens), prednet["embedding_size"]
    g = torch.Generator().manual_seed(seed + 1)
    tokens = torch.randint(1, V, (B, U1), generator=g)
    tokens[:, 0] = prednet["pad_token_id"]
    for b in range(B):
        tokens[b, lens[b]:] = prednet["pad_token_id"]
    with torch.no_grad():
        out, hidden = net(tokens, lens)
    rec = {"tokens": tokens.numpy(), "lens": np.array(lens, np.int32), "out": out.numpy()}
    if isinstance(hidden, tuple):
        rec["h_n"], rec["c_n"] = hidden[0].numpy(), hidden[1].numpy()
    else:
        rec["h_n"] = hidden.numpy()
    for k, v in net.state_dict().items():
        rec["param/" + k] = v.numpy()
    path = os.path.join(HERE, tag + ".npz")
    np.savez_compressed(path, **rec)
    print(tag, {k: v.shape for k, v in rec.items() if not k.startswith("param/")}, os.path.getsize(path))


if __name__ == "__main__":
    run("h1_lstm_hidden", dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=2, rnn_type="lstm",
                               dropout=0.0), [3, 6, 6, 1, 4], seed=5)   # a tie in the lengths pins the sort's tie-break
    run("h2_gru_hidden", dict(embedding_size=12, pad_token_id=0, hidden_size=8, output_size=8, num_layers=1, rnn_type="gru",
                              dropout=0.0), [5, 2, 7], seed=6)
