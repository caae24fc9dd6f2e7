"""Generates tests/golden/d*_greedy.npz from the REFERENCE's own JointNet.recognize_greedy (networks/transducer.py:95-145).

Run ONLY in the build container (where /root/reference is mounted):  python tests/golden/make_golden_decode.py
The fixtures are data (parameters, inputs, decoded token ids); no reference source or bytecode is written anywhere.
Import method: as tests/golden/make_golden.py (three empty pyctcdecode placeholder modules).

The reference can only decode one utterance per call (its final torch.stack needs equal lengths), so each fixture
utterance is decoded in its own B=1 call, at full padded length T_max of the fixture batch (zero-padded audio with the
true length passed, which is what a batched call would feed the encoder).  Random-init weights decode to (almost)
nothing, so every parameter is scaled up after init to give varied, multi-symbol-per-frame outputs; `margin` records the
smallest top-1/top-2 logit gap the oracle restatement sees on the same decode — fixtures are only kept if it is
>= 1e-3, so fp32 summation order cannot flip a decision.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
for name, attrs in (("pyctcdecode", ["LanguageModel"]), ("pyctcdecode.language_model", ["HotwordScorer"]),
                    ("pyctcdecode.constants", ["DEFAULT_HOTWORD_WEIGHT"])):
    mod = types.ModuleType(name)
    for a in attrs:
        setattr(mod, a, None)
    sys.modules[name] = mod

from networks import JointNet  # noqa: E402  (the reference's)
from oracle.rnnt_oracle import OracleJointNet  # noqa: E402


def run(tag, transnet, prednet, V, t_list, seed, scale, max_iters=3):
    torch.manual_seed(seed)
    net = JointNet(dict(transnet), dict(prednet), V)
    with torch.no_grad():
        for n, p in net.named_parameters():
            p.mul_(scale["fc"] if n.startswith("fc.") else scale["rest"])
        net.decoder.embedding.weight[prednet["pad_token_id"]].zero_()
    net.eval()
    B, T = len(t_list), max(t_list)
    g = torch.Generator().manual_seed(seed + 1)
    audios = torch.randn(B, T, transnet["input_size"], generator=g)
    for b in range(B):
        audios[b, t_list[b]:] = 0.0
    toks = []
    for b in range(B):
        out = net.recognize_greedy(audios[b:b + 1], [t_list[b]], prednet["pad_token_id"], max_iters)
        toks.append(out.reshape(-1).tolist())
    ora = OracleJointNet(dict(transnet), dict(prednet), V)
    ora.load_state_dict(net.state_dict())
    ora.eval()
    o_toks, margin = ora.recognize_greedy(audios, t_list, prednet["pad_token_id"], max_iters, return_margin=True)
    assert o_toks == toks, (tag, o_toks, toks)
    assert margin >= 1e-3, (tag, margin)
    n = max(len(t) for t in toks)
    tok_arr = np.full((B, max(n, 1)), -1, np.int64)
    for b, t in enumerate(toks):
        tok_arr[b, :len(t)] = t
    out = {"audios": audios.numpy(), "t_lens": np.array(t_list, np.int32), "tokens": tok_arr,
           "ntok": np.array([len(t) for t in toks], np.int32), "max_iters": np.int32(max_iters),
           "margin": np.float64(margin)}
    for k, v in net.state_dict().items():
        out["param/" + k] = v.numpy()
    path = os.path.join(HERE, tag + ".npz")
    np.savez_compressed(path, **out)
    print(tag, "ntok", [len(t) for t in toks], "distinct", len({x for t in toks for x in t}), "margin %.3g" % margin,
          "bytes", os.path.getsize(path))


if __name__ == "__main__":
    # D1: config-1-shaped network (1x128 bi-LSTM encoder, 1x128 LSTM prediction net, V=72), ragged batch
    run("d1_greedy",
        dict(input_size=80, hidden_size=128, output_size=128, num_layers=1, rnn_type="lstm", dropout=0.0,
             bidirectional=True),
        dict(embedding_size=72, pad_token_id=0, hidden_size=128, output_size=128, num_layers=1, rnn_type="lstm",
             dropout=0.0),
        72, [40, 33, 25], seed=5, scale=dict(fc=6.0, rest=3.0))
    # D2: the shipped config's cell mix (bi-GRU encoder, 2-layer LSTM prediction net), scaled down; max_iters=2
    run("d2_greedy",
        dict(input_size=12, hidden_size=16, output_size=8, num_layers=2, rnn_type="gru", dropout=0.0,
             bidirectional=True),
        dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=2, rnn_type="lstm",
             dropout=0.0),
        10, [15, 9, 12, 4], seed=7, scale=dict(fc=8.0, rest=3.0), max_iters=2)
    # D3: GRU prediction net, Elman encoder, non-zero blank id
    run("d3_greedy",
        dict(input_size=12, hidden_size=16, output_size=8, num_layers=1, rnn_type="rnn", dropout=0.0,
             bidirectional=True),
        dict(embedding_size=10, pad_token_id=3, hidden_size=16, output_size=8, num_layers=1, rnn_type="gru",
             dropout=0.0),
        10, [14, 10], seed=9, scale=dict(fc=8.0, rest=3.0))
