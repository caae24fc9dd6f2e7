"""Generates tests/golden/g1_cfg1.npz and g2_stack.npz from the REFERENCE's own networks/*.

Run ONLY in the build container (where /root/reference is mounted):  python tests/golden/make_golden.py
The fixtures are data (inputs + outputs); no reference source or bytecode is written anywhere.

Import method: /root/reference on sys.path; the three pyctcdecode names that
networks/transducer.py:21-23 imports for beam search only are registered as empty placeholder modules.
model.py itself cannot be imported (pytorch_lightning / warprnnt_pytorch / torchaudio / torchmetrics are
absent), so the loss applied on top of the reference logits is oracle.rnnt_oracle.rnnt_nll_torch
(autograd through an independent float64-capable DP) with reduction "mean" as model.py:39 asks.
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
from oracle.rnnt_oracle import rnnt_nll_torch  # noqa: E402


def run(tag, transnet, prednet, V, t_list, u_list, seed):
    torch.manual_seed(seed)
    net = JointNet(dict(transnet), dict(prednet), V).double()  # float64 so the fixture is a tight pin
    net.train()  # dropout is 0.0 in every fixture config, so train == eval numerically
    B, T, U = len(t_list), max(t_list), max(u_list)
    g = torch.Generator().manual_seed(seed + 1)
    audios = torch.randn(B, T, transnet["input_size"], generator=g, dtype=torch.float64)
    targets = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32)
    for b in range(B):
        audios[b, t_list[b]:] = 0.0
        targets[b, u_list[b]:] = 0
    texts = torch.cat([torch.zeros(B, 1, dtype=torch.int64), targets.long()], 1)
    text_lens = [u + 1 for u in u_list]
    enc = net.encoder(audios, t_list)
    dec, _ = net.decoder(texts, text_lens)
    logits = net.joint(enc, dec)
    assert torch.equal(logits, net(audios, t_list, texts, text_lens))
    nll = rnnt_nll_torch(logits, targets.tolist(), t_list, u_list, blank=0)
    loss = nll.mean()
    loss.backward()
    out = {"audios": audios.numpy(), "t_lens": np.array(t_list, np.int32), "u_lens": np.array(u_list, np.int32),
           "targets": targets.numpy(), "texts": texts.numpy(), "enc": enc.detach().numpy(),
           "dec": dec.detach().numpy(), "logits": logits.detach().numpy(), "nll": nll.detach().numpy(),
           "loss": loss.detach().numpy()}
    for k, v in net.state_dict().items():
        out["param/" + k] = v.numpy()
    for k, p in net.named_parameters():
        out["grad/" + k] = p.grad.numpy()
    path = os.path.join(HERE, tag + ".npz")
    np.savez_compressed(path, **out)
    print(tag, "loss", float(loss), "bytes", os.path.getsize(path))


if __name__ == "__main__":
    # G1: BASELINE config 1 (B=2,T=100,U=20,V=72, 1x128 bi-LSTM enc / 1x128 pred, O=128), ragged lengths
    run("g1_cfg1",
        dict(input_size=80, hidden_size=128, output_size=128, num_layers=1, rnn_type="lstm", dropout=0.0,
             bidirectional=True),
        dict(embedding_size=72, pad_token_id=0, hidden_size=128, output_size=128, num_layers=1, rnn_type="lstm",
             dropout=0.0),
        72, [100, 90], [20, 15], seed=11)
    # G2: 2-layer bidirectional encoder + 2-layer prediction net, very ragged (inter-layer path, reverse start)
    run("g2_stack",
        dict(input_size=12, hidden_size=16, output_size=8, num_layers=2, rnn_type="lstm", dropout=0.0,
             bidirectional=True),
        dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=2, rnn_type="lstm",
             dropout=0.0),
        10, [12, 9, 5], [4, 2, 3], seed=23)
    # G3: the cell mix of the reference's SHIPPED config.json (bidirectional GRU encoder, LSTM prediction net), scaled down
    run("g3_gru",
        dict(input_size=12, hidden_size=16, output_size=8, num_layers=2, rnn_type="gru", dropout=0.0,
             bidirectional=True),
        dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=2, rnn_type="lstm",
             dropout=0.0),
        10, [11, 6, 9], [3, 4, 1], seed=41)
    # G3r: Elman-RNN encoder + GRU prediction net
    run("g3_rnn",
        dict(input_size=12, hidden_size=16, output_size=8, num_layers=2, rnn_type="rnn", dropout=0.0,
             bidirectional=True),
        dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=1, rnn_type="gru",
             dropout=0.0),
        10, [8, 10, 4], [2, 5, 3], seed=43)
    # G2u: uni-directional encoder variant
    run("g2_uni",
        dict(input_size=12, hidden_size=16, output_size=8, num_layers=2, rnn_type="lstm", dropout=0.0,
             bidirectional=False),
        dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=1, rnn_type="lstm",
             dropout=0.0),
        10, [7, 10, 3, 10], [1, 5, 2, 4], seed=37)
