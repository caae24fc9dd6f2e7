"""The oracle's greedy-search restatement vs fixtures produced by the REFERENCE's JointNet.recognize_greedy
(tests/golden/d*_greedy.npz, made by tests/golden/make_golden_decode.py)."""
import os

import numpy as np
import pytest
import torch

from oracle.rnnt_oracle import OracleJointNet

DECODE_CONFIGS = {
    "d1_greedy": (dict(input_size=80, hidden_size=128, output_size=128, num_layers=1, dropout=0.0, bidirectional=True),
                  dict(embedding_size=72, pad_token_id=0, hidden_size=128, output_size=128, num_layers=1, dropout=0.0), 72),
    "d2_greedy": (dict(input_size=12, hidden_size=16, output_size=8, num_layers=2, rnn_type="gru", dropout=0.0, bidirectional=True),
                  dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=2, rnn_type="lstm", dropout=0.0), 10),
    "d3_greedy": (dict(input_size=12, hidden_size=16, output_size=8, num_layers=1, rnn_type="rnn", dropout=0.0, bidirectional=True),
                  dict(embedding_size=10, pad_token_id=3, hidden_size=16, output_size=8, num_layers=1, rnn_type="gru", dropout=0.0), 10),
}


def fixture_tokens(g):
    return [g["tokens"][b, :n].tolist() for b, n in enumerate(g["ntok"].tolist())]


@pytest.mark.parametrize("tag", list(DECODE_CONFIGS))
def test_oracle_greedy_matches_reference_fixture(golden_dir, tag):
    g = dict(np.load(os.path.join(golden_dir, tag + ".npz")))
    tn, pn, V = DECODE_CONFIGS[tag]
    net = OracleJointNet(tn, pn, V)
    net.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")})
    net.eval()
    audios, t_list = torch.from_numpy(g["audios"]), g["t_lens"].tolist()
    want = fixture_tokens(g)
    # batched call (per-utterance frame counts) and one-utterance-at-a-time calls, as the reference was run
    got, margin = net.recognize_greedy(audios, t_list, pn["pad_token_id"], int(g["max_iters"]), return_margin=True)
    assert got == want
    assert margin >= 1e-3  # no decision in the fixture hangs on fp32 rounding
    for b, t in enumerate(t_list):
        assert net.recognize_greedy(audios[b:b + 1, :t], [t], pn["pad_token_id"], int(g["max_iters"])) == [want[b]]
    # fixture sanity: blank never emitted, no immediate repeats (transducer.py:132-133), bound T*max_iters
    for b, toks in enumerate(want):
        assert pn["pad_token_id"] not in toks
        assert all(x != y for x, y in zip(toks, toks[1:]))
        assert len(toks) <= t_list[b] * int(g["max_iters"])


def test_oracle_greedy_padded_frames_mode_extends_only():
    """Walking the padded frames (what a batched reference call does) can only append to the per-utterance result."""
    tn, pn, V = DECODE_CONFIGS["d2_greedy"]
    torch.manual_seed(3)
    net = OracleJointNet(tn, pn, V).eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(3.0)
    audios = torch.randn(3, 12, 12)
    lens = [12, 7, 3]
    own = net.recognize_greedy(audios, lens, 0, 2)
    padded = net.recognize_greedy(audios, lens, 0, 2, visit_padded_frames=True)
    assert own[0] == padded[0]
    for a, b in zip(own, padded):
        assert b[:len(a)] == a
