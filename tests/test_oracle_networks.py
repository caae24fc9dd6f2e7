"""The oracle's network restatement vs fixtures generated from the REFERENCE's networks/* (G1, G2)."""
import os

import numpy as np
import pytest
import torch

from oracle.rnnt_oracle import OracleJointNet, lstm_stack_np, rnnt_loss_c

CONFIGS = {
    "g1_cfg1": (dict(input_size=80, hidden_size=128, output_size=128, num_layers=1, dropout=0.0, bidirectional=True),
                dict(embedding_size=72, pad_token_id=0, hidden_size=128, output_size=128, num_layers=1, dropout=0.0), 72),
    "g2_stack": (dict(input_size=12, hidden_size=16, output_size=8, num_layers=2, dropout=0.0, bidirectional=True),
                 dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=2, dropout=0.0), 10),
    "g3_gru": (dict(input_size=12, hidden_size=16, output_size=8, num_layers=2, rnn_type="gru", dropout=0.0, bidirectional=True),
               dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=2, rnn_type="lstm", dropout=0.0), 10),
    "g3_rnn": (dict(input_size=12, hidden_size=16, output_size=8, num_layers=2, rnn_type="rnn", dropout=0.0, bidirectional=True),
               dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=1, rnn_type="gru", dropout=0.0), 10),
    "g2_uni": (dict(input_size=12, hidden_size=16, output_size=8, num_layers=2, dropout=0.0, bidirectional=False),
               dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=1, dropout=0.0), 10),
}


def load(golden_dir, tag):
    return dict(np.load(os.path.join(golden_dir, tag + ".npz")))


@pytest.mark.parametrize("tag", list(CONFIGS))
def test_oracle_matches_reference_fixture(golden_dir, tag):
    g = load(golden_dir, tag)
    tn, pn, V = CONFIGS[tag]
    net = OracleJointNet(tn, pn, V).double()
    net.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")})
    t_list, u_list = g["t_lens"].tolist(), g["u_lens"].tolist()
    audios, texts = torch.from_numpy(g["audios"]), torch.from_numpy(g["texts"])
    enc = net.encoder(audios, t_list)
    dec = net.decoder(texts, [u + 1 for u in u_list])
    logits = net.joint(enc, dec)
    np.testing.assert_allclose(enc.detach().numpy(), g["enc"], atol=1e-12)
    np.testing.assert_allclose(dec.detach().numpy(), g["dec"], atol=1e-12)
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], atol=1e-12)
    # loss + analytic logits-gradient from the C restatement, pushed through autograd of the composite
    nll, dlogits = rnnt_loss_c(logits.detach().numpy(), g["targets"], t_list, u_list, 0)
    np.testing.assert_allclose(nll, g["nll"], rtol=1e-12)
    logits.backward(torch.from_numpy(dlogits) / len(t_list))
    for name, p in net.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), g["grad/" + name], atol=1e-10, err_msg=name)
    # embedding row of the blank/pad token is zero and gets no gradient (decoder.py:69)
    assert np.all(g["param/decoder.embedding.weight"][0] == 0) and np.all(g["grad/decoder.embedding.weight"][0] == 0)


def test_numpy_lstm_second_opinion(golden_dir):
    g = load(golden_dir, "g2_stack")
    sd = {k[6:]: v for k, v in g.items() if k.startswith("param/")}
    y = lstm_stack_np(g["audios"], g["t_lens"].tolist(), sd, "encoder.rnn.", 2, True)
    enc = y @ sd["encoder.out_proj.weight"].T + sd["encoder.out_proj.bias"]
    np.testing.assert_allclose(enc, g["enc"], atol=1e-12)
    # padded frames: LSTM output is zero, so enc == out_proj.bias there
    assert np.all(y[2, 5:] == 0)
    np.testing.assert_allclose(g["enc"][2, 5:], np.broadcast_to(sd["encoder.out_proj.bias"], (7, 8)), atol=1e-15)


def test_joint_is_separable(golden_dir):
    """SURVEY §0 key algebraic fact: logits == A[b,t] + C[b,u] + bias (what the fused kernel exploits)."""
    g = load(golden_dir, "g2_stack")
    W, bias = g["param/fc.weight"], g["param/fc.bias"]
    gelu = lambda x: torch.nn.functional.gelu(torch.from_numpy(x), approximate="tanh").numpy()
    O = g["enc"].shape[-1]
    A = gelu(g["enc"]) @ W[:, :O].T
    C = gelu(g["dec"]) @ W[:, O:].T
    np.testing.assert_allclose(A[:, :, None, :] + C[:, None, :, :] + bias, g["logits"], atol=1e-13)


def test_per_utterance_lstm_equals_the_packed_one():
    """oracle._per_utterance_lstm (used for the float64 checks at config 3 / 5 sizes, where autograd through a PackedSequence is
    O(T^2) on the CPU) is the same function as the packed path the reference fixtures pin: outputs and every gradient to 1e-12,
    2 layers, both directions, LSTM and GRU; ragged batch (one call per utterance) and equal lengths below the padded T (one call)."""
    from oracle import rnnt_oracle as ro
    for cell in (torch.nn.LSTM, torch.nn.GRU):
        torch.manual_seed(3)
        rnn = cell(6, 5, 2, batch_first=True, bidirectional=True).double()
        for lens in ([9, 4, 7, 1], [6, 6, 6, 6]):
            x = torch.randn(4, 9, 6, dtype=torch.float64)
            dy = torch.randn(4, 9, 10, dtype=torch.float64)
            res = []
            for fn in (ro._packed_lstm, ro._per_utterance_lstm):
                rnn.zero_grad()
                xr = x.clone().requires_grad_(True)
                y = fn(rnn, xr, lens)
                y.backward(dy)
                res.append([y.detach(), xr.grad] + [p.grad.clone() for p in rnn.parameters()])
            for a, b in zip(*res):
                assert (a - b).abs().max().item() < 1e-12
            assert torch.all(res[1][0][1, lens[1]:] == 0) and torch.all(res[1][0][3, lens[3]:] == 0)


HIDDEN = {
    "h1_lstm_hidden": dict(embedding_size=10, pad_token_id=0, hidden_size=16, output_size=8, num_layers=2, rnn_type="lstm", dropout=0.0),
    "h2_gru_hidden": dict(embedding_size=12, pad_token_id=0, hidden_size=8, output_size=8, num_layers=1, rnn_type="gru", dropout=0.0),
}


@pytest.mark.parametrize("tag", list(HIDDEN))
def test_oracle_prednet_hidden_states_match_reference_fixture(golden_dir, tag):
    """Second return value of TextPredNet.forward's training branch (decoder.py:115,126), generated by the reference itself
    (tests/golden/make_golden_hidden.py): final states in the reference's length-sorted batch order."""
    g = load(golden_dir, tag)
    dec = OracleJointNet._Dec(**HIDDEN[tag]).double()
    dec.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")})
    out, hidden = dec.forward_with_hidden(torch.from_numpy(g["tokens"]), g["lens"].tolist())
    np.testing.assert_allclose(out.detach().numpy(), g["out"], atol=1e-12)
    if "c_n" in g:
        np.testing.assert_allclose(hidden[0].detach().numpy(), g["h_n"], atol=1e-12)
        np.testing.assert_allclose(hidden[1].detach().numpy(), g["c_n"], atol=1e-12)
    else:
        np.testing.assert_allclose(hidden.detach().numpy(), g["h_n"], atol=1e-12)
