"""Greedy decoding on the GPU (one kernel launch, csrc/decode.hip) vs
  * fixtures produced by the REFERENCE's JointNet.recognize_greedy (tests/golden/d*_greedy.npz), bit-exact token ids;
  * the CPU oracle restatement on seeded synthetic inputs at the config-2 layer sizes (H=512)."""
import os

import numpy as np
import pytest
import torch

from tests.test_oracle_decode import DECODE_CONFIGS, fixture_tokens

pytestmark = pytest.mark.gpu


def _jointnet(tn, pn, V):
    from rnntransducer_amd.networks import JointNet
    return JointNet(dict(tn), dict(pn), V)


@pytest.mark.parametrize("tag", list(DECODE_CONFIGS))
def test_greedy_matches_reference_fixture(golden_dir, tag):
    g = dict(np.load(os.path.join(golden_dir, tag + ".npz")))
    tn, pn, V = DECODE_CONFIGS[tag]
    net = _jointnet(tn, pn, V)
    net.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param/")})
    net = net.cuda().eval()
    audios, t_list = torch.from_numpy(g["audios"]).cuda(), g["t_lens"].tolist()
    want = fixture_tokens(g)
    got = net.recognize_greedy(audios, t_list, pn["pad_token_id"], int(g["max_iters"]))
    assert [x.tolist() for x in got] == want
    # single utterance: (1, n) LongTensor like the reference's torch.stack
    one = net.recognize_greedy(audios[:1, :t_list[0]].contiguous(), [t_list[0]], pn["pad_token_id"], int(g["max_iters"]))
    assert one.dtype == torch.int64 and one.shape == (1, len(want[0])) and one[0].tolist() == want[0]


@pytest.mark.parametrize("cells", [("lstm", "lstm", 2), ("gru", "lstm", 1), ("lstm", "gru", 2), ("lstm", "rnn", 1)])
def test_greedy_vs_oracle_config2_sizes(cells):
    from oracle.rnnt_oracle import OracleJointNet
    enc_cell, dec_cell, dec_layers = cells
    tn = dict(input_size=80, hidden_size=256, output_size=320, num_layers=1, rnn_type=enc_cell, dropout=0.0, bidirectional=True)
    pn = dict(embedding_size=72, pad_token_id=0, hidden_size=512, output_size=320, num_layers=dec_layers, rnn_type=dec_cell,
              dropout=0.0)
    torch.manual_seed(17)
    ora = OracleJointNet(tn, pn, 72).eval()
    with torch.no_grad():
        for n, p in ora.named_parameters():
            p.mul_(6.0 if n.startswith("fc.") else 3.0)
        ora.decoder.embedding.weight[0].zero_()
    audios = torch.randn(4, 60, 80)
    lens = [60, 48, 31, 5]
    for b, t in enumerate(lens):
        audios[b, t:] = 0
    net = _jointnet(tn, pn, 72)
    net.load_state_dict(ora.state_dict())
    net = net.cuda().eval()
    for padded in (False, True):
        want, margin = ora.recognize_greedy(audios, lens, 0, 3, return_margin=True, visit_padded_frames=padded)
        got = [x.tolist() for x in net.recognize_greedy(audios.cuda(), lens, 0, 3, visit_padded_frames=padded)]
        assert sum(len(w) for w in want) > 40  # the case really decodes something
        if margin >= 1e-4:
            assert got == want
        else:  # a near-tie somewhere: fp32 summation order may legitimately flip it; everything before must agree
            for a, b in zip(got, want):
                if a != b:
                    k = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
                    assert k >= 3, (margin, a[:12], b[:12])


def test_greedy_max_iters_one_and_training_mode_guard():
    from oracle.rnnt_oracle import OracleJointNet
    tn, pn, V = DECODE_CONFIGS["d2_greedy"]
    torch.manual_seed(5)
    ora = OracleJointNet(tn, pn, V).eval()
    with torch.no_grad():
        for p in ora.parameters():
            p.mul_(3.0)
    net = _jointnet(tn, pn, V)
    net.load_state_dict(ora.state_dict())
    net = net.cuda()
    audios, lens = torch.randn(2, 20, 12), [20, 13]
    with pytest.raises(RuntimeError):
        net.train().recognize_greedy(audios.cuda(), lens, 0, 1)
    got = net.eval().recognize_greedy(audios.cuda(), lens, 0, 1)
    assert [x.tolist() for x in got] == ora.recognize_greedy(audios, lens, 0, 1)
    assert all(len(x) <= t for x, t in zip(got, lens))
