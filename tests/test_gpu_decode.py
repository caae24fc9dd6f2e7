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


def test_validation_step_after_overfitting_one_batch_recovers_labels():
    """End to end through the drop-in surface: training_step (fused loss) + FlatAdamW overfit one small batch, then
    validation_step's on-device greedy search reads (most of) the labels back and agrees token for token with the oracle's
    search on the trained weights (targets have no immediate repeats, which the search collapses: transducer.py:132-133)."""
    from argparse import Namespace
    from rnntransducer_amd import RNNTransducer
    from rnntransducer_amd.data import synthetic_batch
    tn = dict(input_size=80, hidden_size=128, output_size=128, num_layers=1, dropout=0.0, bidirectional=True)
    pn = dict(embedding_size=72, pad_token_id=0, hidden_size=128, output_size=128, num_layers=1, dropout=0.0)
    args = Namespace(learning_rate=4e-3, weight_decay=0.0, warmup_ratio=0.1, final_div_factor=10.0, total_steps=400,
                     move_metrics_to_cpu=False)
    torch.manual_seed(1)
    model = RNNTransducer(pn, tn, dict(num_classes=72), args).cuda().train()
    batch = list(synthetic_batch(2, 60, 12, 72, device="cuda", seed=4))
    targets = batch[5].clone()
    for b in range(targets.size(0)):  # remove immediate repeats, keep ids in [1, V)
        for u in range(1, targets.size(1)):
            if targets[b, u] == targets[b, u - 1]:
                targets[b, u] = targets[b, u] % 71 + 1
    batch[5] = targets
    batch[3] = torch.cat([torch.zeros_like(batch[3][:, :1]), targets.long()], 1)
    # make the "audio" alignable (pure noise trains to a diffuse alignment that no greedy search can follow): label u is
    # announced on frame 5u+2 by a fixed random code vector of its id
    codes = torch.randn(72, 80, generator=torch.Generator().manual_seed(8)).cuda()
    audio = 0.1 * batch[0]
    for b in range(targets.size(0)):
        for u in range(targets.size(1)):
            audio[b, 5 * u + 2] = 2.0 * codes[int(targets[b, u])]
    batch[0] = audio.contiguous()
    cfg = model.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    for step in range(400):
        opt.zero_grad()
        loss = model.training_step(tuple(batch), step)["loss"]
        loss.backward()
        opt.step()
        sched.step()
    assert loss.item() < 0.05, loss.item()
    out = model.validation_step(tuple(batch), 0)
    assert model.jointnet.training  # mode restored
    assert out["loss"].item() < 0.05
    ep = model.validation_epoch_end([out, out])  # model.py:81-108: mean loss + error rate over the epoch's steps
    assert abs(ep["val_loss"].item() - out["loss"].item()) < 1e-6 and 0.0 <= ep["val_ter"].item() <= 0.5
    from oracle.rnnt_oracle import OracleJointNet
    labels = [l.tolist() for l in out["label_tokens"]]
    # a greedy search need not follow the most probable alignment even at P(y|x) > 0.95 (emission mass may be spread
    # below 0.5 per frame), so the read-back check is: an ordered subsequence of the labels, at least half of them
    for pred, label in zip(out["pred_tokens"], labels):
        it = iter(label)
        assert all(tok in it for tok in pred.tolist()), (pred.tolist(), label)
        assert len(pred) >= len(label) // 2
    ora = OracleJointNet(tn, pn, 72).eval()
    ora.load_state_dict({k[len("jointnet."):]: v.cpu() for k, v in model.state_dict().items()})
    want, margin = ora.recognize_greedy(batch[0].cpu(), batch[1], 0, 3, return_margin=True)
    if margin >= 1e-4:
        assert [p.tolist() for p in out["pred_tokens"]] == want


def test_synthetic_pipeline_example_learns():
    """examples/pipeline_synthetic.py: front-end -> collate -> training steps -> validation (greedy search) on the GPU."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "pipeline_synthetic.py")
    spec = importlib.util.spec_from_file_location("pipeline_synthetic", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    first, last, ep = mod.main(["--steps", "60", "--batch", "4", "--seconds", "1.2"])
    assert last < 0.5 * first and 0.0 <= ep["val_ter"].item() <= 1.5


@pytest.mark.parametrize("cell,layers", [("lstm", 2), ("gru", 1), ("rnn", 2)])
def test_prednet_step_branch_and_1d_joint_match_oracle(cell, layers):
    """The step-wise surface the reference's search loops use (decoder.py:121-123 with carried state, transducer.py:64-69 on
    1-D vectors) against torch.nn stepping on the CPU: a (B,3) token block from None, then one more column from the state."""
    from oracle.rnnt_oracle import OracleJointNet
    tn = dict(input_size=12, hidden_size=16, output_size=8, num_layers=1, dropout=0.0, bidirectional=True)
    pn = dict(embedding_size=10, pad_token_id=0, hidden_size=32, output_size=8, num_layers=layers, rnn_type=cell, dropout=0.0)
    torch.manual_seed(11)
    ora = OracleJointNet(tn, pn, 10).eval()
    net = _jointnet(tn, pn, 10)
    net.load_state_dict(ora.state_dict())
    net = net.cuda().eval()
    toks = torch.tensor([[0, 3, 7], [0, 9, 1], [0, 2, 2], [0, 5, 4]])
    nxt = torch.tensor([[6], [8], [1], [3]])
    with torch.no_grad():
        y_ref, st_ref = ora.decoder.rnn(ora.decoder.embedding(toks), None)
        y2_ref, st2_ref = ora.decoder.rnn(ora.decoder.embedding(nxt), st_ref)
        want1, want2 = ora.decoder.out_proj(y_ref), ora.decoder.out_proj(y2_ref)
        got1, st = net.decoder(toks.cuda(), None, None)
        got2, st2 = net.decoder(nxt.cuda(), prev_hidden_state=st)
        assert (got1.cpu() - want1).abs().max() < 2e-5 and (got2.cpu() - want2).abs().max() < 2e-5
        flat = lambda s: torch.cat([t.reshape(-1) for t in (s if isinstance(s, tuple) else (s,))])  # noqa: E731
        assert (flat(tuple(t.cpu() for t in st2) if isinstance(st2, tuple) else st2.cpu()) - flat(st2_ref)).abs().max() < 2e-5
        enc = torch.randn(8)
        z = net.joint(enc.cuda(), got2[1, 0])
        z_ref = ora.fc(torch.nn.functional.gelu(torch.cat((enc, want2[1, 0])), approximate="tanh"))
        assert z.shape == (10,) and (z.cpu() - z_ref).abs().max() < 2e-5
    with pytest.raises(RuntimeError):
        net.decoder(nxt.cuda(), prev_hidden_state=st)  # inference-only: needs no_grad
