"""End-to-end parity of the drop-in modules (RNNTransducer / JointNet on HIP) against
  * fixtures generated from the REFERENCE's own networks (tests/golden/g*.npz, float64), and
  * the CPU oracle on seeded synthetic batches,
plus the north_star memory property (no (B,T,U+1,V) tensor in the fused training step)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from tests.test_oracle_networks import CONFIGS

pytestmark = pytest.mark.gpu
ARGS = Namespace(learning_rate=1e-3, weight_decay=1e-4, warmup_ratio=0.2, final_div_factor=1e4, total_steps=100,
                 move_metrics_to_cpu=False)


def _build(tag):
    from rnntransducer_amd import RNNTransducer
    tn, pn, V = CONFIGS[tag]
    return RNNTransducer(dict(pn), dict(tn), dict(num_classes=V), ARGS)


def _batch_from_fixture(g, dev):
    t_list, u_list = g["t_lens"].tolist(), g["u_lens"].tolist()
    return (torch.from_numpy(g["audios"]).float().to(dev), t_list, torch.tensor(t_list, dtype=torch.int32, device=dev),
            torch.from_numpy(g["texts"]).to(dev), [u + 1 for u in u_list], torch.from_numpy(g["targets"]).to(dev),
            torch.tensor(u_list, dtype=torch.int32, device=dev))


@pytest.mark.parametrize("tag", list(CONFIGS))
def test_reference_fixture_parity(golden_dir, tag):
    g = dict(np.load(os.path.join(golden_dir, tag + ".npz")))
    model = _build(tag)
    model.load_state_dict({"jointnet." + k[6:]: torch.from_numpy(v).float() for k, v in g.items() if k.startswith("param/")})
    model = model.cuda().train()
    batch = _batch_from_fixture(g, "cuda")
    # forward(): full logits, as model.py:47-50
    logits = model(batch[0], batch[1], batch[3], batch[4])
    valid = np.zeros(g["logits"].shape[:3], bool)
    for b, (t, u) in enumerate(zip(g["t_lens"], g["u_lens"])):
        valid[b, :t, :u + 1] = True
    assert np.abs(logits.detach().cpu().numpy() - g["logits"]).max() < 5e-5
    enc = model.jointnet.encoder(batch[0], batch[1])
    dec, _ = model.jointnet.decoder(batch[3], batch[4])
    assert np.abs(enc.detach().cpu().numpy() - g["enc"]).max() < 2e-5
    assert np.abs(dec.detach().cpu().numpy() - g["dec"]).max() < 2e-5
    # fused training_step: loss + every parameter gradient
    out = model.training_step(batch, 0)
    out["loss"].backward()
    assert abs(out["loss"].item() - float(g["loss"])) / float(g["loss"]) < 1e-5   # north_star: <= 1e-4 relative
    for name, p in model.jointnet.named_parameters():
        ref = g["grad/" + name]
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err < 2e-4 * max(np.abs(ref).max(), 1e-2), f"{name}: {err} vs scale {np.abs(ref).max()}"
    # the unfused path the reference's training_step takes (forward() then RNNTLoss) gives the same loss
    loss2 = model.rnnt_loss(logits, batch[5], batch[2], batch[6])
    assert abs(loss2.item() - out["loss"].item()) < 1e-5 * abs(loss2.item())
    assert loss2.dim() == 0


def test_config1_synthetic_vs_oracle_train_step():
    """BASELINE config 1 on a seeded synthetic ragged batch: loss and gradients vs the float64 CPU oracle."""
    from oracle.rnnt_oracle import OracleJointNet, make_batch, training_loss
    tn, pn, V = CONFIGS["g1_cfg1"]
    torch.manual_seed(0)
    model = _build("g1_cfg1")
    oracle = OracleJointNet(dict(tn), dict(pn), V).double()
    oracle.load_state_dict({k[len("jointnet."):]: v.double() for k, v in model.state_dict().items()})
    batch = make_batch(2, 100, 20, V, ragged=True, seed=99)
    ref_loss = training_loss(oracle, (batch[0].double(),) + batch[1:])
    ref_loss.backward()
    model = model.cuda()
    dev_batch = tuple(x.cuda() if isinstance(x, torch.Tensor) else x for x in batch)
    out = model.training_step(dev_batch, 0)
    out["loss"].backward()
    assert abs(out["loss"].item() - ref_loss.item()) / ref_loss.item() < 1e-5
    for (name, p), (_, q) in zip(model.jointnet.named_parameters(), oracle.named_parameters()):
        scale = max(q.grad.abs().max().item(), 1e-2)
        assert (p.grad.double().cpu() - q.grad).abs().max().item() < 2e-4 * scale, name


def test_ragged_batch_in_collate_order_takes_the_valid_frame_plan_and_matches_the_oracle(monkeypatch):
    """A ragged batch in COLLATE order (not sorted) through `training_step`-style calls with the host list of lengths: the module
    sorts the rows by length (networks/encoder.py:94-96), the recurrences / big products skip the padding (ops.RaggedPlan), and the
    per-utterance NLL comes back in the caller's order.  Every utterance's NLL and every parameter gradient against the float64
    oracle; and the same numbers as the dense path (no host list).  RNNT_GEMM_FORCE_HP puts this small shape on the half-pair
    products, where the valid-frame table is honoured."""
    from argparse import Namespace
    from oracle.rnnt_oracle import OracleJointNet, make_batch, training_loss
    from rnntransducer_amd import RNNTransducer, _lib
    monkeypatch.setenv("RNNT_GEMM_FORCE_HP", "1")
    V, B, T, U = 40, 6, 200, 12
    tn = dict(input_size=80, hidden_size=128, output_size=64, num_layers=2, rnn_type="lstm", dropout=0.0, bidirectional=True)
    pn = dict(embedding_size=V, hidden_size=64, output_size=64, num_layers=1, rnn_type="lstm", dropout=0.0)
    import os
    if not any(os.environ.get(k) for k in ("RNNT_LSTM_NO_V5", "RNNT_GEMM_NO_HP", "RNNT_LSTM_V1", "RNNT_LSTM_V2", "RNNT_LSTM_EXACT_MATH")):
        assert _lib.lib().rnnt_hip_lstm_takes_row_idx(T, B, 80, 128, 2, 0) == 1
    args = Namespace(learning_rate=1e-3, weight_decay=1e-4, warmup_ratio=0.2, final_div_factor=1e4, total_steps=10, move_metrics_to_cpu=False)
    torch.manual_seed(5)
    model = RNNTransducer(dict(pn), dict(tn), dict(num_classes=V), args)
    oracle = OracleJointNet(dict(tn), dict(pn, pad_token_id=0), V).double()
    oracle.load_state_dict({k[len("jointnet."):]: v.double() for k, v in model.state_dict().items()})
    batch = list(make_batch(B, T, U, V, ragged=True, seed=21))
    lens = [T, 117, 180, 101, 199, 150]                     # collate order: unsorted, every row different
    for b in range(B):
        batch[0][b, lens[b]:] = 0
    batch[1], batch[2] = lens, torch.tensor(lens, dtype=torch.int32)
    gw = torch.tensor([1.0, 0.5, 2.0, 1.5, 0.25, 1.0], dtype=torch.float64)   # a different weight per utterance: order matters
    from oracle.rnnt_oracle import _RNNTLossFn
    logits = oracle(batch[0].double(), batch[1], batch[3], batch[4])
    ref_nll = _RNNTLossFn.apply(logits, batch[5], batch[2], batch[6], 0)
    (ref_nll * gw).sum().backward()
    model = model.cuda().train()
    dev = tuple(x.cuda() if isinstance(x, torch.Tensor) else x for x in batch)
    res = {}
    for name, host_list in (("plan", dev[1]), ("dense", None)):
        for p in model.parameters():
            p.grad = None
        nll = model.jointnet.loss(dev[0], dev[2], dev[3], dev[5], dev[6], 0, reduction="none", audio_lengths=host_list)
        (nll * gw.float().cuda()).sum().backward()
        torch.cuda.synchronize()
        res[name] = (nll.detach().double().cpu(), {k: p.grad.double().cpu() for k, p in model.jointnet.named_parameters()})
        assert torch.allclose(res[name][0], ref_nll.detach(), rtol=1e-5, atol=0), (name, res[name][0], ref_nll)
        for k, q in oracle.named_parameters():
            scale = max(q.grad.abs().max().item(), 1e-2)
            assert (res[name][1][k] - q.grad).abs().max().item() < 2e-4 * scale, (name, k)
    assert torch.allclose(res["plan"][0], res["dense"][0], rtol=2e-6, atol=0)


def test_fused_step_never_materialises_btuv():
    """config-3-shaped (scaled to fit a quick test): peak memory of the fused step stays far below one (B,T,U+1,V)
    tensor plus one (B,T,U+1,2*O) tensor, which the reference allocates at networks/transducer.py:61-69."""
    from oracle.rnnt_oracle import make_batch
    from rnntransducer_amd import RNNTransducer
    B, T, U, V, O, H = 4, 800, 120, 72, 512, 64
    model = RNNTransducer(dict(embedding_size=V, hidden_size=H, output_size=O, num_layers=1),
                          dict(input_size=80, hidden_size=H, output_size=O, num_layers=1, bidirectional=True),
                          dict(num_classes=V), ARGS).cuda()
    batch = tuple(x.cuda() if isinstance(x, torch.Tensor) else x for x in make_batch(B, T, U, V, seed=5))
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    model.training_step(batch, 0)["loss"].backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    concat_bytes = B * T * (U + 1) * 2 * O * 4
    logits_bytes = B * T * (U + 1) * V * 4
    assert peak < 0.5 * logits_bytes + 0.05 * concat_bytes, (peak, logits_bytes, concat_bytes)


def test_optimizer_contract_and_one_adamw_step_changes_loss():
    from oracle.rnnt_oracle import make_batch
    model = _build("g2_stack").cuda()
    conf = model.configure_optimizers()
    assert conf["lr_scheduler"]["interval"] == "step" and isinstance(conf["optimizer"], torch.optim.AdamW)
    batch = tuple(x.cuda() if isinstance(x, torch.Tensor) else x for x in make_batch(3, 12, 4, 10, n_mels=12, ragged=True, seed=1))
    losses = []
    for _ in range(8):
        conf["optimizer"].zero_grad()
        loss = model.training_step(batch, 0)["loss"]
        loss.backward()
        conf["optimizer"].step()
        conf["lr_scheduler"]["scheduler"].step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]


def test_flat_adamw_matches_torch_adamw_step_for_step():
    """FlatAdamW (one fused HIP kernel over flat buffers) vs torch.optim.AdamW on identical parameters/gradients,
    driven by the same OneCycleLR schedule, 5 steps."""
    from rnntransducer_amd.optim import FlatAdamW
    torch.manual_seed(0)
    shapes = [(7, 5), (13,), (4, 3, 2), (1,), (130, 70)]
    ref_p = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    hip_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    ref = torch.optim.AdamW(ref_p, lr=1e-2, weight_decay=1e-2)
    hip = FlatAdamW(hip_p, lr=1e-2, weight_decay=1e-2)
    assert isinstance(hip, torch.optim.AdamW)
    s_ref = torch.optim.lr_scheduler.OneCycleLR(ref, max_lr=1e-2, total_steps=20, pct_start=0.2)
    s_hip = torch.optim.lr_scheduler.OneCycleLR(hip, max_lr=1e-2, total_steps=20, pct_start=0.2)
    for it in range(5):
        hip.zero_grad()
        for a, b in zip(ref_p, hip_p):
            gr = torch.randn_like(a) * (it + 1)
            a.grad = gr.clone()
            b.grad += gr.cuda()          # accumulates into the flat view, like autograd does
        ref.step(); hip.step(); s_ref.step(); s_hip.step()
    for a, b in zip(ref_p, hip_p):
        assert (a.detach() - b.detach().cpu()).abs().max().item() < 2e-6
    sd = hip.state_dict()
    assert len(sd["state"]) == len(shapes) and "exp_avg" in sd["state"][0]


def test_training_is_bitwise_reproducible_run_to_run():
    """Same seed, same batch, two fresh runs of a few full training steps (dropout on): every loss and every parameter
    bitwise equal.  No float atomics anywhere on the path: split-K slabs, bias gradients, dC tiles and the lattice
    gradient's LDS merge are all summed in a fixed order; the dropout mask is a counter-based hash."""
    from rnntransducer_amd.data import synthetic_batch

    def run():
        torch.manual_seed(0)
        model = _build("g1_cfg1").cuda().train()
        batch = synthetic_batch(4, 120, 14, 72, ragged=True, seed=5, device="cuda")
        conf = model.configure_optimizers()
        opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
        losses = []
        for _ in range(4):
            opt.zero_grad()
            loss = model.training_step(batch, 0)["loss"]
            loss.backward()
            opt.step()
            sched.step()
            losses.append(loss.item())
        return losses, [p.detach().clone() for p in model.parameters()]

    la, pa = run()
    lb, pb = run()
    assert la == lb
    assert all(torch.equal(x, y) for x, y in zip(pa, pb))
