"""BASELINE.json configs at their FULL lattice / layer sizes on the HIP path.

  * the FUSED joint + RNN-T loss entry (`JointLossFn`: lse_sep / alphabeta / grad_sep kernels — what `training_step`
    runs, networks/transducer.py:58-69 + model.py:56-57) at T=1000/U=40/V=72, T=2000/U=120/V=72, T=1500/U=80/V=2048
    on two utterances, against the float64 C oracle (oracle/rnnt_loss_ref.c) + float64 torch autograd;
  * a full-model step at config-2 layer sizes (4x512 bi-LSTM, 1x512 prediction net, T=1000, U=40) against
    `OracleJointNet(...).double()`: loss and every parameter gradient;
  * config 3 at full size: the joint + loss segment never allocates anything near a (B,T,U+1,V) tensor.

Tolerances (fp32 path vs float64 oracle): NLL 1e-5 relative (north_star: 1e-4); input/weight gradients of the fused
entry 5e-5 of max(1, |ref|max); parameter gradients of the whole model 2e-4 of the tensor's max.
"""
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle.rnnt_oracle import OracleJointNet, make_batch, rnnt_loss_c, training_loss

pytestmark = pytest.mark.gpu
ARGS = Namespace(learning_rate=1e-3, weight_decay=1e-4, warmup_ratio=0.2, final_div_factor=1e4, total_steps=100,
                 move_metrics_to_cpu=False)
SHAPES = {  # name: (T, U, V, O) of BASELINE.json configs[1], [2], [4]
    "c2": (1000, 40, 72, 512),
    "c3": (2000, 120, 72, 512),
    "c5": (1500, 80, 2048, 640),
}


def _gelu64(x):
    return torch.nn.functional.gelu(x, approximate="tanh")


@pytest.mark.parametrize("cfg", list(SHAPES))
def test_fused_joint_loss_at_baseline_lattice_shapes(cfg):
    """Two utterances (one full length, one ragged) at the config's T, U, V and joint width O."""
    from rnntransducer_amd.ops import JointLossFn
    T, U, V, O = SHAPES[cfg]
    B = 2
    g = torch.Generator().manual_seed(1000 + T + U + V)
    enc = torch.randn(B, T, O, generator=g, dtype=torch.float64)
    dec = torch.randn(B, U + 1, O, generator=g, dtype=torch.float64)
    W = torch.randn(V, 2 * O, generator=g, dtype=torch.float64) * (1.0 / O ** 0.5)
    bias = torch.randn(V, generator=g, dtype=torch.float64) * 0.1
    y = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32)
    t_lens, u_lens = [T, (2 * T) // 3 + 1], [U, (2 * U) // 3]
    # float64 oracle.  c2: the MATERIALISING joint of networks/transducer.py:58-69; c3 / c5: its separable equal
    # (A[b,t] + C[b,u] + bias — equality pinned at small sizes by test_gpu_loss.py and tests/test_oracle_networks.py),
    # because the (B,T,U+1,2*O) float64 concat of c3/c5 is 8-10 GB per copy
    e, d, w, bb = (x.clone().requires_grad_(True) for x in (enc, dec, W, bias))
    if cfg == "c2":
        cat = torch.cat((e[:, :, None, :].expand(-1, -1, U + 1, -1), d[:, None, :, :].expand(-1, T, -1, -1)), -1)
        logits = _gelu64(cat) @ w.T + bb
    else:
        A = _gelu64(e) @ w[:, :O].T
        Cm = _gelu64(d) @ w[:, O:].T
        logits = A[:, :, None, :] + Cm[:, None, :, :] + bb
    ref_nll, dlog = rnnt_loss_c(logits.detach().numpy(), y.numpy(), t_lens, u_lens, 0)
    gw = torch.tensor([1.0, 0.5], dtype=torch.float64)
    logits.backward(torch.from_numpy(dlog) * gw.view(-1, 1, 1, 1))
    del dlog, logits
    dev = "cuda"
    te = enc.float().transpose(0, 1).contiguous().to(dev).requires_grad_(True)
    td = dec.float().transpose(0, 1).contiguous().to(dev).requires_grad_(True)
    tw = W.float().to(dev).requires_grad_(True)
    tb = bias.float().to(dev).requires_grad_(True)
    nll = JointLossFn.apply(te, td, tw, tb, y.to(dev), torch.tensor(t_lens, dtype=torch.int32, device=dev),
                            torch.tensor(u_lens, dtype=torch.int32, device=dev), 0)
    (nll * gw.float().to(dev)).sum().backward()
    np.testing.assert_allclose(nll.detach().cpu().numpy(), ref_nll, rtol=1e-5)
    for name, got, ref in (("d_enc", te.grad.transpose(0, 1), e.grad), ("d_dec", td.grad.transpose(0, 1), d.grad),
                           ("d_fc.weight", tw.grad, w.grad), ("d_fc.bias", tb.grad, bb.grad)):
        err = (got.double().cpu() - ref).abs().max().item()
        assert err < 5e-5 * max(1.0, ref.abs().max().item()), f"{cfg} {name}: {err} (ref max {ref.abs().max().item()})"
    # padded frames / label positions get no gradient (warp-transducer convention)
    assert te.grad[t_lens[1]:, 1].abs().max().item() == 0.0
    assert td.grad[u_lens[1] + 1:, 1].abs().max().item() == 0.0


def test_full_model_config2_dims_step_vs_float64_oracle():
    """BASELINE configs[1] layer sizes (enc 4x512 bi-LSTM, pred 1x512, O=512, V=72), T=1000, U=40, two utterances
    (second one ragged), dropout off on both sides: loss + every parameter gradient against the float64 oracle."""
    from rnntransducer_amd import RNNTransducer
    V = 72
    tn = dict(input_size=80, hidden_size=512, output_size=512, num_layers=4, rnn_type="lstm", dropout=0.0, bidirectional=True)
    pn = dict(embedding_size=V, hidden_size=512, output_size=512, num_layers=1, rnn_type="lstm", dropout=0.0)
    torch.manual_seed(0)
    model = RNNTransducer(dict(pn), dict(tn), dict(num_classes=V), ARGS)
    oracle = OracleJointNet(dict(tn), dict(pn, pad_token_id=0), V).double()
    oracle.load_state_dict({k[len("jointnet."):]: v.double() for k, v in model.state_dict().items()})
    batch = make_batch(2, 1000, 40, V, ragged=True, seed=7)
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    ref = training_loss(oracle, (batch[0].double(),) + batch[1:])
    ref.backward()
    model = model.cuda().train()
    dev_batch = tuple(x.cuda() if isinstance(x, torch.Tensor) else x for x in batch)
    out = model.training_step(dev_batch, 0)
    out["loss"].backward()
    assert abs(out["loss"].item() - ref.item()) / ref.item() < 1e-5
    worst = ("", 0.0)
    for (name, p), (_, q) in zip(model.jointnet.named_parameters(), oracle.named_parameters()):
        scale = max(q.grad.abs().max().item(), 1e-3)
        rel = (p.grad.double().cpu() - q.grad).abs().max().item() / scale
        if rel > worst[1]:
            worst = (name, rel)
        assert rel < 2e-4, f"{name}: {rel:.3e} of max |grad| {scale:.3e}"
    print(f"config-2 dims: loss rel {abs(out['loss'].item() - ref.item()) / ref.item():.2e}; worst gradient {worst[0]} {worst[1]:.2e} of max")


def test_config3_full_size_joint_loss_never_materialises_btuv():
    """BASELINE configs[2] (B=8, T=2000, U=120, V=72, O=512) at FULL size: allocator peak of the fused joint + loss +
    its backward stays below 30 % of ONE (B,T,U+1,V) logits tensor (the reference builds that plus three
    (B,T,U+1,2*O) tensors, networks/transducer.py:61-69)."""
    from rnntransducer_amd.ops import JointLossFn
    B, T, U, V, O = 8, 2000, 120, 72, 512
    g = torch.Generator(device="cuda").manual_seed(3)
    enc = torch.randn(T, B, O, device="cuda", generator=g).requires_grad_(True)
    dec = torch.randn(U + 1, B, O, device="cuda", generator=g).requires_grad_(True)
    W = (torch.randn(V, 2 * O, device="cuda", generator=g) * 0.05).requires_grad_(True)
    bias = torch.zeros(V, device="cuda", requires_grad=True)
    y = torch.randint(1, V, (B, U), device="cuda", generator=g, dtype=torch.int32)
    t_lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    u_lens = torch.full((B,), U, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    nll = JointLossFn.apply(enc, dec, W, bias, y, t_lens, u_lens, 0)
    nll.mean().backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    logits_bytes = B * T * (U + 1) * V * 4
    assert torch.isfinite(nll).all()
    assert peak < 0.3 * logits_bytes, (peak, logits_bytes)
