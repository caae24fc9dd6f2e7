"""HIP RNN-T lattice kernels vs the CPU oracle (oracle/rnnt_loss_ref.c, float64): dense-logits entry and fused
separable entry, through the C ABI.  Tolerances: NLL 1e-5 relative (north_star asks 1e-4), grads 2e-5 absolute
(each |grad| <= 1)."""
import numpy as np
import pytest
import torch

from oracle.rnnt_oracle import rnnt_loss_c

pytestmark = pytest.mark.gpu
NLL_RTOL, GRAD_ATOL = 1e-5, 2e-5


def _run_dense(z, y, t_lens, u_lens, blank=0):
    from rnntransducer_amd.ops import RnntLossFromLogitsFn
    dev = "cuda"
    zt = torch.tensor(z, dtype=torch.float32, device=dev, requires_grad=True)
    nll = RnntLossFromLogitsFn.apply(zt, torch.tensor(y, dtype=torch.int32, device=dev),
                                     torch.tensor(t_lens, dtype=torch.int32, device=dev),
                                     torch.tensor(u_lens, dtype=torch.int32, device=dev), blank)
    nll.sum().backward()
    return nll.detach().cpu().numpy(), zt.grad.cpu().numpy()


def test_g3_known_answer_on_gpu():
    from tests.test_oracle_loss import G3_GRAD, G3_LOGITS
    nll, grad = _run_dense(G3_LOGITS, np.array([[1, 2]]), [2], [2])
    assert abs(nll[0] - 4.495666) < 5e-6
    np.testing.assert_allclose(grad, G3_GRAD, atol=1e-6)


def test_second_upstream_known_answer_vector_on_gpu():
    """The B=2, T=4, U=2, V=3 known-answer vector of warp-transducer's / torchaudio's own loss tests (tests/test_oracle_loss.py: expected
    costs 4.28065286 / 3.93843698) through the HIP lattice kernels; gradients against the float64 restatement."""
    from tests.test_oracle_loss import KAT2_ACTS, KAT2_COSTS, KAT2_LABELS
    nll, grad = _run_dense(KAT2_ACTS, KAT2_LABELS, [4, 4], [2, 2])
    np.testing.assert_allclose(nll, KAT2_COSTS, atol=5e-6)
    _, ref_grad = rnnt_loss_c(KAT2_ACTS.astype(np.float64), KAT2_LABELS, [4, 4], [2, 2], 0)
    np.testing.assert_allclose(grad, ref_grad, atol=1e-6)


@pytest.mark.parametrize("B,T,U,V,blank,ragged", [(1, 1, 0, 3, 0, False), (2, 1, 3, 5, 0, False), (3, 7, 0, 4, 1, True),
                                                  (4, 50, 20, 72, 0, True), (2, 33, 70, 9, 3, True),
                                                  (2, 20, 150, 6, 0, True), (3, 40, 300, 5, 0, True),
                                                  (2, 12, 6, 2048, 5, True), (5, 257, 40, 72, 0, True)])
def test_dense_matches_oracle(B, T, U, V, blank, ragged):
    rng = np.random.default_rng(B * 1000 + T * 10 + U)
    z = (rng.normal(size=(B, T, U + 1, V)) * 1.5).astype(np.float32)
    y = rng.integers(0, V - 1, size=(B, U))
    y[y >= blank] += 1
    t_lens = [T] + list(rng.integers(1, T + 1, size=B - 1)) if ragged else [T] * B
    u_lens = [U] + list(rng.integers(0, U + 1, size=B - 1)) if ragged else [U] * B
    ref_nll, ref_grad = rnnt_loss_c(z.astype(np.float64), y, t_lens, u_lens, blank)
    nll, grad = _run_dense(z, y, t_lens, u_lens, blank)
    np.testing.assert_allclose(nll, ref_nll, rtol=NLL_RTOL)
    assert np.abs(grad - ref_grad).max() < GRAD_ATOL
    for b in range(B):  # zero outside the valid lattice (warp-transducer convention)
        assert np.all(grad[b, t_lens[b]:] == 0) and np.all(grad[b, :, u_lens[b] + 1:] == 0)


@pytest.mark.parametrize("B,T,U,V,Oe,Od,ragged", [(2, 9, 4, 10, 8, 8, True), (3, 70, 20, 72, 32, 16, True), (2, 40, 130, 12, 8, 12, True),
                                                  (2, 10, 5, 300, 16, 16, False)])
def test_fused_joint_loss_matches_oracle(B, T, U, V, Oe, Od, ragged):
    """enc/dec -> (A, C) GEMMs with fused GELU -> lattice -> dA/dC -> d_enc, d_dec, d_fc: against torch-CPU float64
    autograd through the MATERIALISING joint (networks/transducer.py:58-69) + the oracle's loss gradient."""
    from rnntransducer_amd.ops import JointLossFn
    g = torch.Generator().manual_seed(B + T + U + V)
    enc = torch.randn(B, T, Oe, generator=g, dtype=torch.float64)
    dec = torch.randn(B, U + 1, Od, generator=g, dtype=torch.float64)
    W = torch.randn(V, Oe + Od, generator=g, dtype=torch.float64) * 0.3
    bias = torch.randn(V, generator=g, dtype=torch.float64) * 0.1
    y = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32)
    t_lens = [T] + torch.randint(1, T + 1, (B - 1,), generator=g).tolist() if ragged else [T] * B
    u_lens = [U] + torch.randint(0, U + 1, (B - 1,), generator=g).tolist() if ragged else [U] * B
    # oracle
    e, d, w, bb = (x.clone().requires_grad_(True) for x in (enc, dec, W, bias))
    cat = torch.cat((e[:, :, None, :].expand(-1, -1, U + 1, -1), d[:, None, :, :].expand(-1, T, -1, -1)), -1)
    logits = torch.nn.functional.gelu(cat, approximate="tanh") @ w.T + bb
    ref_nll, dlog = rnnt_loss_c(logits.detach().numpy(), y.numpy(), t_lens, u_lens, 0)
    gw = torch.linspace(0.5, 1.5, B, dtype=torch.float64)  # non-uniform upstream gradient per utterance
    logits.backward(torch.from_numpy(dlog) * gw.view(-1, 1, 1, 1))
    # HIP (time-major inputs)
    dev = "cuda"
    te = enc.float().transpose(0, 1).contiguous().to(dev).requires_grad_(True)
    td = dec.float().transpose(0, 1).contiguous().to(dev).requires_grad_(True)
    tw = W.float().to(dev).requires_grad_(True)
    tb = bias.float().to(dev).requires_grad_(True)
    nll = JointLossFn.apply(te, td, tw, tb, y.to(dev), torch.tensor(t_lens, dtype=torch.int32, device=dev),
                            torch.tensor(u_lens, dtype=torch.int32, device=dev), 0)
    (nll * gw.float().to(dev)).sum().backward()
    np.testing.assert_allclose(nll.detach().cpu().numpy(), ref_nll, rtol=NLL_RTOL)
    tol = 5e-5
    for name, got, ref in (("d_enc", te.grad.transpose(0, 1), e.grad), ("d_dec", td.grad.transpose(0, 1), d.grad),
                           ("d_fc.weight", tw.grad, w.grad), ("d_fc.bias", tb.grad, bb.grad)):
        err = (got.double().cpu() - ref).abs().max().item()
        assert err < tol * max(1.0, ref.abs().max().item()), f"{name}: {err}"


def test_joint_logits_materialising_matches_separable_oracle():
    from rnntransducer_amd.ops import JointLogitsFn
    g = torch.Generator().manual_seed(3)
    B, T, U1, V, O = 2, 6, 4, 11, 8
    enc, dec = torch.randn(B, T, O, generator=g), torch.randn(B, U1, O, generator=g)
    W, bias = torch.randn(V, 2 * O, generator=g), torch.randn(V, generator=g)
    cat = torch.cat((enc[:, :, None, :].expand(-1, -1, U1, -1), dec[:, None, :, :].expand(-1, T, -1, -1)), -1).double()
    ref = torch.nn.functional.gelu(cat, approximate="tanh") @ W.double().T + bias.double()
    out = JointLogitsFn.apply(enc.transpose(0, 1).contiguous().cuda(), dec.transpose(0, 1).contiguous().cuda(), W.cuda(), bias.cuda())
    assert (out.double().cpu() - ref).abs().max().item() < 2e-5


def test_full_size_config2_lattice_properties():
    """BASELINE config 2 lattice (B=32,T=1000,U=40,V=72): size-independent properties + oracle on 2 utterances."""
    from rnntransducer_amd.ops import RnntLossFromLogitsFn
    B, T, U, V = 32, 1000, 40, 72
    g = torch.Generator(device="cuda").manual_seed(0)
    z = torch.randn(B, T, U + 1, V, device="cuda", generator=g)
    y = torch.randint(1, V, (B, U), device="cuda", generator=g, dtype=torch.int32)
    t_lens = torch.randint(T // 2, T + 1, (B,), device="cuda", generator=g, dtype=torch.int32)
    t_lens[0] = T
    u_lens = torch.clamp((t_lens.float() * U / T).round().int(), 1, U)
    z.requires_grad_(True)
    nll = RnntLossFromLogitsFn.apply(z, y, t_lens, u_lens, 0)
    nll.sum().backward()
    assert torch.isfinite(nll).all() and (nll > 0).all()
    assert z.grad.sum(-1).abs().max().item() < 1e-4          # softmax-fused grads sum to zero over V
    sl = slice(0, 2)
    ref_nll, ref_grad = rnnt_loss_c(z.detach()[sl].double().cpu().numpy(), y[sl].cpu().numpy(), t_lens[sl].cpu().numpy(),
                                    u_lens[sl].cpu().numpy(), 0)
    np.testing.assert_allclose(nll[sl].detach().cpu().numpy(), ref_nll, rtol=NLL_RTOL)
    assert np.abs(z.grad[sl].cpu().numpy() - ref_grad).max() < GRAD_ATOL


def test_loss_rejects_bad_dtypes_and_shapes():
    from rnntransducer_amd.loss import RNNTLoss
    z = torch.zeros(1, 2, 2, 3, device="cuda")
    t = torch.tensor([2], dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        RNNTLoss()(z, torch.zeros(1, 1, dtype=torch.int64, device="cuda"), t, t)
    with pytest.raises(ValueError):
        RNNTLoss()(z, torch.zeros(1, 5, dtype=torch.int32, device="cuda"), t, t)


@pytest.mark.parametrize("dtype,grad_tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.5e-2)])
def test_half_precision_logits_like_torchaudio_loss(dtype, grad_tol):
    """the precision-16 branch (model.py:28-31) hands half logits to the loss.  Storage converts, arithmetic
    stays fp32/fp64: NLL must equal the fp32 kernel run on the same (rounded) logits; grads are rounded to the dtype."""
    from rnntransducer_amd.loss import RNNTLoss
    rng = np.random.default_rng(11)
    B, T, U, V = 3, 40, 9, 72
    z = torch.tensor(rng.normal(size=(B, T, U + 1, V)), dtype=dtype)
    y = torch.tensor(rng.integers(1, V, size=(B, U)), dtype=torch.int32)
    t_lens, u_lens = [40, 33, 7], [9, 4, 0]
    ref_nll, ref_grad = rnnt_loss_c(z.double().numpy(), y.numpy(), t_lens, u_lens, 0)
    zg = z.cuda().requires_grad_(True)
    loss = RNNTLoss(0, "sum")(zg, y.cuda(), torch.tensor(t_lens, dtype=torch.int32, device="cuda"),
                              torch.tensor(u_lens, dtype=torch.int32, device="cuda"))
    loss.backward()
    assert abs(loss.item() - ref_nll.sum()) / ref_nll.sum() < 1e-5
    assert zg.grad.dtype == dtype
    assert np.abs(zg.grad.float().cpu().numpy() - ref_grad).max() < grad_tol
