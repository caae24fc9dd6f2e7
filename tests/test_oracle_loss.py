"""Pins for the CPU restatement of the RNN-T loss (oracle/rnnt_loss_ref.c) — SURVEY.md §8(c) G3-G5.

The reference has no test at this boundary and its loss packages are not installable offline, so these
pins are: upstream known-answer vector, brute force, autograd through an independent DP, invariants.
"""
import numpy as np
import pytest
import torch

from oracle.rnnt_oracle import rnnt_loss_c, rnnt_nll_bruteforce, rnnt_nll_torch

G3_LOGITS = np.array([[[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1], [0.1, 0.1, 0.2, 0.8, 0.1]],
                      [[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.2, 0.1, 0.1], [0.7, 0.1, 0.2, 0.1, 0.1]]])[None]
G3_GRAD = np.array([[[-0.13116688, -0.3999269, 0.17703125, 0.17703125, 0.17703125],
                     [-0.18572757, 0.12247056, -0.18168412, 0.12247056, 0.12247056],
                     [-0.32091254, 0.06269141, 0.06928472, 0.12624499, 0.06269141]],
                    [[0.05456069, -0.21824276, 0.05456069, 0.05456069, 0.05456069],
                     [0.12073959, 0.12073959, -0.48295835, 0.12073959, 0.12073959],
                     [-0.6925882, 0.16871116, 0.18645467, 0.16871116, 0.16871116]]])[None]


@pytest.mark.parametrize("dtype,tol", [(np.float64, 2e-7), (np.float32, 5e-7)])
def test_g3_known_answer(dtype, tol):
    nll, grad = rnnt_loss_c(G3_LOGITS.astype(dtype), np.array([[1, 2]]), [2], [2], blank=0)
    assert abs(float(nll[0]) - 4.495666) < 2e-6
    np.testing.assert_allclose(grad, G3_GRAD, atol=tol)


# Second upstream known-answer vector [upstream, recalled]: the B=2, T=4, U=2, V=3 case of warp-transducer's / torchaudio's RNN-T loss tests
# (labels [[1,2],[1,1]] — a repeated label —, blank 0, expected costs 4.2806528590890736 and 3.9384369822503591).  The 72 inputs are
# written from memory of the upstream test; the expected costs are an 8-digit checksum of them: a single wrong input digit moves a
# cost by ~1e-6 or more, so the vector validates itself together with the restatement.
KAT2_ACTS = np.array([0.065357, 0.787530, 0.081592, 0.529716, 0.750675, 0.754135, 0.609764, 0.868140, 0.622532, 0.668522, 0.858039, 0.164539,
                      0.989780, 0.944298, 0.603168, 0.946783, 0.666203, 0.286882, 0.094184, 0.366674, 0.736168, 0.166680, 0.714154, 0.399400,
                      0.535982, 0.291821, 0.612642, 0.324241, 0.800764, 0.524106, 0.779195, 0.183314, 0.113745, 0.240222, 0.339470, 0.134160,
                      0.505562, 0.051597, 0.640290, 0.430733, 0.829473, 0.177467, 0.320700, 0.042883, 0.302803, 0.675178, 0.569537, 0.558474,
                      0.083132, 0.060165, 0.107958, 0.748615, 0.943918, 0.486356, 0.418199, 0.652408, 0.024243, 0.134582, 0.366342, 0.295830,
                      0.923670, 0.689929, 0.741898, 0.250005, 0.603430, 0.987289, 0.592606, 0.884672, 0.543450, 0.660770, 0.377128, 0.358021]
                     ).reshape(2, 4, 3, 3)
KAT2_LABELS, KAT2_COSTS = np.array([[1, 2], [1, 1]]), np.array([4.2806528590890736, 3.9384369822503591])


@pytest.mark.parametrize("dtype,tol", [(np.float64, 5e-7), (np.float32, 2e-6)])
def test_second_upstream_known_answer_vector_batch_of_two(dtype, tol):
    nll, grad = rnnt_loss_c(KAT2_ACTS.astype(dtype), KAT2_LABELS, [4, 4], [2, 2], blank=0)
    np.testing.assert_allclose(nll, KAT2_COSTS, atol=tol)
    np.testing.assert_allclose(grad.sum(-1), 0, atol=1e-6)   # softmax-fused gradients sum to zero over the vocabulary
    for b in range(2):   # and the restatement agrees with the path enumeration on it
        assert abs(float(nll[b]) - rnnt_nll_bruteforce(KAT2_ACTS[b].astype(np.float64), list(KAT2_LABELS[b]), 0)) < (1e-9 if dtype == np.float64 else 2e-6)


@pytest.mark.parametrize("T,U,V,seed", [(1, 0, 3, 0), (1, 3, 4, 1), (4, 0, 5, 2), (2, 2, 5, 3), (5, 4, 6, 4), (3, 3, 2, 5)])
def test_g4_bruteforce(T, U, V, seed):
    rng = np.random.default_rng(seed)
    z = rng.normal(size=(1, T, U + 1, V)) * 2.0
    y = rng.integers(1, V, size=(1, max(U, 1)))[:, :U].reshape(1, U)
    blank = 0
    nll, _ = rnnt_loss_c(z, y, [T], [U], blank)
    assert abs(nll[0] - rnnt_nll_bruteforce(z[0], list(y[0]), blank)) < 1e-10


def test_autograd_through_dp_ragged_and_nonzero_blank():
    rng = np.random.default_rng(7)
    B, T, U, V, blank = 3, 6, 4, 7, 2
    z = rng.normal(size=(B, T, U + 1, V))
    y = rng.integers(0, V - 1, size=(B, U))
    y[y >= blank] += 1  # labels never equal blank
    t_lens, u_lens = [6, 4, 1], [4, 0, 3]
    nll, grad = rnnt_loss_c(z, y, t_lens, u_lens, blank)
    zt = torch.tensor(z, requires_grad=True)
    ref = rnnt_nll_torch(zt, y.tolist(), t_lens, u_lens, blank)
    ref.sum().backward()
    np.testing.assert_allclose(nll, ref.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(grad, zt.grad.numpy(), atol=1e-12)
    # G5 invariants: zero outside the valid lattice, softmax-fused grads sum to zero over V
    for b in range(B):
        assert np.all(grad[b, t_lens[b]:] == 0) and np.all(grad[b, :, u_lens[b] + 1:] == 0)
    np.testing.assert_allclose(grad.sum(-1), 0, atol=1e-12)


def test_f32_tracks_f64_on_long_lattice():
    rng = np.random.default_rng(3)
    z = rng.normal(size=(2, 300, 21, 72))
    y = rng.integers(1, 72, size=(2, 20))
    n64, g64 = rnnt_loss_c(z, y, [300, 250], [20, 11])
    n32, g32 = rnnt_loss_c(z.astype(np.float32), y, [300, 250], [20, 11])
    assert np.max(np.abs(n32 - n64) / n64) < 2e-6
    # alpha/beta reach |1e3| here, so one fp32 ulp in log space is ~6e-5: grads agree to ~1e-3 only
    assert np.max(np.abs(g32 - g64)) < 2e-3


def test_bad_lengths_rejected():
    z = np.zeros((1, 3, 2, 4))
    with pytest.raises(ValueError):
        rnnt_loss_c(z, np.array([[1]]), [4], [1])
    with pytest.raises(ValueError):
        rnnt_loss_c(z, np.array([[1]]), [3], [2])


def test_cpu_build_of_the_c_abi_entry_matches_the_restatement():
    """oracle/rnnt_loss_ref.c also exports `rnnt_hip_loss_from_logits_fwd_bwd` with the argument list of include/rnnt_hip.h (host
    pointers): same numbers as rnnt_loss_c on the G3 known-answer vector, gscale applied to the gradient."""
    import ctypes
    from oracle import build_oracle
    lib = ctypes.CDLL(build_oracle.build())
    z = np.ascontiguousarray(G3_LOGITS[None].astype(np.float32)) if G3_LOGITS.ndim == 3 else np.ascontiguousarray(G3_LOGITS.astype(np.float32))
    B, T, U1, V = z.shape
    y = np.array([[1, 2]], np.int32)
    t_lens, u_lens = np.array([T], np.int32), np.array([U1 - 1], np.int32)
    nll, grad = np.empty(B, np.float32), np.empty_like(z)
    p = ctypes.c_void_p
    rc = lib.rnnt_hip_loss_from_logits_fwd_bwd(p(z.ctypes.data), p(y.ctypes.data), p(t_lens.ctypes.data), p(u_lens.ctypes.data), B, T, U1, V, 0,
                                               ctypes.c_float(2.0), p(nll.ctypes.data), p(grad.ctypes.data), None, ctypes.c_size_t(0), None)
    assert rc == 0 and abs(nll[0] - 4.495666) < 5e-6
    np.testing.assert_allclose(grad, 2.0 * np.asarray(G3_GRAD, np.float32).reshape(grad.shape), atol=2e-6)
