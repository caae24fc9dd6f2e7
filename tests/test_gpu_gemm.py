"""rnnt_hip_gemm_f32 vs a float64 CPU product: every operand map the hot path uses, in the library's default arithmetic
(fp32 operands split exactly into three bf16 pieces, six bf16 MFMA products, fp32 accumulate) and, in
test_gemm_arithmetic_modes, in the exact-fp32-MFMA and first-order-split modes as well."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 2e-5  # fp32 fma chains over K <= 4096 against an fp64 reference, relative to sum |a||b|


def _ref_check(out, ref, scale):
    err = (out.double().cpu() - ref).abs().max().item()
    assert err <= RTOL * scale, f"max err {err} vs budget {RTOL * scale}"


def gelu64(x):
    return torch.nn.functional.gelu(x.double(), approximate="tanh")


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (130, 70, 33), (257, 129, 80), (512, 256, 1024), (1000, 72, 512), (64, 2048, 640),
                                   (37, 10, 16), (5, 3, 7), (2048, 1024, 96), (2100, 768, 40),
                                   # 256x256-tile kernel: unaligned K (scalar loads), exact tile, ragged edges
                                   (300, 260, 33), (256, 256, 16), (515, 402, 400)])
@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
def test_gemm_modes(M, N, K, mode):
    from rnntransducer_amd.ops import gemm
    g = torch.Generator().manual_seed(M * 31 + N * 7 + K)
    dev = "cuda"
    if mode == "nt":      # C = A (M,K) . W (N,K)^T + bias   (forward projections)
        A, W, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
        out = torch.full((M, N), float("nan"), device=dev)
        gemm(M, N, K, A.to(dev), W.to(dev), out, bias=bias.to(dev))
        ref = A.double() @ W.double().T + bias.double()
        scale = (A.abs().double() @ W.abs().double().T).max().item()
    elif mode == "nn":    # C = G (M,K) . W (K,N)            (dX = dG . W)
        A, W = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
        out = torch.full((M, N), float("nan"), device=dev)
        gemm(M, N, K, A.to(dev), W.to(dev), out, b_sn=1, b_sk=N)
        ref = A.double() @ W.double()
        scale = (A.abs().double() @ W.abs().double()).max().item()
    else:                 # C = G (K,M)^T . X (K,N)          (dW = dG^T . X)
        A, W = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
        out = torch.full((M, N), float("nan"), device=dev)
        gemm(M, N, K, A.to(dev), W.to(dev), out, a_mc=True, a_sk=M, b_sn=1, b_sk=N)
        ref = A.double().T @ W.double()
        scale = (A.abs().double().T @ W.abs().double()).max().item()
    _ref_check(out, ref, scale)


def test_gemm_row_maps_gelu_and_epilogues():
    from rnntransducer_amd._lib import GEMM_ACCUM, GEMM_GELU_A, GEMM_GELU_B, GEMM_MUL_DGELU
    from rnntransducer_amd.ops import gemm
    g = torch.Generator().manual_seed(5)
    dev = "cuda"
    T, B, K, N = 9, 5, 24, 40
    x_bm = torch.randn(B, T, K, generator=g)                      # batch-major source read as rows (t,b)
    W = torch.randn(N, K, generator=g)
    out = torch.zeros(T * B, N, device=dev)
    gemm(T * B, N, K, x_bm.to(dev), W.to(dev), out, a_div=B, a_so=K, a_si=T * K, flags=GEMM_GELU_A)
    ref = gelu64(x_bm.transpose(0, 1).reshape(T * B, K)) @ W.double().T
    _ref_check(out, ref, 40.0)
    # rows (t,b) scattered back into a batch-major destination + ACCUM
    dst = torch.ones(B, T, N, device=dev)
    gemm(T * B, N, K, x_bm.transpose(0, 1).contiguous().to(dev), W.to(dev), dst, c_div=B, c_so=N, c_si=T * N, flags=GEMM_ACCUM)
    ref2 = (x_bm.double() @ W.double().T) + 1.0
    _ref_check(dst, ref2, 40.0)
    # gathered rows (embedding-style) via a_rowidx
    table = torch.randn(11, K, generator=g)
    idx = torch.randint(0, 11, (T * B,), generator=g)
    out3 = torch.zeros(T * B, N, device=dev)
    gemm(T * B, N, K, table.to(dev), W.to(dev), out3, a_rowidx=idx.to(dev), a_si=K)
    _ref_check(out3, table[idx].double() @ W.double().T, 40.0)
    # GELU on B + sub-block destination (dW_e / dW_d halves of fc.weight's gradient)
    dA = torch.randn(T * B, N, generator=g)
    enc = torch.randn(T * B, K, generator=g)
    dW = torch.zeros(N, K + 8, device=dev)
    gemm(N, K, T * B, dA.to(dev), enc.to(dev), dW, a_mc=True, a_sk=N, b_sn=1, b_sk=K, c_off=8, c_div=1, c_so=K + 8, c_si=0,
         flags=GEMM_GELU_B)
    _ref_check(dW[:, 8:], dA.double().T @ gelu64(enc), 60.0)
    assert torch.all(dW[:, :8] == 0)
    # epilogue x gelu'(aux)
    d_enc = torch.zeros(T * B, K, device=dev)
    Wk = torch.randn(N, K, generator=g)
    gemm(T * B, K, N, dA.to(dev), Wk.to(dev), d_enc, b_sn=1, b_sk=K, aux=enc.to(dev), flags=GEMM_MUL_DGELU)
    e = enc.double().requires_grad_(True)
    gelu64(e).backward(dA.double() @ Wk.double())
    _ref_check(d_enc, e.grad, 60.0)


def test_gemm_rejects_bad_arguments():
    from rnntransducer_amd.ops import gemm
    a = torch.zeros(4, 4, device="cuda")
    with pytest.raises(ValueError):
        gemm(4, 4, 4, a, a, a, b_sn=2, b_sk=2)


@pytest.mark.parametrize("M,N,K", [(72, 512, 8000), (256, 96, 4100), (2048, 512, 3000), (130, 70, 129)])
def test_gemm_split_k_is_exact_sum_and_deterministic(M, N, K):
    """Weight-gradient shape (small output, deep K): split-K slabs + fixed-order reduce; twice -> bitwise equal."""
    from rnntransducer_amd._lib import GEMM_ACCUM
    from rnntransducer_amd.ops import gemm
    g = torch.Generator().manual_seed(K)
    A, W = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    ref = A.double().T @ W.double()
    scale = (A.abs().double().T @ W.abs().double()).max().item()
    outs = []
    for _ in range(2):
        out = torch.ones(M, N, device="cuda")
        gemm(M, N, K, A.cuda(), W.cuda(), out, a_mc=True, a_sk=M, b_sn=1, b_sk=N, flags=GEMM_ACCUM, split_k=True)
        outs.append(out)
    _ref_check(outs[0], ref + 1.0, scale)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N", [(1, 1), (1000, 72), (32000, 4096), (513, 130)])
def test_colsum(M, N):
    from rnntransducer_amd.ops import colsum
    x = torch.randn(M, N, generator=torch.Generator().manual_seed(M + N))
    out = colsum(x.cuda(), M, N)
    ref = x.double().sum(0)
    assert (out.double().cpu() - ref).abs().max().item() < 1e-5 * max(1.0, x.abs().double().sum(0).max().item())


@pytest.fixture
def gemm_mode_env():
    saved = os.environ.get("RNNT_GEMM_MODE")
    yield
    if saved is None:
        os.environ.pop("RNNT_GEMM_MODE", None)
    else:
        os.environ["RNNT_GEMM_MODE"] = saved


@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
def test_gemm_arithmetic_modes(gemm_mode_env, form):
    """Error of each arithmetic mode against an fp64 product of the same fp32 inputs.  The default split-bf16 (6 products)
    must be as accurate as the exact fp32 fma chain (measured: slightly better); the 3-product form is ~2^-16."""
    from rnntransducer_amd._lib import GEMM_EXACT_F32
    from rnntransducer_amd.ops import gemm
    M, N, K = 384, 512, 2048
    g = torch.Generator().manual_seed(77)
    # wide dynamic range: magnitudes from 1e-6 to 1e3 so that piece exponents matter
    A = torch.randn(M, K, generator=g) * torch.exp(torch.empty(M, K).uniform_(-14, 7, generator=g))
    W = torch.randn(N, K, generator=g) * torch.exp(torch.empty(N, K).uniform_(-14, 7, generator=g))
    if form == "nt":
        a, w, kw = A, W, {}
    elif form == "nn":
        a, w, kw = A, W.t().contiguous(), dict(b_sn=1, b_sk=N)
    else:
        a, w, kw = A.t().contiguous(), W.t().contiguous(), dict(a_mc=True, a_sk=M, b_sn=1, b_sk=N)
    ref = A.double() @ W.double().T
    scale = (A.abs().double() @ W.abs().double().T)

    def run(mode, flags=0):
        os.environ["RNNT_GEMM_MODE"] = mode
        out = torch.full((M, N), float("nan"), device="cuda")
        gemm(M, N, K, a.cuda(), w.cuda(), out, flags=flags, **kw)
        return out.cpu()

    outs = {m: run(m) for m in ("f32", "bf16x6", "bf16x3")}
    err = {m: ((o.double() - ref).abs() / scale).max().item() for m, o in outs.items()}
    assert err["f32"] < 2e-6 and err["bf16x6"] < 2e-6, err
    assert err["bf16x6"] <= 1.5 * err["f32"], err
    assert 1e-7 < err["bf16x3"] < 1e-4, err
    # the per-call flag forces the exact-fp32 MFMA whatever the library default is
    assert torch.equal(run("bf16x6", GEMM_EXACT_F32), outs["f32"])
    assert torch.equal(run("", 0), outs["bf16x6"])  # unset / empty = the default


def test_gemm_split_pieces_are_exact_on_hard_values(gemm_mode_env):
    """Values whose bf16 pieces straddle exponents (1 + 2^-8 + 2^-16 patterns, negative, tiny, powers of two, all-ones
    mantissas): one-term dot products must come out bit-exact, as an fp32 multiply gives them."""
    from rnntransducer_amd.ops import gemm
    os.environ["RNNT_GEMM_MODE"] = "bf16x6"
    vals = torch.tensor([1.0, -1.0, 1.0 + 2 ** -8, 1.0 + 2 ** -16, 1.0 + 2 ** -23, 2.0 - 2 ** -23, -(2.0 - 2 ** -23), 3.0e-30, 7.0e20,
                         0.1, -0.3, 65504.0, 2 ** -100, 1.0 + 2 ** -7 + 2 ** -15 + 2 ** -23, 0.0, 123456.789])
    M = N = len(vals)
    A = torch.zeros(M, 16)
    A[:, 3] = vals               # a single non-zero k: the product is one multiplication, no summation error
    W = torch.zeros(N, 16)
    W[:, 3] = torch.tensor([1.0, 2.0, -4.0, 0.5, 2 ** -20, 2 ** 20, 1.0, -1.0, 8.0, 0.25, 1.0, 2.0, 1.0, -2.0, 1.0, 16.0])
    out = torch.empty(M, N, device="cuda")
    gemm(M, N, 16, A.cuda(), W.cuda(), out)
    assert torch.equal(out.cpu(), A[:, 3:4] * W[:, 3:4].T)  # power-of-two multipliers: exact in fp32


# ------------------------------------------------------------------------------------------------------------------
# half-pair (hp) operands + the f16-MFMA GEMM (csrc/gemm_hp.hip)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (300, 257, 45), (1000, 520, 1024), (512, 256, 4096), (33, 700, 96), (1, 1, 1),
                                   (1400, 700, 64), (5381, 3328, 40)])  # 6 x 3 tiles (bands of 4, tail of 2), 22 x 13 tiles (bands of 8, tail of 6)
def test_gemm_hp_matches_fp64(M, N, K):
    """C = A . B^T on hp operands against an fp64 product of the same fp32 inputs, wide dynamic range (rows of A over six
    decades): error relative to sum |a||b| no worse than the exact fp32 fma chain's bound; ragged edges in M, N and K."""
    from rnntransducer_amd.ops import gemm_hp, hp_split
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.exp(torch.empty(M, 1).uniform_(-14, 3, generator=g))
    W = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().T
    scale = A.abs().double() @ W.abs().double().T + 1e-30
    out = gemm_hp(hp_split(A.cuda()), hp_split(W.cuda()))
    err = ((out.double().cpu() - ref).abs() / scale).max().item()
    assert err < 2e-6, err
    # bias + accumulate epilogues
    base = torch.randn(M, N, generator=g)
    out2 = base.clone().cuda()
    gemm_hp(hp_split(A.cuda()), hp_split(W.cuda()), out=out2, bias=bias.cuda(), accumulate=True)
    ref2 = ref + bias.double() + base.double()
    assert ((out2.double().cpu() - ref2).abs() / (scale + bias.abs().double() + base.abs().double())).max().item() < 2e-6


def test_gemm_hp_transposed_and_shifted_operands_split_k():
    """The weight-gradient form: dW = dG^T . X with both operands given row-major over the contraction index (transposed split),
    deep K (deterministic split-K slabs), and the time-shifted operand of dW_hh (zero fill outside the source)."""
    from rnntransducer_amd.ops import gemm_hp, hp_split
    g = torch.Generator().manual_seed(9)
    Kc, M, N, shift = 5000, 512, 260, 17
    dG = torch.randn(Kc, M, generator=g) * torch.exp(torch.empty(Kc, 1).uniform_(-10, 0, generator=g))
    X = torch.randn(Kc, N, generator=g)
    a, b = hp_split(dG.cuda(), transpose=True), hp_split(X.cuda(), transpose=True)
    out = gemm_hp(a, b)
    ref = dG.double().T @ X.double()
    scale = dG.abs().double().T @ X.abs().double()
    assert ((out.double().cpu() - ref).abs() / scale).max().item() < 2e-6
    assert torch.equal(out, gemm_hp(a, b))                       # split-K slabs are summed in a fixed order
    for sh in (shift, -shift):
        bs = hp_split(X.cuda(), transpose=True, shift=sh)
        Xs = torch.zeros_like(X)
        if sh > 0:
            Xs[:Kc - sh] = X[sh:]
        else:
            Xs[-sh:] = X[:Kc + sh]
        refs = dG.double().T @ Xs.double()
        assert ((gemm_hp(a, bs).double().cpu() - refs).abs() / scale).max().item() < 2e-6


def test_hp_split_represents_fp32_to_two_ulp_and_handles_extremes():
    """hi + lo reproduces x * scale to 2^-23 relative (|x| within 2^17 of amax) and to 2^-40 amax below; zeros, a zero tensor
    and values near the fp32 extremes survive (one-term products against 1.0)."""
    from rnntransducer_amd.ops import gemm_hp, hp_split
    vals = torch.tensor([1.0, -1.0, 1.0 + 2 ** -11, 1.0 + 2 ** -12, 1.0 + 2 ** -23, 2.0 - 2 ** -23, 0.3333333, -0.1, 3.0e-5, 7.1e-6, 0.0,
                         1.5e-7, 65504.0 / 65536, 2 ** -20, -(2 ** -24) * 1.7])
    for mag in (1.0, 1e-30, 1e30):
        A = torch.zeros(len(vals), 32)
        A[:, 0] = vals * mag
        one = torch.zeros(1, 32)
        one[0, 0] = 1.0
        out = gemm_hp(hp_split(A.cuda()), hp_split(one.cuda())).cpu().flatten()
        amax = (vals * mag).abs().max().item()
        tol = (vals * mag).abs() * 2.0 ** -22 + amax * 2.0 ** -39
        assert torch.all((out.double() - (vals * mag).double()).abs() <= tol.double()), (mag, out, vals * mag)
    z = gemm_hp(hp_split(torch.zeros(4, 40).cuda()), hp_split(torch.ones(3, 40).cuda()))
    assert torch.all(z == 0)


@pytest.mark.parametrize("xcd_skip", [0x00, 0x0F, 0xFE])
def test_gemm_hp_grouped_queue_launch(xcd_skip):
    """Three weight-gradient-shaped products (ragged M / N, deep K -> split-K slabs) in ONE queue-driven launch, on all XCDs, on
    XCDs 4-7 only and on XCD 0 only: each result against fp64, run-to-run bitwise reproducible (which workgroup draws which unit
    varies, the arithmetic of a unit does not), and the accumulate epilogue."""
    from rnntransducer_amd.ops import gemm_hp_grouped, hp_split
    g = torch.Generator().manual_seed(21)
    Kc = 9000
    shapes = [(700, 300), (512, 130), (260, 513)]
    mats = [(torch.randn(Kc, m, generator=g) * torch.exp(torch.empty(Kc, 1).uniform_(-8, 0, generator=g)), torch.randn(Kc, n, generator=g))
            for m, n in shapes]
    pairs = [(hp_split(a.cuda(), transpose=True), hp_split(b.cuda(), transpose=True)) for a, b in mats]
    outs = gemm_hp_grouped(pairs, xcd_skip=xcd_skip, check=True)   # check: the launch's own "every unit was drawn" word is clean
    again = gemm_hp_grouped(pairs, xcd_skip=xcd_skip)
    base = [torch.randn(m, n, generator=g) for m, n in shapes]
    acc = gemm_hp_grouped(pairs, outs=[b.clone().cuda() for b in base], accumulate=True, xcd_skip=xcd_skip)
    for (a, b), o, o2, o3, bs in zip(mats, outs, again, acc, base):
        ref = a.double().T @ b.double()
        scale = a.abs().double().T @ b.abs().double() + 1e-30
        assert ((o.double().cpu() - ref).abs() / scale).max().item() < 2e-6
        assert torch.equal(o, o2)
        assert ((o3.double().cpu() - ref - bs.double()).abs() / (scale + bs.abs().double())).max().item() < 2e-6


def test_gemm_hp_grouped_rejects_bad_arguments():
    from rnntransducer_amd.ops import gemm_hp_grouped, hp_split
    a, b = hp_split(torch.randn(64, 40).cuda(), transpose=True), hp_split(torch.randn(64, 24).cuda(), transpose=True)
    with pytest.raises(ValueError):
        gemm_hp_grouped([(a, b)] * 5)
    with pytest.raises(ValueError):
        gemm_hp_grouped([(a, b)], xcd_skip=0xFF)            # no XCD left
    with pytest.raises(ValueError):
        gemm_hp_grouped([(a, hp_split(torch.randn(65, 24).cuda(), transpose=True))])   # contraction lengths differ


@pytest.mark.parametrize("M,C", [(64, 256), (1000, 4096), (77, 300), (33, 40)])
def test_hp_split_both_orientations_in_one_pass_is_bitwise_two_splits(M, C):
    """rnnt_hip_hp_split_both (row maxima and column maxima given) writes exactly the planes the row-major and the transposed
    rnnt_hip_hp_split write — ragged M and C (zero fill of the padded k range in both orientations)."""
    from rnntransducer_amd import _lib
    from rnntransducer_amd.ops import HpTensor, _addr, _stream, hp_split
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * torch.exp(torch.empty(M, 1).uniform_(-9, 2, generator=g))).cuda()
    x[M // 2] = 0                                   # an all-zero row (a padded frame)
    rm, tr = hp_split(x), hp_split(x, transpose=True)
    rm2, tr2 = HpTensor(M, C, x.device), HpTensor(C, M, x.device)
    rm2.planes.fill_(0x5A)
    tr2.planes.fill_(0x5A)
    _lib.check(_lib.lib().rnnt_hip_hp_split_both(_addr(x), M, C, C, _addr(rm.amax), _addr(tr.amax), _addr(rm2.planes), _addr(tr2.planes),
                                                 _stream()), "hp_split_both")
    n_rm, n_tr = _lib.lib().rnnt_hip_hp_bytes(M, C), _lib.lib().rnnt_hip_hp_bytes(C, M)
    assert torch.equal(rm2.planes[:n_rm], rm.planes[:n_rm])
    assert torch.equal(tr2.planes[:n_tr], tr.planes[:n_tr])


@pytest.mark.parametrize("hp", [True, False])
def test_linear_big_products_on_the_half_pair_path_match_fp64(hp, monkeypatch):
    """nn.Linear forward / backward (networks/encoder.py:76,103) at a shape LinearFn routes through the half-pair GEMM
    (M >= 1024, N >= 256, K >= 1024) and, with RNNT_GEMM_NO_HP, through gemm.hip: y, dx, dW, db against an fp64 product of the
    same fp32 values, error relative to sum |a||b| per element (the bound the hp GEMM tests use)."""
    from rnntransducer_amd.ops import LinearFn
    if not hp:
        monkeypatch.setenv("RNNT_GEMM_NO_HP", "1")
    g = torch.Generator().manual_seed(5)
    T, B, K, N = 70, 32, 1024, 320
    x = (torch.randn(T, B, K, generator=g) * torch.exp(torch.empty(T, B, 1).uniform_(-6, 2, generator=g))).requires_grad_()
    W = (torch.randn(N, K, generator=g) * 0.05).requires_grad_()
    b = torch.randn(N, generator=g).requires_grad_()
    dy = torch.randn(T, B, N, generator=g) * torch.exp(torch.empty(T, B, 1).uniform_(-8, 0, generator=g))
    xd, Wd, bd = (t.detach().double().requires_grad_() for t in (x, W, b))
    yd = xd @ Wd.T + bd
    yd.backward(dy.double())
    xc, Wc, bc = (t.detach().cuda().requires_grad_() for t in (x, W, b))
    y = LinearFn.apply(xc, Wc, bc)
    y.backward(dy.cuda())
    x2, dy2 = x.detach().double().view(-1, K), dy.double().view(-1, N)
    scales = {"y": x2.abs() @ Wd.detach().abs().T + bd.detach().abs(), "dx": dy2.abs() @ Wd.detach().abs(),
              "dW": dy2.abs().T @ x2.abs(), "db": dy2.abs().sum(0)}
    got = {"y": y.detach().view(-1, N), "dx": xc.grad.view(-1, K), "dW": Wc.grad, "db": bc.grad}
    want = {"y": yd.detach().view(-1, N), "dx": xd.grad.view(-1, K), "dW": Wd.grad, "db": bd.grad}
    for k in got:
        err = ((got[k].double().cpu() - want[k]).abs() / (scales[k] + 1e-30)).max().item()
        assert err < (2e-5 if k == "db" else 3e-6 if hp else 6e-6), (k, err)   # db: an fp32 column sum over 2240 rows
