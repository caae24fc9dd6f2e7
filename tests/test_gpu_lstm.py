"""Persistent HIP LSTM (fwd + bwd) vs torch.nn.LSTM on the CPU in float64 over a PackedSequence — the oracle for
networks/encoder.py:93-102 / decoder.py:105-120 semantics (zero outputs on padding, reverse direction starting at
each sequence's last frame, no gradient from padded frames)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
FWD_ATOL, GRAD_RTOL = 2e-5, 2e-4


def _oracle(x, lens, ref, dy):
    T = x.shape[1]
    xr = x.double().requires_grad_(True)
    packed = nn.utils.rnn.pack_padded_sequence(xr, torch.tensor(lens), batch_first=True, enforce_sorted=False)
    out, _ = ref(packed)
    out, _ = nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=T)
    out.backward(dy.double())
    return out.detach(), xr.grad


@pytest.mark.parametrize("B,T,I,H,L,bi", [(1, 1, 4, 4, 1, False), (3, 7, 5, 8, 1, True), (2, 9, 12, 12, 2, True),
                                          (5, 20, 80, 16, 2, True), (17, 13, 8, 32, 1, False), (33, 6, 16, 20, 1, True),
                                          (4, 30, 80, 128, 1, True), (64, 5, 8, 8, 1, True), (2, 25, 16, 512, 1, True),
                                          (3, 12, 40, 640, 1, True),
                                          # register-resident bf16-piece kernels (H % 128 == 0): 16-row groups, H = 256 two layers
                                          # one direction, prediction-net shape (one direction, 4-row groups), ragged 8-row groups
                                          (64, 9, 16, 512, 1, True), (7, 15, 24, 256, 2, False), (32, 10, 16, 128, 1, False),
                                          (32, 17, 80, 512, 2, True),
                                          # H = 1024 (the shipped config.json size): 8-wave register-form kernels
                                          (16, 6, 24, 1024, 1, True), (5, 4, 8, 1024, 2, False),
                                          (6, 7, 16, 384, 1, True), (9, 5, 16, 768, 1, True),
                                          # register form at the edges: one frame, one utterance, 3 ragged rows, 17 rows (two 9-row groups)
                                          (1, 1, 8, 128, 1, True), (3, 2, 8, 256, 1, False), (17, 4, 8, 128, 2, True),
                                          # 8-wave forms with 16-row groups
                                          (32, 4, 8, 1024, 1, True), (64, 3, 8, 640, 1, True)])
def test_lstm_stack_fwd_bwd(B, T, I, H, L, bi):
    from rnntransducer_amd.networks.rnn import HipLSTM
    from rnntransducer_amd.ops import lstm_check, lstm_workspace
    torch.manual_seed(B * 100 + T + H)
    ref = nn.LSTM(I, H, L, batch_first=True, bidirectional=bi).double()
    hip = HipLSTM(I, H, L, dropout=0.0, bidirectional=bi)
    hip.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    hip = hip.cuda()
    g = torch.Generator().manual_seed(1)
    lens = [T] + torch.randint(1, T + 1, (B - 1,), generator=g).tolist()
    x = torch.randn(B, T, I, generator=g)
    for b in range(B):
        x[b, lens[b]:] = 0
    D = 2 if bi else 1
    dy = torch.randn(B, T, D * H, generator=g)
    ref_out, ref_dx = _oracle(x, lens, ref, dy)

    x_tm = x.transpose(0, 1).contiguous().cuda().requires_grad_(True)
    lens_dev = torch.tensor(lens, dtype=torch.int32, device="cuda")
    y = hip(x_tm, lens_dev)
    y.backward(dy.transpose(0, 1).contiguous().cuda())
    torch.cuda.synchronize()
    err = (y.detach().transpose(0, 1).double().cpu() - ref_out).abs().max().item()
    assert err < FWD_ATOL, f"forward err {err}"
    for b in range(B):  # padded outputs exactly zero
        assert torch.all(y[lens[b]:, b] == 0)

    def close(name, got, want):
        scale = max(want.abs().max().item(), 1e-3)
        e = (got.double().cpu() - want).abs().max().item()
        assert e < GRAD_RTOL * scale + 1e-6, f"{name}: err {e} scale {scale}"

    close("dx", x_tm.grad.transpose(0, 1), ref_dx)
    for name, p in ref.named_parameters():
        close(name, getattr(hip, name).grad, p.grad)


@pytest.mark.parametrize("B,T,I,H,L,bi,cell,p", [
    (32, 40, 80, 512, 2, True, "lstm", 0.0),     # config-2 groups (8 rows), layer-0 width + inner width, half-pair products
    (8, 150, 128, 512, 1, True, "lstm", 0.0),    # 4-row groups (config-3 geometry)
    (16, 70, 160, 640, 2, True, "lstm", 0.0),    # 20-unit workgroups, two-barrier backward
    (32, 80, 256, 256, 2, True, "gru", 0.0),     # GRU: hidden-side gate gradients have their own planes
    (32, 260, 128, 128, 1, False, "rnn", 0.0),   # Elman, one direction
    (32, 40, 80, 512, 3, True, "lstm", 0.3),     # dropout between layers: plan vs dense on the same seed
])
def test_ragged_plan_skips_padding_without_changing_results(B, T, I, H, L, bi, cell, p):
    """rnnt_lstm_desc.row_idx (ops.RaggedPlan): the big products run over the valid frames only (operand tiles gathered, results
    scattered, transposed planes packed along the contraction) and every sync group of the recurrences runs max(lens of its rows)
    steps.  Against torch float64 on the CPU (dropout 0) with the same tolerances as the dense path, for rows in collate order AND
    sorted by length; with dropout, against the dense HIP run on the same seed."""
    from rnntransducer_amd import _lib
    from rnntransducer_amd.networks.rnn import RNN_CELLS
    from rnntransducer_amd.ops import LstmStackFn, RaggedPlan
    torch.manual_seed(B + T + H)
    D = 2 if bi else 1
    cell_id = RNN_CELLS[cell].CELL if cell != "rnn" else 2
    import os
    if not any(os.environ.get(k) for k in ("RNNT_LSTM_NO_V5", "RNNT_GEMM_NO_HP", "RNNT_LSTM_V1", "RNNT_LSTM_V2", "RNNT_LSTM_EXACT_MATH")):
        assert _lib.lib().rnnt_hip_lstm_takes_row_idx(T, B, I, H, D, cell_id) == 1   # the table is honoured at this shape, not ignored
    # (under the fallback switches the library ignores the table and computes all rows: the results below must hold either way)
    assert _lib.lib().rnnt_hip_lstm_takes_row_idx(7, 3, 8, 16, D, cell_id) == 0     # small shapes ignore it and stay dense
    ref_cls = {"lstm": nn.LSTM, "gru": nn.GRU, "rnn": nn.RNN}[cell]
    ref = ref_cls(I, H, L, batch_first=True, bidirectional=bi).double()
    hip = RNN_CELLS[cell](I, H, L, dropout=p, bidirectional=bi)
    hip.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    hip = hip.cuda().train()
    g = torch.Generator().manual_seed(2)
    for order in ("collate", "sorted"):
        lens = [T] + torch.randint(max(1, T // 3), T + 1, (B - 1,), generator=g).tolist()
        if order == "sorted":
            lens = sorted(lens, reverse=True)
        x = torch.randn(B, T, I, generator=g)
        for b in range(B):
            x[b, lens[b]:] = 0
        dy = torch.randn(B, T, D * H, generator=g)
        x_tm = x.transpose(0, 1).contiguous().cuda()
        dy_tm = dy.transpose(0, 1).contiguous().cuda()
        lens_dev = torch.tensor(lens, dtype=torch.int32, device="cuda")
        plan = RaggedPlan(lens, T, "cuda")
        assert not plan.dense and plan.n_rows == sum(lens)

        def run(lens_arg):
            for q in hip.parameters():
                q.grad = None
            xx = x_tm.clone().requires_grad_(True)
            pp = p if L > 1 else 0.0
            y = LstmStackFn.apply(xx, lens_arg, H, L, bi, pp, 4242, hip.CELL, False, *hip.flat_weights())
            y.backward(dy_tm)
            torch.cuda.synchronize()
            return y.detach(), xx.grad, {k: q.grad.clone() for k, q in hip.named_parameters()}

        y_p, dx_p, gr_p = run(plan)
        for b in range(B):
            assert torch.all(y_p[lens[b]:, b] == 0)
        if p > 0:   # same seed, same masks: the dense run is the reference
            y_d, dx_d, gr_d = run(lens_dev)
            assert (y_p - y_d).abs().max().item() < 1e-6
            for b in range(B):   # dx of padded frames is unspecified with a plan (never consumed): compare valid frames
                assert (dx_p[:lens[b], b] - dx_d[:lens[b], b]).abs().max().item() < 1e-5 * max(1.0, dx_d.abs().max().item())
            for k in gr_d:
                assert (gr_p[k] - gr_d[k]).abs().max().item() < 2e-5 * max(gr_d[k].abs().max().item(), 1e-3), k
            continue
        ref.zero_grad()
        ref_out, ref_dx = _oracle(x, lens, ref, dy)
        err = (y_p.transpose(0, 1).double().cpu() - ref_out).abs().max().item()
        assert err < FWD_ATOL, f"{order}: forward err {err}"

        def close(name, got, want):
            scale = max(want.abs().max().item(), 1e-3)
            e = (got.double().cpu() - want).abs().max().item()
            assert e < GRAD_RTOL * scale + 1e-6, f"{order} {name}: err {e} scale {scale}"

        dxv = dx_p.transpose(0, 1).clone()
        for b in range(B):
            dxv[b, lens[b]:] = 0     # (unspecified with a plan; the oracle has 0 there)
        close("dx", dxv, ref_dx)
        for name, q in ref.named_parameters():
            close(name, gr_p[name], q.grad)


@pytest.mark.parametrize("B,T,lens", [
    (1, 1100, [700]),                                  # one utterance shorter than the padded length
    (9, 130, [130, 1, 1, 1, 64, 1, 129, 2, 1]),        # one-frame rows (the reverse direction starts AND ends at frame 0), 9 rows = 3 groups
    (4, 300, [1, 1, 1, 300]),                          # fewer valid rows than one 256-row GEMM tile apart from a single long row
    (70, 16, [16] * 35 + [3] * 35),                    # B above the per-launch limit: the module slices the batch and runs the slices dense
])
def test_ragged_plan_edge_cases(B, T, lens, monkeypatch):
    """Valid-frame table at the edges (RNNT_GEMM_FORCE_HP puts these small shapes on the half-pair products, where the table is
    honoured): outputs and every gradient against torch float64."""
    from rnntransducer_amd.networks.rnn import HipLSTM
    from rnntransducer_amd.ops import RaggedPlan
    monkeypatch.setenv("RNNT_GEMM_FORCE_HP", "1")
    I, H = 64, 128
    torch.manual_seed(B * 7 + T)
    ref = nn.LSTM(I, H, 2, batch_first=True, bidirectional=True).double()
    hip = HipLSTM(I, H, 2, dropout=0.0, bidirectional=True)
    hip.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    hip = hip.cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, I, generator=g)
    for b in range(B):
        x[b, lens[b]:] = 0
    dy = torch.randn(B, T, 2 * H, generator=g)
    ref_out, ref_dx = _oracle(x, lens, ref, dy)
    x_tm = x.transpose(0, 1).contiguous().cuda().requires_grad_(True)
    y = hip(x_tm, RaggedPlan(lens, T, "cuda"))
    y.backward(dy.transpose(0, 1).contiguous().cuda())
    torch.cuda.synchronize()
    assert (y.detach().transpose(0, 1).double().cpu() - ref_out).abs().max().item() < FWD_ATOL
    dx = x_tm.grad.transpose(0, 1).clone()
    for b in range(B):
        assert torch.all(y[lens[b]:, b] == 0)
        dx[b, lens[b]:] = 0
    for name, got, want in [("dx", dx, ref_dx)] + [(k, getattr(hip, k).grad, p.grad) for k, p in ref.named_parameters()]:
        scale = max(want.abs().max().item(), 1e-3)
        e = (got.double().cpu() - want).abs().max().item()
        assert e < GRAD_RTOL * scale + 1e-6, f"{name}: err {e} scale {scale}"


def test_lstm_init_matches_torch_rng_stream():
    from rnntransducer_amd.networks.rnn import HipLSTM
    torch.manual_seed(7)
    ref = nn.LSTM(6, 8, 2, bidirectional=True)
    torch.manual_seed(7)
    hip = HipLSTM(6, 8, 2, bidirectional=True)
    assert list(ref.state_dict()) == list(hip.state_dict())
    for k, v in ref.state_dict().items():
        assert torch.equal(v, hip.state_dict()[k]), k


def test_lstm_dropout_mask_is_consistent_between_fwd_and_bwd():
    """Inter-layer dropout: the backward must regenerate the forward's mask.  Check d(sum y)/dx by finite differences
    on the SAME seed (the op is deterministic given the seed), and that ~p of the layer-0 outputs are dropped."""
    from rnntransducer_amd.ops import LstmStackFn
    from rnntransducer_amd.networks.rnn import HipLSTM
    torch.manual_seed(0)
    hip = HipLSTM(8, 16, 2, dropout=0.5, bidirectional=True).cuda()
    x = torch.randn(6, 3, 8, device="cuda", dtype=torch.float32)
    lens = torch.tensor([6, 4, 2], dtype=torch.int32, device="cuda")
    w = hip.flat_weights()

    def run(xx):
        return LstmStackFn.apply(xx, lens, 16, 2, True, 0.5, 1234, 0, False, *w)

    xr = x.clone().requires_grad_(True)
    proj = torch.randn(6, 3, 32, device="cuda")
    (run(xr) * proj).sum().backward()
    eps = 1e-2
    for idx in [(0, 0, 0), (3, 1, 5), (1, 2, 7)]:
        xp, xm = x.clone(), x.clone()
        xp[idx] += eps
        xm[idx] -= eps
        fd = ((run(xp) * proj).sum() - (run(xm) * proj).sum()).item() / (2 * eps)
        assert abs(fd - xr.grad[idx].item()) < 2e-2 * max(1.0, abs(fd)), (idx, fd, xr.grad[idx].item())
    assert torch.equal(run(x), run(x))


def test_lstm_stack_validates_weight_shapes_on_the_host():
    """A wrong weight list must never reach the kernels (they index raw pointers): ValueError before any launch."""
    from rnntransducer_amd.networks.rnn import HipLSTM
    from rnntransducer_amd.ops import LstmStackFn
    hip = HipLSTM(8, 16, 2, dropout=0.0, bidirectional=True).cuda()
    x = torch.randn(6, 3, 8, device="cuda")
    lens = torch.tensor([6, 4, 2], dtype=torch.int32, device="cuda")
    w = hip.flat_weights()
    with pytest.raises(ValueError):
        LstmStackFn.apply(x, lens, 16, 2, True, 0.0, 1, 0, False, *w[1:])            # one tensor short
    with pytest.raises(ValueError):
        LstmStackFn.apply(x, lens, 16, 2, True, 0.0, 1, 0, False, *(w[1:] + w[:1]))  # rotated: shapes do not match
    with pytest.raises(ValueError):
        LstmStackFn.apply(x, lens, 32, 2, True, 0.0, 1, 0, False, *w)                # wrong hidden size
    with pytest.raises(ValueError):
        LstmStackFn.apply(x, lens[:2], 16, 2, True, 0.0, 1, 0, False, *w)            # lens does not cover the batch
    torch.cuda.synchronize()


def test_lstm_rejects_unsupported():
    from rnntransducer_amd._lib import RnntHipError
    from rnntransducer_amd.networks.rnn import HipLSTM
    with pytest.raises(ValueError):
        HipLSTM(4, 6, 1)
    hip = HipLSTM(4, 8, 1).cuda()
    with pytest.raises(ValueError):
        hip(torch.zeros(2, 2, 4, device="cuda"), torch.tensor([2, 2], device="cuda"))  # int64 lengths
    with pytest.raises(RnntHipError):
        hip(torch.zeros(2, 2, 4), torch.tensor([2, 2], dtype=torch.int32))            # CPU tensors: no fallback


@pytest.mark.parametrize("B,T,I,H,L,bi", [(3, 7, 5, 8, 1, True), (33, 6, 16, 20, 2, True), (2, 25, 16, 512, 1, True)])
def test_lstm_v1_fallback_kernels_still_match(monkeypatch, B, T, I, H, L, bi):
    """The 128-workgroup-per-direction kernels (used when a shape does not fit the grouped v2 decomposition, e.g.
    H = 1024 with B > 16) stay covered: force them with RNNT_LSTM_V1=1."""
    monkeypatch.setenv("RNNT_LSTM_V1", "1")
    test_lstm_stack_fwd_bwd(B, T, I, H, L, bi)


@pytest.mark.parametrize("env", ["RNNT_LSTM_FWD5_4W", "RNNT_LSTM_NO_V5"])
def test_lstm_h512_alternative_forward_forms_still_match(monkeypatch, env):
    """H = 512 runs the v5 forward with 8 waves x 2 k-steps by default; the 4-wave x 4 k-step form (RNNT_LSTM_FWD5_4W) and round 1's
    flag-protocol / bf16-piece kernels (RNNT_LSTM_NO_V5) stay selectable and correct."""
    monkeypatch.setenv(env, "1")
    test_lstm_stack_fwd_bwd(32, 17, 80, 512, 2, True)


@pytest.mark.parametrize("B,T,I,H,L,bi", [(32, 11, 16, 512, 1, True), (3, 12, 40, 640, 1, True), (64, 9, 16, 512, 1, True)])
def test_lstm_v2_lds_resident_kernels_still_match(monkeypatch, B, T, I, H, L, bi):
    """The v2 kernels (W_hh slice in LDS, f32-input 4x4x1 MFMA, gathered dG) remain the path for every H outside
    {128, 256, 512, 640, 1024}: keep them covered at the sizes the register-resident kernels normally take (RNNT_LSTM_V2=1)."""
    monkeypatch.setenv("RNNT_LSTM_V2", "1")
    test_lstm_stack_fwd_bwd(B, T, I, H, L, bi)


@pytest.mark.parametrize("B,T,I,H,L,bi", [(32, 40, 144, 128, 2, True), (32, 33, 80, 256, 2, False), (64, 17, 128, 512, 1, True)])
def test_lstm_big_products_on_half_pair_operands(monkeypatch, B, T, I, H, L, bi):
    """The hp path (gemm_hp.hip: input projection, dX, dW_ih, dW_hh on the f16 matrix cores) forced on at sizes the oracle
    finishes quickly — ragged T*B (not a multiple of 256), I not a multiple of 32, one and two directions; and the same
    shapes with it disabled (RNNT_GEMM_NO_HP)."""
    monkeypatch.setenv("RNNT_GEMM_FORCE_HP", "1")
    test_lstm_stack_fwd_bwd(B, T, I, H, L, bi)
    monkeypatch.delenv("RNNT_GEMM_FORCE_HP")
    monkeypatch.setenv("RNNT_GEMM_NO_HP", "1")
    test_lstm_stack_fwd_bwd(B, T, I, H, L, bi)


def test_xcd_local_exchange_is_bitwise_identical_to_write_through(monkeypatch):
    """B=32, H=512, bidirectional -> 8 sync groups x 32 workgroups = the BASELINE config-2 decomposition.  When a group
    is verified to sit on one XCD it exchanges through that XCD's L2 (plain stores); otherwise / when disabled it uses
    the sc1 write-through protocol.  Same arithmetic either way: outputs and gradients must be bitwise equal, and
    both must match the oracle."""
    from rnntransducer_amd.networks.rnn import HipLSTM
    B, T, I, H = 32, 24, 16, 512
    torch.manual_seed(3)
    ref = nn.LSTM(I, H, 1, batch_first=True, bidirectional=True).double()
    hip = HipLSTM(I, H, 1, bidirectional=True)
    hip.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    hip = hip.cuda()
    g = torch.Generator().manual_seed(5)
    lens = [T] + torch.randint(1, T + 1, (B - 1,), generator=g).tolist()
    x = torch.randn(B, T, I, generator=g)
    dy = torch.randn(B, T, 2 * H, generator=g)
    ref_out, ref_dx = _oracle(x, lens, ref, dy)
    results = []
    for disable in (False, True):
        if disable:
            monkeypatch.setenv("RNNT_LSTM_NO_XCD_LOCAL", "1")
        hip.zero_grad()
        x_tm = x.transpose(0, 1).contiguous().cuda().requires_grad_(True)
        y = hip(x_tm, torch.tensor(lens, dtype=torch.int32, device="cuda"))
        y.backward(dy.transpose(0, 1).contiguous().cuda())
        torch.cuda.synchronize()
        results.append((y.detach().clone(), x_tm.grad.clone(), hip.weight_hh_l0.grad.clone(), hip.weight_ih_l0_reverse.grad.clone()))
    for a, b in zip(*results):
        assert torch.equal(a, b)
    assert (results[0][0].transpose(0, 1).double().cpu() - ref_out).abs().max().item() < FWD_ATOL
    assert (results[0][1].transpose(0, 1).double().cpu() - ref_dx).abs().max().item() < GRAD_RTOL * max(ref_dx.abs().max().item(), 1e-3) + 1e-6


@pytest.mark.parametrize("cell", ["gru", "rnn_tanh", "rnn_relu"])
@pytest.mark.parametrize("B,T,I,H,L,bi", [(1, 1, 4, 4, 1, False), (3, 7, 5, 8, 2, True), (5, 20, 80, 16, 2, True), (17, 9, 8, 32, 1, True),
                                          (32, 11, 16, 512, 1, True), (4, 14, 24, 640, 1, True), (16, 5, 16, 1024, 1, True), (9, 5, 16, 768, 1, True)])
def test_gru_and_elman_cells_fwd_bwd(cell, B, T, I, H, L, bi):
    """the reference's other supported_rnns (encoder.py:48-52) on the same persistent kernels, against
    torch.nn.GRU / torch.nn.RNN on the CPU in float64 over a PackedSequence."""
    from rnntransducer_amd.networks.rnn import HipGRU, HipRNN
    torch.manual_seed(B * 100 + T + H)
    if cell == "gru":
        ref = nn.GRU(I, H, L, batch_first=True, bidirectional=bi).double()
        hip = HipGRU(I, H, L, dropout=0.0, bidirectional=bi)
    else:
        nl = cell.split("_")[1]
        ref = nn.RNN(I, H, L, nonlinearity=nl, batch_first=True, bidirectional=bi).double()
        hip = HipRNN(I, H, L, dropout=0.0, bidirectional=bi, nonlinearity=nl)
    assert list(ref.state_dict()) == list(hip.state_dict())
    hip.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    hip = hip.cuda()
    g = torch.Generator().manual_seed(1)
    lens = [T] + torch.randint(1, T + 1, (B - 1,), generator=g).tolist()
    x = torch.randn(B, T, I, generator=g)
    for b in range(B):
        x[b, lens[b]:] = 0
    D = 2 if bi else 1
    dy = torch.randn(B, T, D * H, generator=g)
    ref_out, ref_dx = _oracle(x, lens, ref, dy)
    x_tm = x.transpose(0, 1).contiguous().cuda().requires_grad_(True)
    y = hip(x_tm, torch.tensor(lens, dtype=torch.int32, device="cuda"))
    y.backward(dy.transpose(0, 1).contiguous().cuda())
    torch.cuda.synchronize()
    assert (y.detach().transpose(0, 1).double().cpu() - ref_out).abs().max().item() < FWD_ATOL
    for b in range(B):
        assert torch.all(y[lens[b]:, b] == 0)

    def close(name, got, want):
        scale = max(want.abs().max().item(), 1e-3)
        e = (got.double().cpu() - want).abs().max().item()
        assert e < GRAD_RTOL * scale + 1e-6, f"{name}: err {e} scale {scale}"

    close("dx", x_tm.grad.transpose(0, 1), ref_dx)
    for name, p in ref.named_parameters():
        close(name, getattr(hip, name).grad, p.grad)


@pytest.mark.parametrize("cell", ["gru", "rnn_tanh"])
def test_gru_and_elman_cells_on_half_pair_operands(monkeypatch, cell):
    """GRU's hidden-side gate gradients take their own transposed planes for dW_hh; Elman cells use one of four gate slots."""
    monkeypatch.setenv("RNNT_GEMM_FORCE_HP", "1")
    test_gru_and_elman_cells_fwd_bwd(cell, 32, 35, 136, 128, 2, True)


def test_batches_beyond_one_launch_are_split_along_b():
    """B=70 LSTM (one launch takes <= 64 rows) and a 1024-wide GRU with B=20 (grouped form takes <= 16 rows at that width):
    the module runs batch slices back to back; results equal the oracle."""
    from rnntransducer_amd.networks.rnn import HipGRU, HipLSTM
    for cls, ref_cls, (B, T, I, H) in ((HipLSTM, nn.LSTM, (70, 5, 8, 16)), (HipGRU, nn.GRU, (20, 6, 8, 1024))):
        torch.manual_seed(B)
        ref = ref_cls(I, H, 1, batch_first=True, bidirectional=True).double()
        hip = cls(I, H, 1, bidirectional=True)
        hip.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        hip = hip.cuda()
        g = torch.Generator().manual_seed(2)
        lens = [T] + torch.randint(1, T + 1, (B - 1,), generator=g).tolist()
        x = torch.randn(B, T, I, generator=g)
        dy = torch.randn(B, T, 2 * H, generator=g)
        ref_out, ref_dx = _oracle(x, lens, ref, dy)
        x_tm = x.transpose(0, 1).contiguous().cuda().requires_grad_(True)
        y = hip(x_tm, torch.tensor(lens, dtype=torch.int32, device="cuda"))
        y.backward(dy.transpose(0, 1).contiguous().cuda())
        assert (y.detach().transpose(0, 1).double().cpu() - ref_out).abs().max().item() < FWD_ATOL
        for name, p in ref.named_parameters():
            scale = max(p.grad.abs().max().item(), 1e-3)
            assert (getattr(hip, name).grad.double().cpu() - p.grad).abs().max().item() < GRAD_RTOL * scale + 1e-6, name


@pytest.mark.parametrize("force_hp", [False, True])
def test_backward_overlap_of_weight_gradients_equals_the_serial_backward(monkeypatch, force_hp):
    """ops.LstmStackFn.backward issues rnnt_hip_lstm_bwd in two phases for stacks of >= 2 layers: recurrence + dx on the caller's
    stream, weight / bias gradients on a second stream beside the next layer's recurrence (alternating workspaces).  Against the
    single-call backward (RNNT_LSTM_NO_OVERLAP): bitwise equal while the products stay on gemm.hip; with the big products forced
    onto the half-pair path the overlapped form uses the grouped queue launch (a different split-K partition): equal to 2e-6 of the
    gradient's scale.  Three layers, dropout, ragged lengths, two backward passes through the same module (workspace reuse)."""
    from rnntransducer_amd.networks.rnn import HipLSTM
    if force_hp:
        monkeypatch.setenv("RNNT_GEMM_FORCE_HP", "1")
    B, T, I, H, L = 32, 37, 136, 128, 3
    torch.manual_seed(11)
    hip = HipLSTM(I, H, L, dropout=0.25, bidirectional=True).cuda().train()
    g = torch.Generator().manual_seed(4)
    lens = torch.tensor([T] + torch.randint(1, T + 1, (B - 1,), generator=g).tolist(), dtype=torch.int32, device="cuda")
    x = torch.randn(T, B, I, generator=g).cuda()
    dy = torch.randn(T, B, 2 * H, generator=g).cuda()
    results = []
    for serial in (False, True, False):
        if serial:
            monkeypatch.setenv("RNNT_LSTM_NO_OVERLAP", "1")
        else:
            monkeypatch.delenv("RNNT_LSTM_NO_OVERLAP", raising=False)
            monkeypatch.setenv("RNNT_LSTM_OVERLAP", "1")   # this shape's recurrence fills the chip: not overlapped by default
        hip.zero_grad()
        xin = x.clone().requires_grad_(True)
        torch.manual_seed(77)
        hip._step = 5           # the dropout seed derives from (torch seed, module step counter): the same masks in all three runs
        y = hip(xin, lens)
        y.backward(dy)
        torch.cuda.synchronize()
        results.append([xin.grad.clone()] + [p.grad.clone() for p in hip.parameters()])
    over, ser, over2 = results
    for a, b, c in zip(over, ser, over2):
        assert torch.equal(a, c)                       # the overlapped form is reproducible run to run
        if force_hp:
            assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1e-3)
        else:
            assert torch.equal(a, b)


@pytest.mark.parametrize("cell,B,T,I,H,bi", [("lstm", 16, 6, 24, 1024, True), ("gru", 9, 5, 16, 768, True), ("lstm", 5, 4, 8, 1024, False)])
def test_v5_recurrences_at_768_and_1024_are_opt_in_and_correct(monkeypatch, cell, B, T, I, H, bi):
    """RNNT_LSTM_V5_WIDE=1 runs the tagged-payload / half-pair recurrences at H = 768 / 1024 (8 waves, 48 / 64 producers per sync
    group).  Not the default — measured slower than the v3 / v4 forms at the shipped config's size (lstm5.hip) — but kept correct."""
    monkeypatch.setenv("RNNT_LSTM_V5_WIDE", "1")
    if cell == "lstm":
        test_lstm_stack_fwd_bwd(B, T, I, H, 1, bi)
    else:
        test_gru_and_elman_cells_fwd_bwd(cell, B, T, I, H, 1, bi)


def test_backward_overlap_is_chosen_where_the_recurrence_leaves_xcds_free():
    """B = 8, H = 512, two directions (the c3 shape): 4 sync groups on 4 XCDs -> phase 2 of a layer runs beside phase 1 of the next
    on the other 4 (grouped queue launch that skips XCDs 0-3).  B = 32 (c2): 8 groups fill the chip -> single-call backward."""
    from rnntransducer_amd import _lib
    from rnntransducer_amd.networks.rnn import HipLSTM
    assert _lib.lib().rnnt_hip_lstm_free_xcds(100, 8, 512, 2, 0) == 4
    assert _lib.lib().rnnt_hip_lstm_free_xcds(100, 32, 512, 2, 0) == 0
    assert _lib.lib().rnnt_hip_lstm_free_xcds(100, 16, 1024, 2, 1) == 0
    B, T, I, H, L = 8, 140, 1024, 512, 2      # T*B >= 1024 and 4H*D*T*B >= 2^22: the half-pair path by itself
    torch.manual_seed(5)
    ref = nn.LSTM(I, H, L, batch_first=True, bidirectional=True).double()
    hip = HipLSTM(I, H, L, dropout=0.0, bidirectional=True)
    hip.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    hip = hip.cuda()
    g = torch.Generator().manual_seed(2)
    lens = [T] + torch.randint(T // 2, T + 1, (B - 1,), generator=g).tolist()
    x = torch.randn(B, T, I, generator=g)
    for b in range(B):
        x[b, lens[b]:] = 0
    dy = torch.randn(B, T, 2 * H, generator=g)
    ref_out, ref_dx = _oracle(x, lens, ref, dy)
    x_tm = x.transpose(0, 1).contiguous().cuda().requires_grad_(True)
    y = hip(x_tm, torch.tensor(lens, dtype=torch.int32, device="cuda"))
    y.backward(dy.transpose(0, 1).contiguous().cuda())
    assert (y.detach().transpose(0, 1).double().cpu() - ref_out).abs().max().item() < FWD_ATOL
    for name, p in ref.named_parameters():
        scale = max(p.grad.abs().max().item(), 1e-3)
        assert (getattr(hip, name).grad.double().cpu() - p.grad).abs().max().item() < GRAD_RTOL * scale + 1e-6, name
    assert (x_tm.grad.transpose(0, 1).double().cpu() - ref_dx).abs().max().item() < GRAD_RTOL * max(ref_dx.abs().max().item(), 1e-3) + 1e-6
