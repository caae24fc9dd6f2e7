"""Boundary / data-parallel / robustness checks of the HIP path that need the GPU:

  * TextPredNet's training branch returns the reference's `hidden_states` (decoder.py:115,126) — fixture from the reference;
  * the module wrapped in torch.nn.parallel.DistributedDataParallel (what Lightning's DDPStrategy does, train.py:45) trains
    step for step like the unwrapped module, with FlatAdamW built before or after the wrap;
  * the flat-gradient "direct" mode (backward kernels add into the flat buffer) equals autograd's accumulation bit for bit;
  * FlatAdamW save -> load -> step equals torch.optim.AdamW (resume);
  * a raised LSTM status word skips the update on device and makes the next optimizer.step() raise;
  * validation (torch.no_grad) launches no gradient kernel;
  * the ctypes stub printed in INTEGRATION.md §B, executed verbatim, against the oracle.
"""
import ctypes
import os
import re
import socket
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(**kw):
    return Namespace(learning_rate=1e-3, weight_decay=1e-4, warmup_ratio=0.2, final_div_factor=1e4, total_steps=100,
                     move_metrics_to_cpu=False, **kw)


def _small_model(args, seed=0, dropout=0.1):
    from rnntransducer_amd import RNNTransducer
    torch.manual_seed(seed)
    tn = dict(input_size=80, hidden_size=128, output_size=64, num_layers=2, dropout=dropout, bidirectional=True)
    pn = dict(embedding_size=30, hidden_size=64, output_size=64, num_layers=2, dropout=dropout)
    return RNNTransducer(pn, tn, dict(num_classes=30), args).cuda().train()


def _batch(seed=5, B=4, T=60, U=9, V=30):
    from rnntransducer_amd.data import synthetic_batch
    return synthetic_batch(B, T, U, V, ragged=True, seed=seed, device="cuda")


# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["h1_lstm_hidden", "h2_gru_hidden"])
def test_prednet_training_branch_returns_reference_hidden_states(golden_dir, tag):
    from rnntransducer_amd.networks import TextPredNet
    from tests.test_oracle_networks import HIDDEN
    g = dict(np.load(os.path.join(golden_dir, tag + ".npz")))
    net = TextPredNet(**HIDDEN[tag])
    net.load_state_dict({k[6:]: torch.from_numpy(v).float() for k, v in g.items() if k.startswith("param/")})
    net = net.cuda().eval()
    out, hidden = net(torch.from_numpy(g["tokens"]).cuda(), g["lens"].tolist())
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() < 2e-5
    if "c_n" in g:
        assert isinstance(hidden, tuple) and hidden[0].shape == g["h_n"].shape
        assert np.abs(hidden[0].cpu().numpy() - g["h_n"]).max() < 2e-5
        assert np.abs(hidden[1].cpu().numpy() - g["c_n"]).max() < 2e-5
    else:
        assert np.abs(hidden.cpu().numpy() - g["h_n"]).max() < 2e-5


# --------------------------------------------------------------------------------------------------------------------
class _StepWrapper(torch.nn.Module):
    """What Lightning's DDP strategy wraps: a module whose forward() is the LightningModule's training_step."""

    def __init__(self, lm):
        super().__init__()
        self.lm = lm

    def forward(self, *batch):
        return self.lm.training_step(tuple(batch), 0)["loss"]


def _train(model, stepper, conf, batch, n):
    opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    losses = []
    for _ in range(n):
        opt.zero_grad()
        loss = stepper(*batch)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.item())
    return losses, [p.detach().clone() for p in model.parameters()]


@pytest.mark.parametrize("optimizer_first", [True, False])
def test_ddp_world1_wrap_trains_like_the_unwrapped_module(optimizer_first):
    """train.py:45: Lightning wraps the LightningModule in DistributedDataParallel.  Here: world 1 (one GPU), RCCL backend;
    FlatAdamW re-points p.data / p.grad into its flat buffers before or after the reducer was built; autograd-accumulated
    gradients (direct_flat_grads off, as configure_optimizers picks under a trainer) so DDP's hooks fire."""
    import torch.distributed as dist
    batch = _batch()
    ref_model = _small_model(_args(direct_flat_grads=False), seed=3)
    ref_losses, ref_params = _train(ref_model, _StepWrapper(ref_model), ref_model.configure_optimizers(), batch, 3)

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        model = _small_model(_args(direct_flat_grads=False), seed=3)
        conf = model.configure_optimizers() if optimizer_first else None
        ddp = torch.nn.parallel.DistributedDataParallel(_StepWrapper(model), device_ids=[0])
        if conf is None:
            conf = model.configure_optimizers()
        losses, params = _train(model, ddp, conf, batch, 3)
        assert conf["optimizer"].flat.views_in_place()
    finally:
        dist.destroy_process_group()
    assert losses == ref_losses
    assert all(torch.equal(a, b) for a, b in zip(params, ref_params))


# --------------------------------------------------------------------------------------------------------------------
def test_direct_flat_gradients_equal_autograd_accumulation_bitwise():
    """direct mode: dW GEMMs / bias sums / embedding scatter ADD into the flat-gradient views and return None to autograd;
    default mode: they return tensors that autograd `+=`s into the same views.  Same bits, also over two accumulated
    micro-batches (scripts/run_train.sh:22 accumulates 16)."""
    b1, b2 = _batch(seed=5), _batch(seed=6)
    grads = {}
    for direct in (False, True):
        model = _small_model(_args(direct_flat_grads=direct), seed=1)
        opt = model.configure_optimizers()["optimizer"]
        assert opt.flat.direct_grads is direct
        opt.zero_grad()
        for b in (b1, b2):
            model.training_step(b, 0)["loss"].backward()
        assert opt.flat.views_in_place()
        grads[direct] = opt.flat_grad.clone()
        assert all(p.grad is not None and p.grad.abs().sum() > 0 for n, p in model.named_parameters())
    assert torch.equal(grads[False], grads[True])


def test_configure_optimizers_picks_direct_gradients_only_without_a_trainer():
    model = _small_model(_args(), seed=1)
    assert model.configure_optimizers()["optimizer"].flat.direct_grads is True
    model._trainer = Namespace(estimated_stepping_batches=50)
    assert model.configure_optimizers()["optimizer"].flat.direct_grads is False


# --------------------------------------------------------------------------------------------------------------------
def test_flat_adamw_resume_from_state_dict_matches_torch_adamw():
    """load_state_dict must land in the flat moments and restore the step count."""
    from rnntransducer_amd.optim import FlatAdamW
    torch.manual_seed(0)
    shapes = [(7, 5), (13,), (4, 3, 2), (130, 70)]
    ref_p = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    hip_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    ref, hip = torch.optim.AdamW(ref_p, lr=1e-2, weight_decay=1e-2), FlatAdamW(hip_p, lr=1e-2, weight_decay=1e-2)

    def feed(it, ps_ref, ps_hip, o_ref, o_hip):
        g = torch.Generator().manual_seed(100 + it)
        o_hip.zero_grad()
        for a, b in zip(ps_ref, ps_hip):
            gr = torch.randn(a.shape, generator=g)
            a.grad = gr.clone()
            b.grad += gr.cuda()
        o_ref.step(); o_hip.step()

    for it in range(3):
        feed(it, ref_p, hip_p, ref, hip)
    sd = hip.state_dict()
    sd_cpu = {"state": {k: {n: (t.cpu().clone() if isinstance(t, torch.Tensor) else t) for n, t in st.items()} for k, st in sd["state"].items()},
              "param_groups": sd["param_groups"]}
    # a fresh process: new parameters (same values), new optimizer, state loaded from the dict
    hip_p2 = [torch.nn.Parameter(p.detach().clone()) for p in hip_p]
    hip2 = FlatAdamW(hip_p2, lr=1e-2, weight_decay=1e-2)
    hip2.load_state_dict(sd_cpu)
    assert hip2._steps == 3 and hip2.flat_m.abs().sum() > 0
    assert hip2.state[hip_p2[0]]["exp_avg"].data_ptr() == hip2.flat_m.data_ptr()
    for it in range(3, 6):
        feed(it, ref_p, hip_p2, ref, hip2)
    for a, b in zip(ref_p, hip_p2):
        assert (a.detach() - b.detach().cpu()).abs().max().item() < 2e-6


def test_flat_adamw_readopts_gradients_after_set_to_none():
    """nn.Module.zero_grad() defaults to set_to_none=True: the next backward allocates fresh .grad tensors outside the flat
    buffer.  step() must pick them up (round 1 silently stepped on an all-zero flat buffer)."""
    model = _small_model(_args(direct_flat_grads=False), seed=2, dropout=0.0)
    twin = _small_model(_args(direct_flat_grads=False), seed=2, dropout=0.0)
    batch = _batch()
    for m, detach in ((model, True), (twin, False)):
        opt = m.configure_optimizers()["optimizer"]
        if detach:
            m.zero_grad(set_to_none=True)          # what a generic training loop does
            assert not opt.flat.views_in_place()
        else:
            opt.zero_grad()
        m.training_step(batch, 0)["loss"].backward()
        opt.step()
        assert opt.flat.views_in_place()
    assert all(torch.equal(a, b) for a, b in zip(model.parameters(), twin.parameters()))


# --------------------------------------------------------------------------------------------------------------------
def test_raised_lstm_status_word_skips_the_update_and_raises_at_the_next_step():
    """The sticky status word (rnnt_lstm_desc.status) pre-set by hand — NOT by provoking a stall: every persistent kernel
    bails out at its first inter-workgroup wait, the guarded AdamW kernel leaves the parameters alone, and the optimizer
    raises RnntHipError when it reads the word back (asynchronously, one step later at most)."""
    from rnntransducer_amd._lib import RnntHipError
    from rnntransducer_amd.ops import lstm_status_check, lstm_status_word
    model = _small_model(_args(), seed=4)
    opt = model.configure_optimizers()["optimizer"]
    batch = _batch()
    opt.zero_grad()
    model.training_step(batch, 0)["loss"].backward()
    opt.step()                                         # a healthy step first
    before = [p.detach().clone() for p in model.parameters()]
    word = lstm_status_word("cuda")
    assert int(word[0].item()) == 0
    word[0] = 1
    try:
        opt.zero_grad()
        model.training_step(batch, 0)["loss"].backward()   # garbage in, but it must come back (no hang)
        opt.step()                                          # update skipped on device; read-back enqueued
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
        assert opt._steps == 2                              # the host counted the skipped update ...
        with pytest.raises(RnntHipError, match="abandoned an inter-workgroup wait"):
            opt.zero_grad()
            model.training_step(batch, 0)["loss"].backward()
            opt.step()
        # ... and rewinds when the read-back shows the device skipped it: bias correction and state_dict's step count
        # the updates actually applied
        assert opt._steps == 1 and all(float(opt.state[p]["step"]) == 1.0 for p in opt.flat.params)
        assert opt.state_dict()["state"][0]["step"] == 1.0
        assert int(word[0].item()) == 0                     # cleared by the raise: the caller may go on
        word[0] = 1
        with pytest.raises(RnntHipError):
            lstm_status_check()
    finally:
        word.zero_()
    opt.zero_grad()
    loss = model.training_step(batch, 0)["loss"]
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and not all(torch.equal(a, b) for a, b in zip(before, model.parameters()))


def test_validation_step_launches_no_gradient_kernel():
    from rnntransducer_amd import _lib
    model = _small_model(_args(), seed=1)
    batch = _batch()
    L = _lib.lib()
    torch.cuda.synchronize()
    L.rnnt_hip_prof_enable(1)
    out = model.validation_step(batch, 0)
    torch.cuda.synchronize()
    L.rnnt_hip_prof_enable(0)
    nk = len(_lib.KERNEL_KINDS)
    ms, work, cnt = (ctypes.c_double * nk)(), (ctypes.c_double * nk)(), (ctypes.c_int64 * nk)()
    _lib.check(L.rnnt_hip_prof_collect(ms, work, cnt, nk), "prof_collect")
    kinds = dict(zip(_lib.KERNEL_KINDS, cnt))
    assert kinds["lattice_grad_kernel"] == 0 and kinds["lstm_bwd_kernel"] == 0
    assert kinds["alphabeta_kernel"] == 1 and kinds["lse_kernel"] == 1
    assert torch.isfinite(out["loss"])


# --------------------------------------------------------------------------------------------------------------------
def test_integration_md_ctypes_stub_runs_verbatim_and_matches_the_oracle():
    """INTEGRATION.md §B prints the stub a maintainer would add next to model.py.  Execute exactly that text."""
    from oracle.rnnt_oracle import rnnt_loss_c
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sect = text[text.index("## B."):]
    code = re.search(r"```python\n(.*?)```", sect, re.S).group(1)
    assert "rnnt_hip_loss_from_logits_fwd_bwd" in code
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)                                   # the stub opens the library by its repo-relative path
    try:
        exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(0)
    B, T, U, V = 3, 21, 6, 17
    z = rng.normal(size=(B, T, U + 1, V)).astype(np.float32)
    y = rng.integers(1, V, size=(B, U)).astype(np.int32)
    t_lens, u_lens = np.array([21, 13, 8], np.int32), np.array([6, 2, 5], np.int32)
    ref_nll, ref_grad = rnnt_loss_c(z.astype(np.float64), y, t_lens, u_lens, 0)
    loss, grad = ns["rnnt_loss_from_logits"](torch.from_numpy(z).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(t_lens).cuda(),
                                             torch.from_numpy(u_lens).cuda(), 0)
    assert abs(loss.item() - ref_nll.mean()) / ref_nll.mean() < 1e-5
    assert np.abs(grad.cpu().numpy() - ref_grad / B).max() < 2e-5


def test_cpu_build_of_the_same_abi_agrees_with_the_hip_library():
    """the oracle's .so exports `rnnt_hip_loss_from_logits_fwd_bwd` with the SAME argument list (host pointers); both
    libraries get identical arguments through raw ctypes and must agree."""
    from oracle import build_oracle
    from rnntransducer_amd import _lib
    cpu = ctypes.CDLL(build_oracle.build())
    hip = ctypes.CDLL(_lib.LIB_PATH)
    hip.rnnt_hip_joint_loss_workspace_bytes.restype = ctypes.c_size_t
    rng = np.random.default_rng(3)
    B, T, U, V = 3, 17, 5, 23
    z = rng.normal(size=(B, T, U + 1, V)).astype(np.float32)
    y = rng.integers(1, V, size=(B, U)).astype(np.int32)
    t_lens, u_lens = np.array([17, 9, 4], np.int32), np.array([5, 3, 0], np.int32)
    nws = hip.rnnt_hip_joint_loss_workspace_bytes(B, T, U + 1, V)

    def call(lib, to_arg, out_nll, out_grad, ws, stream):
        p = ctypes.c_void_p
        rc = lib.rnnt_hip_loss_from_logits_fwd_bwd(p(to_arg(z)), p(to_arg(y)), p(to_arg(t_lens)), p(to_arg(u_lens)), B, T, U + 1, V, 0,
                                                   ctypes.c_float(0.5), p(out_nll), p(out_grad), p(ws), ctypes.c_size_t(nws), p(stream))
        assert rc == 0
    nll_c, grad_c = np.empty(B, np.float32), np.empty_like(z)
    call(cpu, lambda a: a.ctypes.data, nll_c.ctypes.data, grad_c.ctypes.data, None, None)
    dev = {id(a): torch.from_numpy(a).cuda() for a in (z, y, t_lens, u_lens)}
    nll_g, grad_g = torch.empty(B, device="cuda"), torch.empty(B, T, U + 1, V, device="cuda")
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    call(hip, lambda a: dev[id(a)].data_ptr(), nll_g.data_ptr(), grad_g.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_allclose(nll_g.cpu().numpy(), nll_c, rtol=1e-5)
    assert np.abs(grad_g.cpu().numpy() - grad_c).max() < 2e-5
    # the same bad-argument convention on both sides
    assert cpu.rnnt_hip_loss_from_logits_fwd_bwd(None, None, None, None, 1, 1, 1, 3, 0, ctypes.c_float(1.0), None, None, None, 0, None) == -1
    assert hip.rnnt_hip_loss_from_logits_fwd_bwd(None, None, None, None, 1, 1, 1, 3, 0, ctypes.c_float(1.0), None, None, None, 0, None) == -1


def test_training_step_under_fp16_autocast_and_grad_scaler_is_the_fp32_step():
    """scripts/run_train.sh:32 launches the reference with `--precision 16`: Lightning then wraps training_step in
    torch.autocast(float16) and drives a GradScaler (model.py:28-31 picks torchaudio's loss for that mode).  This module computes in
    fp32 whatever the autocast state says — its kernels are not torch ops autocast could down-cast, which is at least the reference's
    precision — so the drop-in must (a) run unchanged inside the autocast region, (b) survive the scaler's loss scaling / unscaling /
    inf check on the flat gradient views, and (c) take the same optimizer step as the plain fp32 loop."""
    batch = _batch(seed=8)
    models, opts = [], []
    for _ in range(2):
        m = _small_model(_args(), seed=6, dropout=0.0)
        models.append(m)
        opts.append(m.configure_optimizers()["optimizer"])
    plain, amp = models
    opts[0].zero_grad()
    l0 = plain.training_step(batch, 0)["loss"]
    l0.backward()
    opts[0].step()
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 14)
    opts[1].zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        l1 = amp.training_step(batch, 0)["loss"]
    assert l1.dtype == torch.float32 and torch.equal(l1, l0)
    scaler.scale(l1).backward()
    scaler.step(opts[1])          # unscales the .grad views in place, checks them for inf / nan, then calls FlatAdamW.step()
    scaler.update()
    torch.cuda.synchronize()
    assert scaler.get_scale() == 2.0 ** 14   # no overflow was found
    for (n, a), (_, b) in zip(plain.named_parameters(), amp.named_parameters()):
        # power-of-two loss scale: the scaled gradients are exact multiples, so the update is the same up to the unscale rounding
        assert torch.allclose(a, b, rtol=0, atol=2e-7), n
