"""torch.autograd wrappers over the C ABI (include/rnnt_hip.h).  PyTorch here is plumbing only: it owns
device memory and the stream and records the graph; every FLOP below runs in librnnt_hip.so.

All internal activations are TIME-MAJOR (T,B,F): one LSTM step touches one contiguous (B,F) slab.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import GEMM_ACCUM, GEMM_GELU_A, GEMM_GELU_B, GEMM_MUL_DGELU, GemmDesc, LstmBwdDesc, LstmDesc, RnntHipError, check

BIG = 1 << 40  # "no second level" divisor for the GEMM row maps


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RnntHipError("rnntransducer_amd runs on the MI355X only: got a CPU tensor. There is no CPU or "
                               "eager fallback; move the module and the batch to cuda:<n>.")


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _addr(t: Optional[torch.Tensor], elem_off: int = 0) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() + elem_off * t.element_size()


# --------------------------------------------------------------------------------------------------
# flat-gradient targets (optim.FlatAdamW(direct_grads=True) tags its parameters): backward kernels then ADD their weight
# gradients straight into the parameter's `.grad` view of the flat buffer and hand autograd `None`, instead of returning
# a fresh tensor that autograd would `+=` into the view with one extra kernel per parameter.
# --------------------------------------------------------------------------------------------------
def _direct_grad(param: torch.Tensor) -> Optional[torch.Tensor]:
    g = param.grad if getattr(param, "_rnnt_direct_grad", False) else None
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.shape != param.shape:
        return None
    return g


# --------------------------------------------------------------------------------------------------
# sticky status word of the persistent recurrences (include/rnnt_hip.h: rnnt_lstm_desc.status), one per device.
# Kernels raise it when an inter-workgroup wait is abandoned; nothing in the library clears it.  FlatAdamW hands it to
# the update kernel as a guard and reads it back asynchronously once per step; `lstm_status_check` is the explicit read.
# --------------------------------------------------------------------------------------------------
_STATUS = {}


def lstm_status_word(device) -> torch.Tensor:
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    w = _STATUS.get(idx)
    if w is None:
        w = torch.zeros(4, dtype=torch.int32, device=torch.device("cuda", idx))
        _STATUS[idx] = w
    return w


def lstm_status_check(device=None) -> None:
    """Synchronous read of the device's sticky status word; raises RnntHipError if a persistent LSTM kernel gave up on an
    inter-workgroup wait (its outputs, and every later LSTM launch on this device, are invalid).  Clears the word so a
    caller that handles the exception can go on."""
    for idx, w in list(_STATUS.items()):
        if device is not None and torch.device(device).index not in (None, idx):
            continue
        if int(w[0].item()) != 0:
            w.zero_()
            raise RnntHipError("a persistent LSTM kernel abandoned an inter-workgroup wait (its workgroups were not "
                               "co-resident for 4 s — is another kernel holding the CUs?): activations and gradients "
                               "of that step are invalid; the guarded AdamW update was skipped")


# --------------------------------------------------------------------------------------------------
# raw GEMM
# --------------------------------------------------------------------------------------------------
def gemm(M: int, N: int, K: int, A: torch.Tensor, B: torch.Tensor, Cout: torch.Tensor, *, a_off=0, a_div=BIG, a_so=0,
         a_si=None, a_sk=1, a_mc=False, a_rowidx=None, b_off=0, b_sn=None, b_sk=1, c_off=0, c_div=BIG, c_so=0, c_si=None,
         bias=None, aux=None, flags=0, split_k=None) -> None:
    """C(m,n) = sum_k A(m,k) B(k,n) (+bias) — see include/rnnt_hip.h for the operand maps.
    split_k=True hands the kernel a slab workspace so small-output / deep-K products (weight gradients) fill the chip;
    None (default): do so when the output is small (<= 2^21 elements: the prediction net's and the joint's products)."""
    _need_gpu(A, B, Cout)
    d = GemmDesc()
    ws = None
    if split_k is None:
        split_k = M * N <= (1 << 21)
    if split_k:
        nws = _lib.lib().rnnt_hip_gemm_workspace_bytes(M, N, K)
        if nws:
            ws = torch.empty(nws, device=Cout.device, dtype=torch.uint8)
            d.workspace, d.workspace_bytes = _addr(ws), nws
    d.M, d.N, d.K = M, N, K
    d.A = _addr(A, a_off)
    d.a_div, d.a_so, d.a_si, d.a_sk = a_div, a_so, (K if a_si is None else a_si), a_sk
    d.a_mc = 1 if a_mc else 0
    d.a_rowidx = _addr(a_rowidx)
    d.B = _addr(B, b_off)
    d.b_sn, d.b_sk = (K if b_sn is None else b_sn), b_sk
    d.C = _addr(Cout, c_off)
    d.c_div, d.c_so, d.c_si = c_div, c_so, (N if c_si is None else c_si)
    d.bias = _addr(bias)
    d.aux = _addr(aux)
    d.flags = flags
    check(_lib.lib().rnnt_hip_gemm_f32(C.byref(d), _stream()), "rnnt_hip_gemm_f32")


# --------------------------------------------------------------------------------------------------
# half-pair (hp) operands + the f16-MFMA GEMM on them (include/rnnt_hip.h).  The LSTM entry points use these internally for
# their big products; exposed here for tests and tools.
# --------------------------------------------------------------------------------------------------
class HpTensor:
    """hp planes of an fp32 matrix (rows x K): device byte buffer + the per-row amax words the row scales derive from."""

    def __init__(self, rows: int, K: int, device):
        self.rows, self.K = int(rows), int(K)
        n = _lib.lib().rnnt_hip_hp_bytes(self.rows, self.K)
        self.planes = torch.empty(max(n, 128), device=device, dtype=torch.uint8)
        self.amax = torch.empty(max(self.rows, 1), device=device, dtype=torch.int32)   # per-row maxima (fp32 bit patterns), written by the split


def hp_split(x: torch.Tensor, transpose: bool = False, shift: int = 0, K: Optional[int] = None) -> HpTensor:
    """x (rows, K) fp32 -> HpTensor(rows, K).  transpose=True: x is (src_rows, rows); plane row r, index k = x[k + shift, r]
    (zero outside the source), contraction length K (default src_rows)."""
    _need_gpu(x)
    x = _f32c(x, "x")
    if x.dim() != 2:
        raise ValueError("hp_split takes a 2-D tensor")
    if not transpose:
        rows, kk = x.shape
        t = HpTensor(rows, kk, x.device)
        check(_lib.lib().rnnt_hip_hp_split(_addr(x), rows, kk, kk, 0, 0, 0, _addr(t.planes), _addr(t.amax), 0, _stream()), "hp_split")
        return t
    src_rows, rows = x.shape
    kk = src_rows if K is None else int(K)
    t = HpTensor(rows, kk, x.device)
    check(_lib.lib().rnnt_hip_hp_split(_addr(x), rows, kk, rows, 1, src_rows, int(shift), _addr(t.planes), _addr(t.amax), 0, _stream()),
          "hp_split")
    return t


def gemm_hp(a: HpTensor, b: HpTensor, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
            accumulate: bool = False, split_k: bool = True) -> torch.Tensor:
    """C (M, N) [+]= A (M, K) . B (N, K)^T + bias on hp operands."""
    if a.K != b.K:
        raise ValueError(f"contraction lengths differ: {a.K} vs {b.K}")
    M, N, K = a.rows, b.rows, a.K
    if out is None:
        out = torch.empty(M, N, device=a.planes.device, dtype=torch.float32)
    if tuple(out.shape) != (M, N) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError(f"out must be a contiguous float32 ({M}, {N}) tensor")
    if bias is not None and tuple(bias.shape) != (N,):
        raise ValueError(f"bias must be ({N},)")
    nws = _lib.lib().rnnt_hip_gemm_hp_workspace_bytes(M, N, K) if split_k else 0
    ws = torch.empty(nws, device=out.device, dtype=torch.uint8) if nws else None
    check(_lib.lib().rnnt_hip_gemm_hp(_addr(a.planes), _addr(a.amax), _addr(b.planes), _addr(b.amax), M, N, K, _addr(out), N, _addr(bias),
                                      GEMM_ACCUM if accumulate else 0, _addr(ws), nws, _stream()), "rnnt_hip_gemm_hp")
    return out


def gemm_hp_grouped(pairs, outs=None, accumulate: bool = False, xcd_skip: int = 0, check: bool = False):
    """[C_i (M_i, N_i) [+]= A_i . B_i^T] for up to 4 (A, B) pairs of hp operands in ONE queue-driven launch; `xcd_skip`: bit mask of
    XCDs whose workgroups leave at once (the launch then runs on the other XCDs only).  `check`: read back (synchronising) the
    launch's self-check — word 9 of the workspace is 1 when units were left undone because the mask named XCDs the device does not
    expose (include/rnnt_hip.h) — and raise RnntHipError in that case."""
    n = len(pairs)
    if not 1 <= n <= 4:
        raise ValueError("1..4 products per grouped launch")
    pr = (_lib.HpProblem * n)()
    res = []
    for i, (a, b) in enumerate(pairs):
        if a.K != b.K:
            raise ValueError(f"contraction lengths differ: {a.K} vs {b.K}")
        out = outs[i] if outs is not None else torch.empty(a.rows, b.rows, device=a.planes.device, dtype=torch.float32)
        if tuple(out.shape) != (a.rows, b.rows) or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError(f"out[{i}] must be a contiguous float32 ({a.rows}, {b.rows}) tensor")
        pr[i].A, pr[i].a_amax, pr[i].B, pr[i].b_amax = _addr(a.planes), _addr(a.amax), _addr(b.planes), _addr(b.amax)
        pr[i].M, pr[i].N, pr[i].K, pr[i].C, pr[i].ldc = a.rows, b.rows, a.K, _addr(out), b.rows
        pr[i].flags = GEMM_ACCUM if accumulate else 0
        res.append(out)
    nws = _lib.lib().rnnt_hip_gemm_hp_grouped_workspace_bytes(pr, n)
    ws = torch.empty(nws, device=res[0].device, dtype=torch.uint8)
    _lib.check(_lib.lib().rnnt_hip_gemm_hp_grouped(pr, n, int(xcd_skip), _addr(ws), nws, _stream()), "rnnt_hip_gemm_hp_grouped")
    if check:
        words = ws[:64].view(torch.int32).tolist()
        if words[9] != 0:
            raise RnntHipError(f"grouped half-pair GEMM left units undone ({words[8]} drawn): xcd_skip = {xcd_skip:#x} names XCDs this "
                               "device does not expose")
    return res


def colsum(X: torch.Tensor, M: int, N: int, ld: Optional[int] = None, into: Optional[torch.Tensor] = None):
    """Column sums of X (M,N).  `into`: add them to this (N,) tensor (a flat-gradient view) and return None."""
    out = torch.empty(N, device=X.device, dtype=torch.float32) if into is None else into
    nws = _lib.lib().rnnt_hip_colsum_workspace_bytes(M, N)
    ws = torch.empty(max(nws, 16), device=X.device, dtype=torch.uint8)
    fn = _lib.lib().rnnt_hip_colsum_f32 if into is None else _lib.lib().rnnt_hip_colsum_f32_acc
    check(fn(_addr(X), M, N, N if ld is None else ld, _addr(out), _addr(ws), nws, _stream()), "colsum")
    return out if into is None else None


# --------------------------------------------------------------------------------------------------
# Linear: y = x W^T + b on rows of a 2-D view (replaces nn.Linear: encoder.py:76,103; decoder.py:80,124)
# --------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, bias):
        _need_gpu(x, W, bias)
        x = _f32c(x, "x")
        W0 = _f32c(W, "weight")
        lead, K = x.shape[:-1], x.shape[-1]
        N = W.shape[0]
        if W.dim() != 2 or W.shape[1] != K or (bias is not None and tuple(bias.shape) != (N,)):
            raise ValueError(f"linear: x (..., {K}) needs weight (N, {K}) and bias (N,); got weight {tuple(W.shape)}, "
                             f"bias {None if bias is None else tuple(bias.shape)}")
        M = x.numel() // K
        y = torch.empty(*lead, N, device=x.device, dtype=torch.float32)
        # big products (the encoder's out_proj: 32000 x 512 x 1024 at config 2) take the half-pair f16 path of the LSTM layers'
        # products (include/rnnt_hip.h: same fp32-grade arithmetic, 440 instead of 140-150 TFLOP/s); its operand splits only pay
        # for themselves on deep, wide shapes
        # (an operand whose planes reach 4 GB is beyond gemm_hp.hip's 32-bit buffer offsets: such a product stays on gemm.hip)
        hpb = _lib.lib().rnnt_hip_hp_bytes
        ctx.hp = (M >= 1024 and N >= 256 and K >= 1024 and not os.environ.get("RNNT_GEMM_NO_HP")
                  and max(hpb(M, K), hpb(K, M), hpb(M, N), hpb(N, M)) < (1 << 32))
        if ctx.hp:
            gemm_hp(hp_split(x.view(M, K)), hp_split(W0), out=y.view(M, N), bias=bias)
        else:
            gemm(M, N, K, x, W0, y, bias=bias)
        ctx.save_for_backward(x, W0)
        ctx.has_bias = bias is not None
        ctx.w_param, ctx.b_param = W, bias  # the Parameter objects themselves (for their flat .grad views)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        N, K = W.shape
        M = x.numel() // K
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if ctx.hp:
                gemm_hp(hp_split(dy.view(M, N)), hp_split(W, transpose=True), out=dx.view(M, K))   # dx = dy . W
            else:
                gemm(M, K, N, dy, W, dx, b_sn=1, b_sk=K)          # dx = dy . W
        if ctx.needs_input_grad[1]:
            tgt = _direct_grad(ctx.w_param)
            dW = torch.empty_like(W) if tgt is None else None
            if ctx.hp:   # dW = dy^T . x: both operands row-major over the contraction index M (transposed splits), split-K slabs
                gemm_hp(hp_split(dy.view(M, N), transpose=True), hp_split(x.view(M, K), transpose=True),
                        out=dW if tgt is None else tgt, accumulate=tgt is not None)
            else:
                gemm(N, K, M, dy, x, dW if tgt is None else tgt, a_mc=True, a_sk=N, b_sn=1, b_sk=K, split_k=True,
                     flags=0 if tgt is None else GEMM_ACCUM)  # dW = dy^T . x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy, M, N, into=_direct_grad(ctx.b_param))
        return dx, dW, db


# --------------------------------------------------------------------------------------------------
# Embedding (networks/decoder.py:69,102)
# --------------------------------------------------------------------------------------------------
class EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, idx, padding_idx):
        _need_gpu(W, idx)
        W_param = W
        W = _f32c(W, "embedding.weight")
        if idx.dtype != torch.int64:
            raise ValueError(f"token ids must be int64 (dataloader.py:28-36), got {idx.dtype}")
        idx = idx.contiguous()
        V, H = W.shape
        out = torch.empty(*idx.shape, H, device=W.device, dtype=torch.float32)
        check(_lib.lib().rnnt_hip_embedding_fwd(_addr(W), _addr(idx), idx.numel(), H, V, _addr(out), _stream()), "embedding_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (V, H)
        ctx.padding_idx = -1 if padding_idx is None else int(padding_idx)
        ctx.w_param = W_param
        return out

    @staticmethod
    def backward(ctx, dE):
        (idx,) = ctx.saved_tensors
        V, H = ctx.shape
        dE = _f32c(dE, "dE")
        tgt = _direct_grad(ctx.w_param)
        if tgt is not None:
            check(_lib.lib().rnnt_hip_embedding_bwd_acc(_addr(dE), _addr(idx), idx.numel(), H, V, ctx.padding_idx, _addr(tgt),
                                                        _stream()), "embedding_bwd_acc")
            return None, None, None
        dW = torch.zeros(V, H, device=dE.device, dtype=torch.float32)
        check(_lib.lib().rnnt_hip_embedding_bwd(_addr(dE), _addr(idx), idx.numel(), H, V, ctx.padding_idx, _addr(dW),
                                                _stream()), "embedding_bwd")
        return dW, None, None


# --------------------------------------------------------------------------------------------------
# LSTM stack (replaces nn.LSTM over a PackedSequence: encoder.py:67-75,93-102; decoder.py:71-79,105-120)
# --------------------------------------------------------------------------------------------------
class RaggedPlan:
    """Valid-frame table of a padded time-major batch (rnnt_lstm_desc.row_idx): what pack_padded_sequence buys the reference
    (networks/encoder.py:93-96,99-101) without a packed copy.  Built on the HOST from the python list of lengths the reference's
    collate hands over (dataloader.py:20) — no device synchronisation — and uploaded once per batch: `row_idx` lists the rows
    t*B + b with t < lens[b] in ascending order, `n_rows` = sum(lens).  Pass it to HipLSTM / LstmStackFn in place of `lens`."""

    def __init__(self, lens_host: Sequence[int], T: int, device):
        import numpy as np
        lens_np = np.asarray(list(lens_host), dtype=np.int64)
        if lens_np.ndim != 1 or lens_np.size < 1 or lens_np.min() < 1 or lens_np.max() > T:
            raise ValueError(f"lengths must lie in [1, {T}]")
        self.T, self.B = int(T), int(lens_np.size)
        idx = np.flatnonzero((np.arange(T, dtype=np.int64)[:, None] < lens_np[None, :]).reshape(-1)).astype(np.int32)
        self.n_rows = int(idx.size)
        self.dense = self.n_rows == self.T * self.B
        self.lens = torch.from_numpy(lens_np.astype(np.int32)).to(device, non_blocking=True)
        self.row_idx = None if self.dense else torch.from_numpy(idx).to(device, non_blocking=True)


def lstm_workspace(T: int, B: int, I: int, H: int, D: int, device) -> torch.Tensor:
    n = _lib.lib().rnnt_hip_lstm_workspace_bytes(T, B, I, H, D)
    if n == 0:
        raise RnntHipError(f"LSTM configuration not supported by the HIP kernels: T={T} B={B} I={I} H={H} D={D} "
                           "(need H % 4 == 0, 1 <= B <= 64, D in {1,2})")
    return torch.empty(n, device=device, dtype=torch.uint8)


def _fill_lstm_desc(d: LstmDesc, T, B, I, H, D, lens, x, weights, y, y_drop, p, seed, gates, cst, ws, cell=0, aux=None, plan=None) -> None:
    d.T, d.B, d.I, d.H, d.D = T, B, I, H, D
    if plan is not None and plan.row_idx is not None:
        d.row_idx, d.n_rows = _addr(plan.row_idx), plan.n_rows
    d.cell = cell
    d.aux = _addr(aux)
    d.lens = _addr(lens)
    d.x = _addr(x)
    d.x_st, d.x_sb = B * I, I
    for k in range(D):
        w_ih, w_hh, b_ih, b_hh = weights[4 * k:4 * k + 4]
        d.w_ih[k], d.w_hh[k], d.b_ih[k], d.b_hh[k] = _addr(w_ih), _addr(w_hh), _addr(b_ih), _addr(b_hh)
    d.y = _addr(y)
    d.y_drop = _addr(y_drop)
    d.dropout_p = float(p)
    d.dropout_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    d.gates = _addr(gates)
    d.cst = _addr(cst)
    d.workspace = _addr(ws)
    d.workspace_bytes = ws.numel()
    d.status = _addr(lstm_status_word(ws.device))


class LstmStackFn(torch.autograd.Function):
    """x (T,B,I) time-major, lens (B) int32 on device -> y (T,B,D*H); zero rows for t >= lens[b].
    `cell`: 0 LSTM, 1 GRU, 2 tanh-RNN, 3 ReLU-RNN (the reference's supported_rnns, encoder.py:48-52).
    `want_final`: also return (h_n, c_n) — (L*D, B, H) states after each sequence's own last step, what torch's RNN modules
    return next to the output (decoder.py:115); not differentiable here (the reference never differentiates through them)."""

    @staticmethod
    def forward(ctx, x, lens, hidden, num_layers, bidirectional, dropout_p, seed, cell, want_final, *weights):
        plan = None
        if isinstance(lens, RaggedPlan):   # ragged batch with its valid-frame table: the big products and the recurrences skip padding
            plan, lens = lens, lens.lens
            if (plan.T, plan.B) != tuple(x.shape[:2]):
                raise ValueError(f"RaggedPlan was built for (T,B) = ({plan.T},{plan.B}), x is {tuple(x.shape)}")
            if plan.dense:
                plan = None
        _need_gpu(x, lens, *weights)
        if lens.dtype != torch.int32:
            raise ValueError(f"lengths must be int32 (dataloader.py:23-24), got {lens.dtype}")
        x = _f32c(x, "x")
        if x.dim() != 3 or lens.shape != (x.shape[1],):
            raise ValueError(f"x must be (T,B,I) with lens (B,): got {tuple(x.shape)} and {tuple(lens.shape)}")
        T, B, I0 = x.shape
        D = 2 if bidirectional else 1
        H = hidden
        # the kernels index raw pointers: every weight's shape is checked here, on the host, before anything is launched
        ngate = {0: 4, 1: 3, 2: 1, 3: 1}.get(cell)
        if ngate is None or not isinstance(want_final, bool):
            raise ValueError(f"cell must be 0..3 and want_final a bool (got cell={cell!r}, want_final={type(want_final).__name__})")
        if len(weights) != 4 * D * num_layers:
            raise ValueError(f"expected {4 * D * num_layers} weight tensors (w_ih, w_hh, b_ih, b_hh per layer and direction), got {len(weights)}")
        for i, w in enumerate(weights):
            layer, kind = i // (4 * D), i % 4
            I_l = I0 if layer == 0 else D * H
            want = [(ngate * H, I_l), (ngate * H, H), (ngate * H,), (ngate * H,)][kind]
            if not isinstance(w, torch.Tensor) or tuple(w.shape) != want:
                raise ValueError(f"lstm weight #{i} (layer {layer}, {['w_ih', 'w_hh', 'b_ih', 'b_hh'][kind]}): expected shape {want}, "
                                 f"got {tuple(w.shape) if isinstance(w, torch.Tensor) else type(w).__name__}")
        params = weights
        weights = [_f32c(w, "lstm weight") for w in weights]
        dev = x.device
        # one workspace for all layers: the larger of what the first (input width I0) and the inner layers (D*H) ask for
        ws = lstm_workspace(T, B, I0, H, D, dev)
        if num_layers > 1 and _lib.lib().rnnt_hip_lstm_workspace_bytes(T, B, D * H, H, D) > ws.numel():
            ws = lstm_workspace(T, B, D * H, H, D, dev)
        saved = []
        cur = x
        for layer in range(num_layers):
            I = cur.shape[-1]
            wl = weights[4 * D * layer:4 * D * (layer + 1)]
            gates = torch.empty(T, B, D * 4 * H, device=dev, dtype=torch.float32)
            cst = torch.empty(D * T * B * H, device=dev, dtype=torch.float32) if cell == 0 else None
            # (with a valid-frame table the recurrence leaves frames beyond a sync group's longest row untouched: they must read 0)
            y = (torch.zeros if plan is not None else torch.empty)(T, B, D * H, device=dev, dtype=torch.float32)
            p = dropout_p if layer < num_layers - 1 else 0.0
            y_drop = torch.empty_like(y) if p > 0 else None
            d = LstmDesc()
            _fill_lstm_desc(d, T, B, I, H, D, lens, cur, wl, y, y_drop, p, seed + layer, gates, cst, ws, cell, plan=plan)
            check(_lib.lib().rnnt_hip_lstm_fwd(C.byref(d), _stream()), "rnnt_hip_lstm_fwd")
            saved.append((cur, y, gates, cst, p))
            cur = y_drop if p > 0 else y
        ctx.meta = (T, B, H, D, num_layers, seed, cell)
        ctx.lens = lens
        ctx.plan = plan
        ctx.ws = ws
        ctx.saved = saved
        ctx.weights = weights
        ctx.params = params
        ctx.x_needs_grad = x.requires_grad
        if not want_final:
            return cur
        # final states: plumbing gathers from the stash (forward direction: frame lens-1, reverse direction: frame 0)
        last = (lens.long() - 1).clamp_(min=0)
        rows = torch.arange(B, device=dev)
        hs, cs = [], []
        for (_, y, _, cst, _) in saved:
            yv = y.view(T, B, D, H)
            cv = cst.view(D, T, H // 4, B, 4) if cst is not None else None
            for dd in range(D):
                tsel = last if dd == 0 else torch.zeros_like(last)
                hs.append(yv[tsel, rows, dd])
                if cv is not None:
                    cs.append(cv[dd, tsel, :, rows, :].reshape(B, H))
        h_n = torch.stack(hs)
        c_n = torch.stack(cs) if cs else torch.empty(0, device=dev)
        ctx.mark_non_differentiable(h_n, c_n)
        return cur, h_n, c_n

    @staticmethod
    def backward(ctx, dy, *_unused):
        T, B, H, D, L, seed, cell = ctx.meta
        dy = _f32c(dy, "dy")
        weights = ctx.weights
        grads: List[Optional[torch.Tensor]] = [None] * len(weights)
        targets = [_direct_grad(p) for p in ctx.params]
        direct = all(t is not None for t in targets)  # all-or-nothing per stack: one accumulate flag per launch
        # Only dx is on the chain to the layer below: the weight / bias gradients of layer l (phase 2 of rnnt_hip_lstm_bwd) go to a
        # second stream and run beside the reverse-time recurrence of layer l-1 (phase 1), which leaves most of the chip idle.
        # Two workspaces alternate by layer parity so that phase 1 of layer l-1 never touches what phase 2 of layer l still reads.
        # Worth it only where the recurrence leaves XCDs free (c3: 4 of 8, +3.9 %): when its groups fill the chip (c2: 8 groups x 32
        # workgroups) the second stream's kernels merely queue behind it (+0.1..0.7 %, and every per-kernel timing turns into a
        # shared-device duration).  RNNT_LSTM_OVERLAP=1 forces it (tests), RNNT_LSTM_NO_OVERLAP=1 forbids it.
        overlap = L >= 2 and not os.environ.get("RNNT_LSTM_NO_OVERLAP") and (
            bool(os.environ.get("RNNT_LSTM_OVERLAP")) or _lib.lib().rnnt_hip_lstm_free_xcds(T, B, H, D, cell) >= 4)
        main = torch.cuda.current_stream()
        side = _side_stream(dy.device) if overlap else None
        wss = [ctx.ws, torch.empty_like(ctx.ws)] if overlap else [ctx.ws, ctx.ws]
        side_done: dict = {}
        keep = []   # tensors phase 2 reads or writes on the side stream: held until the streams have joined
        dx = None
        for layer in range(L - 1, -1, -1):
            x_l, y_l, gates, cst, p = ctx.saved[layer]
            I = x_l.shape[-1]
            wl = weights[4 * D * layer:4 * D * (layer + 1)]
            bd = LstmBwdDesc()
            aux = torch.empty_like(gates) if cell == 1 else None
            _fill_lstm_desc(bd.f, T, B, I, H, D, ctx.lens, x_l, wl, y_l, y_l if p > 0 else None, p, seed + layer, gates,
                            cst, wss[layer % 2], cell, aux, plan=ctx.plan)
            if layer > 0 and cell != 3:   # x of this layer is the (dropped) output of a bounded cell: |x| <= 1 / (1 - p) of the layer below
                bd.f.x_abs_bound = 1.0 / (1.0 - ctx.saved[layer - 1][4])
            bd.dy = _addr(dy)
            need_dx = layer > 0 or ctx.x_needs_grad
            dx = torch.empty(T, B, I, device=dy.device, dtype=torch.float32) if need_dx else None
            bd.dx = _addr(dx)
            bd.accumulate = 1 if direct else 0
            for k in range(D):
                base = 4 * D * layer + 4 * k
                if direct:  # += straight into the flat-gradient views; b_hh's view is the second destination of db
                    bd.dw_ih[k], bd.dw_hh[k] = _addr(targets[base]), _addr(targets[base + 1])
                    bd.db[k], bd.db_hh[k] = _addr(targets[base + 2]), _addr(targets[base + 3])
                    continue
                dw_ih = torch.empty_like(wl[4 * k])
                dw_hh = torch.empty_like(wl[4 * k + 1])
                db = torch.empty_like(wl[4 * k + 2])
                db_hh = torch.empty_like(db) if cell == 1 else db  # GRU: b_hn sits inside r * (.), its gradient differs
                bd.dw_ih[k], bd.dw_hh[k], bd.db[k] = _addr(dw_ih), _addr(dw_hh), _addr(db)
                bd.db_hh[k] = _addr(db_hh) if cell == 1 else None
                grads[base], grads[base + 1], grads[base + 2], grads[base + 3] = dw_ih, dw_hh, db, db_hh
            if not overlap:
                check(_lib.lib().rnnt_hip_lstm_bwd(C.byref(bd), main.cuda_stream), "rnnt_hip_lstm_bwd")
                dy = dx
                continue
            if layer + 2 in side_done:   # this layer's workspace was last read by phase 2 of layer + 2
                main.wait_event(side_done[layer + 2])
            bd.phase = 1
            check(_lib.lib().rnnt_hip_lstm_bwd(C.byref(bd), main.cuda_stream), "rnnt_hip_lstm_bwd (recurrence + dx)")
            side.wait_event(main.record_event())
            bd.phase, bd.beside_recurrence = 2, 1 if layer > 0 else 0
            check(_lib.lib().rnnt_hip_lstm_bwd(C.byref(bd), side.cuda_stream), "rnnt_hip_lstm_bwd (weight gradients)")
            side_done[layer] = side.record_event()
            keep.append((aux, dy, dx))
            dy = dx
        if overlap:
            main.wait_stream(side)   # everything below (autograd's accumulation, the optimizer, frees) is ordered after phase 2
        del keep
        ctx.saved = None  # release the stash
        return (dx if ctx.x_needs_grad else None, None, None, None, None, None, None, None, None, *grads)


_SIDE_STREAMS: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    """One extra stream per device for work that overlaps the persistent recurrences (created once, reused)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


def lstm_check(ws: torch.Tensor) -> None:
    """C-ABI diagnostic for callers that pass status = NULL (the workspace's own per-launch word); the package itself uses
    the sticky device word: see lstm_status_check()."""
    check(_lib.lib().rnnt_hip_lstm_check(_addr(ws), _stream()), "rnnt_hip_lstm_check")


# --------------------------------------------------------------------------------------------------
# fused joint + RNN-T loss  (replaces transducer.py:54-69 + model.py:39,57)
# --------------------------------------------------------------------------------------------------
def _joint_ac(enc, dec, W, Oe, Od, V):
    """A (T,B,V) = gelu(enc) W[:, :Oe]^T ; C (U1,B,V) = gelu(dec) W[:, Oe:]^T  (two small GEMMs, GELU on load)."""
    T, B = enc.shape[:2]
    U1 = dec.shape[0]
    A = torch.empty(T, B, V, device=enc.device, dtype=torch.float32)
    Cm = torch.empty(U1, B, V, device=enc.device, dtype=torch.float32)
    gemm(T * B, V, Oe, enc, W, A, b_sn=Oe + Od, b_sk=1, flags=GEMM_GELU_A)
    gemm(U1 * B, V, Od, dec, W, Cm, b_off=Oe, b_sn=Oe + Od, b_sk=1, flags=GEMM_GELU_A)
    return A, Cm


def _joint_backward(enc, dec, W, dA, dC, needs, w_tgt=None, b_tgt=None):
    """d_enc, d_dec, dW, db from dA (T,B,V), dC (U1,B,V).  w_tgt / b_tgt: flat-gradient views to add into (then None is
    returned for that gradient)."""
    T, B, Oe = enc.shape
    U1, _, Od = dec.shape
    V = W.shape[0]
    O = Oe + Od
    d_enc = d_dec = dW = db = None
    if needs[0]:
        d_enc = torch.empty_like(enc)
        gemm(T * B, Oe, V, dA, W, d_enc, b_sn=1, b_sk=O, aux=enc, flags=GEMM_MUL_DGELU)
    if needs[1]:
        d_dec = torch.empty_like(dec)
        gemm(U1 * B, Od, V, dC, W, d_dec, b_off=Oe, b_sn=1, b_sk=O, aux=dec, flags=GEMM_MUL_DGELU)
    if needs[2]:
        dW = torch.empty_like(W) if w_tgt is None else None
        out = dW if w_tgt is None else w_tgt
        fl = GEMM_GELU_B | (0 if w_tgt is None else GEMM_ACCUM)
        gemm(V, Oe, T * B, dA, enc, out, a_mc=True, a_sk=V, b_sn=1, b_sk=Oe, c_div=1, c_so=O, c_si=0, flags=fl, split_k=True)
        gemm(V, Od, U1 * B, dC, dec, out, a_mc=True, a_sk=V, b_sn=1, b_sk=Od, c_off=Oe, c_div=1, c_so=O, c_si=0, flags=fl,
             split_k=True)
    if needs[3]:
        db = colsum(dA, T * B, V, into=b_tgt)
    return d_enc, d_dec, dW, db


class JointLossFn(torch.autograd.Function):
    """enc (T,B,Oe), dec (U1,B,Od) time-major -> per-utterance NLL (B,).  Never builds (B,T,U1,V).
    forward: A/C pre-GEMMs + log-softmax terms + alpha/beta (kept in a workspace); backward: the lattice gradient kernel with
    the upstream per-utterance gradient folded in (1/B under reduction="mean", model.py:39), then the joint's own backward.
    Under torch.no_grad() (validation_step) no gradient kernel runs and nothing is kept."""

    @staticmethod
    def forward(ctx, enc, dec, W, bias, labels, t_lens, u_lens, blank, want_grad=True, reduction="none"):
        _need_gpu(enc, dec, W, bias, labels, t_lens, u_lens)
        w_param, b_param = W, bias
        enc, dec, W, bias = _f32c(enc, "enc"), _f32c(dec, "dec"), _f32c(W, "fc.weight"), _f32c(bias, "fc.bias")
        for name, t in (("targets", labels), ("frame lengths", t_lens), ("target lengths", u_lens)):
            if t.dtype != torch.int32:
                raise ValueError(f"{name} must be int32 (dataloader.py:21-24), got {t.dtype}")
        labels = labels.contiguous()
        T, B, Oe = enc.shape
        U1, B2, Od = dec.shape
        V = W.shape[0]
        if B2 != B or W.shape[1] != Oe + Od or labels.shape != (B, U1 - 1):
            raise ValueError(f"shape mismatch: enc {tuple(enc.shape)} dec {tuple(dec.shape)} fc {tuple(W.shape)} "
                             f"targets {tuple(labels.shape)}")
        A, Cm = _joint_ac(enc, dec, W, Oe, Od, V)
        nll = torch.empty(B, device=enc.device, dtype=torch.float32)
        nws = _lib.lib().rnnt_hip_joint_loss_workspace_bytes(B, T, U1, V)
        ws = torch.empty(nws, device=enc.device, dtype=torch.uint8)
        check(_lib.lib().rnnt_hip_joint_loss_fwd_bwd(_addr(A), V, B * V, _addr(Cm), V, B * V, _addr(bias), _addr(labels),
                                                     _addr(t_lens), _addr(u_lens), B, T, U1, V, int(blank), 1.0,
                                                     _addr(nll), None, None, _addr(ws), nws, _stream()),
              "rnnt_hip_joint_loss_fwd_bwd")
        if want_grad and any(ctx.needs_input_grad[:4]):  # want_grad: the caller's torch.is_grad_enabled() (always off in here)
            ctx.save_for_backward(enc, dec, W, bias, labels, t_lens, u_lens, A, Cm, ws)
            ctx.blank = int(blank)
            ctx.w_param, ctx.b_param = w_param, b_param
        ctx.red_scale = {"none": None, "sum": 1.0, "mean": 1.0 / B}[reduction]
        if ctx.red_scale is None:
            return nll
        out = torch.empty((), device=enc.device, dtype=torch.float32)   # reduction inside the library: no torch arithmetic on the path
        check(_lib.lib().rnnt_hip_scaled_sum_f32(_addr(nll), B, ctx.red_scale, _addr(out), _stream()), "rnnt_hip_scaled_sum_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        enc, dec, W, bias, labels, t_lens, u_lens, A, Cm, ws = ctx.saved_tensors
        T, B, _ = enc.shape
        U1, V = dec.shape[0], W.shape[0]
        gvec = _f32c(g.to(torch.float32), "grad of the loss")
        scalar = ctx.red_scale is not None   # reduced loss: ONE upstream scalar, the 1/B of "mean" rides in gscale
        dA, dC = torch.empty_like(A), torch.empty_like(Cm)
        check(_lib.lib().rnnt_hip_joint_loss_bwd(_addr(A), V, B * V, _addr(Cm), V, B * V, _addr(bias), _addr(labels),
                                                 _addr(t_lens), _addr(u_lens), B, T, U1, V, ctx.blank,
                                                 ctx.red_scale if scalar else 1.0, _addr(gvec), 0 if scalar else 1,
                                                 _addr(dA), _addr(dC), _addr(ws), ws.numel(), _stream()),
              "rnnt_hip_joint_loss_bwd")
        d_enc, d_dec, dW, db = _joint_backward(enc, dec, W, dA, dC, ctx.needs_input_grad[:4],
                                               _direct_grad(ctx.w_param), _direct_grad(ctx.b_param))
        return d_enc, d_dec, dW, db, None, None, None, None, None, None


class JointLogitsFn(torch.autograd.Function):
    """Materialising joint for RNNTransducer.forward() (model.py:47-50): logits (B,T,U1,V)."""

    @staticmethod
    def forward(ctx, enc, dec, W, bias):
        _need_gpu(enc, dec, W, bias)
        enc, dec, W, bias = _f32c(enc, "enc"), _f32c(dec, "dec"), _f32c(W, "fc.weight"), _f32c(bias, "fc.bias")
        T, B, Oe = enc.shape
        U1, _, Od = dec.shape
        V = W.shape[0]
        A, Cm = _joint_ac(enc, dec, W, Oe, Od, V)
        logits = torch.empty(B, T, U1, V, device=enc.device, dtype=torch.float32)
        check(_lib.lib().rnnt_hip_joint_logits_fwd(_addr(A), V, B * V, _addr(Cm), V, B * V, _addr(bias), B, T, U1, V,
                                                   _addr(logits), _stream()), "rnnt_hip_joint_logits_fwd")
        ctx.save_for_backward(enc, dec, W)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        enc, dec, W = ctx.saved_tensors
        # compatibility path only (the fused training_step never comes here): two axis sums by torch
        dA = dlogits.sum(dim=2).transpose(0, 1).contiguous()   # (T,B,V)
        dC = dlogits.sum(dim=1).transpose(0, 1).contiguous()   # (U1,B,V)
        return _joint_backward(enc, dec, W, dA, dC, ctx.needs_input_grad[:4])


class RnntLossFromLogitsFn(torch.autograd.Function):
    """warp-transducer-shaped loss on dense logits (model.py:39,57) -> per-utterance NLL (B,)."""

    @staticmethod
    def forward(ctx, logits, targets, t_lens, u_lens, blank):
        _need_gpu(logits, targets, t_lens, u_lens)
        codes = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
        if logits.dtype not in codes:
            raise ValueError(f"logits must be float32, float16 or bfloat16, got {logits.dtype}")
        logits = logits if logits.is_contiguous() else logits.contiguous()
        for name, t in (("targets", targets), ("logit lengths", t_lens), ("target lengths", u_lens)):
            if t.dtype != torch.int32:
                raise ValueError(f"{name} must be int32, got {t.dtype}")
        B, T, U1, V = logits.shape
        if targets.shape != (B, U1 - 1):
            raise ValueError(f"targets must be (B, U) = ({B}, {U1 - 1}), got {tuple(targets.shape)}")
        targets = targets.contiguous()
        nll = torch.empty(B, device=logits.device, dtype=torch.float32)
        grad = torch.empty_like(logits) if logits.requires_grad else None
        nws = _lib.lib().rnnt_hip_joint_loss_workspace_bytes(B, T, U1, V)
        ws = torch.empty(nws, device=logits.device, dtype=torch.uint8)
        check(_lib.lib().rnnt_hip_loss_from_logits_fwd_bwd_ex(_addr(logits), codes[logits.dtype], _addr(targets), _addr(t_lens),
                                                              _addr(u_lens), B, T, U1, V, int(blank), 1.0, _addr(nll),
                                                              _addr(grad), _addr(ws), nws, _stream()),
              "rnnt_hip_loss_from_logits_fwd_bwd_ex")
        ctx.grad = grad
        return nll

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        ctx.grad = None
        return (grad.float() * g.to(torch.float32).view(-1, 1, 1, 1)).to(grad.dtype), None, None, None, None


# --------------------------------------------------------------------------------------------------
# greedy decoding (replaces the host loop of transducer.py:95-145)
# --------------------------------------------------------------------------------------------------
def greedy_decode(enc_tm: torch.Tensor, fc_w: torch.Tensor, fc_b: torch.Tensor, emb_w: torch.Tensor, rnn_weights,
                  cell: int, out_w: torch.Tensor, out_b: torch.Tensor, blank: int, max_iters: int,
                  t_lens: Optional[torch.Tensor] = None):
    """enc_tm (T,B,Oe) encoder outputs (time-major) -> (tokens (B, T*max_iters) int64, ntok (B,) int32), all on device.
    rnn_weights: [w_ih, w_hh, b_ih, b_hh] per prediction-net layer; t_lens (B) int32 on device = frames visited per
    utterance (None: all T)."""
    _need_gpu(enc_tm, fc_w, emb_w)
    enc_tm = _f32c(enc_tm, "encoder outputs")
    T, B, Oe = enc_tm.shape
    V, Ocat = fc_w.shape
    Od = Ocat - Oe
    Hp = emb_w.shape[1]
    L = len(rnn_weights) // 4
    if L > _lib.DECODE_MAX_LAYERS:
        raise ValueError(f"greedy decode supports at most {_lib.DECODE_MAX_LAYERS} prediction-net layers")
    _check_prednet_weights(rnn_weights, cell, Hp)
    if Od < 1 or tuple(out_w.shape) != (Od, Hp) or tuple(out_b.shape) != (Od,) or tuple(fc_b.shape) != (V,) or emb_w.shape[0] < 1:
        raise ValueError(f"greedy decode: fc {tuple(fc_w.shape)} / out_proj {tuple(out_w.shape)} / embedding {tuple(emb_w.shape)} "
                         f"do not fit encoder width {Oe}")
    if not 0 <= blank < V or max_iters < 1:
        raise ValueError(f"greedy decode: blank {blank} outside [0,{V}) or max_iters {max_iters} < 1")
    A = torch.empty(T, B, V, device=enc_tm.device, dtype=torch.float32)
    gemm(T * B, V, Oe, enc_tm, fc_w, A, b_sn=Ocat, b_sk=1, bias=fc_b, flags=GEMM_GELU_A)
    max_out = T * max_iters
    tokens = torch.full((B, max_out), blank, device=enc_tm.device, dtype=torch.int64)
    ntok = torch.zeros(B, device=enc_tm.device, dtype=torch.int32)
    d = _lib.DecodeDesc()
    d.T, d.B, d.V, d.Hp, d.O, d.L, d.cell = T, B, V, Hp, Od, L, cell
    d.blank, d.max_iters, d.max_out = blank, max_iters, max_out
    d.A, d.t_lens, d.emb = _addr(A), _addr(t_lens), _addr(emb_w)
    keep = []
    for l in range(L):
        w = [_f32c(t, "prediction-net weight") for t in rnn_weights[4 * l:4 * l + 4]]
        keep.append(w)
        d.w_ih[l], d.w_hh[l], d.b_ih[l], d.b_hh[l] = (_addr(t) for t in w)
    d.w_o, d.b_o = _addr(out_w), _addr(out_b)
    d.w_d, d.ld_d = _addr(fc_w, Oe), Ocat
    d.tokens, d.ntok = _addr(tokens), _addr(ntok)
    check(_lib.lib().rnnt_hip_greedy_decode(C.byref(d), _stream()), "rnnt_hip_greedy_decode")
    return tokens, ntok


def _check_prednet_weights(rnn_weights, cell: int, H: int) -> None:
    """Host-side shape check of a uni-directional prediction-net stack (input width == hidden width): the decode kernels
    index raw pointers."""
    ngate = {0: 4, 1: 3, 2: 1, 3: 1}.get(cell)
    if ngate is None or len(rnn_weights) % 4 != 0 or not rnn_weights:
        raise ValueError(f"prediction net: cell {cell!r} / {len(rnn_weights)} weight tensors")
    for i, w in enumerate(rnn_weights):
        want = [(ngate * H, H), (ngate * H, H), (ngate * H,), (ngate * H,)][i % 4]
        if tuple(w.shape) != want:
            raise ValueError(f"prediction-net weight #{i}: expected {want}, got {tuple(w.shape)}")


def prednet_step(tokens: torch.Tensor, emb_w: torch.Tensor, rnn_weights, cell: int, h_in=None, c_in=None):
    """One prediction-net step for a batch with carried state (decoder.py:121-123).  tokens (B,) int64; h_in / c_in (L,B,H)
    or None (zeros) -> (h_out, c_out) with c_out None unless LSTM; the layer output is h_out[-1]."""
    _need_gpu(tokens, emb_w)
    L = len(rnn_weights) // 4
    B, H = tokens.numel(), emb_w.shape[1]
    if L > _lib.DECODE_MAX_LAYERS:
        raise ValueError(f"at most {_lib.DECODE_MAX_LAYERS} prediction-net layers")
    _check_prednet_weights(rnn_weights, cell, H)
    for name, t in (("h_in", h_in), ("c_in", c_in)):
        if t is not None and tuple(t.shape) != (L, B, H):
            raise ValueError(f"{name} must be (L,B,H) = ({L},{B},{H}), got {tuple(t.shape)}")
    tokens = tokens.reshape(-1).to(torch.int64).contiguous()
    h_out = torch.empty(L, B, H, device=emb_w.device, dtype=torch.float32)
    c_out = torch.empty_like(h_out) if cell == _lib.CELL_LSTM else None
    d = _lib.PrednetStepDesc()
    d.B, d.Hp, d.L, d.cell = B, H, L, cell
    d.tokens, d.emb = _addr(tokens), _addr(emb_w)
    keep = []
    for l in range(L):
        w = [_f32c(t, "prediction-net weight") for t in rnn_weights[4 * l:4 * l + 4]]
        keep.append(w)
        d.w_ih[l], d.w_hh[l], d.b_ih[l], d.b_hh[l] = (_addr(t) for t in w)
    h_in = None if h_in is None else _f32c(h_in, "h_in")
    c_in = None if c_in is None else _f32c(c_in, "c_in")
    d.h_in, d.c_in, d.h_out, d.c_out = _addr(h_in), _addr(c_in), _addr(h_out), _addr(c_out)
    check(_lib.lib().rnnt_hip_prednet_step(C.byref(d), _stream()), "rnnt_hip_prednet_step")
    return h_out, c_out
