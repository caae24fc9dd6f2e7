"""FlatParams / FlatAdamW — the optimiser and the data-parallel exchange of the hot path on flat fp32 buffers.

`FlatParams` (device-agnostic: it also builds on CPU tensors, which is how the world-2 gloo test runs the very code the GPU
path ships) lays every trainable parameter and its gradient out back to back in ONE buffer each:

  * `p.data` / `p.grad` become views into `flat_param` / `flat_grad`;
  * the data-parallel exchange (`all_reduce_grads`, replacing Lightning's DDPStrategy -> DistributedDataParallel bucketed NCCL
    all-reduce of train.py:45 / scripts/run_train.sh:9,26) is ONE collective over `flat_grad` — RCCL over xGMI on the GPU
    (config 2: 24.3 M parameters = 97.3 MB), SUM over ranks; the x 1/world of DDP's average is folded into the update kernel;
  * with `direct_grads=True` the backward kernels add weight gradients straight into the views (ops._direct_grad) instead
    of returning tensors that autograd would `+=` with one extra kernel per parameter.  Leave it off when
    torch.nn.parallel.DistributedDataParallel wraps the module: DDP's reducer listens to autograd's accumulation hooks.

`FlatAdamW` = torch.optim.AdamW (what the reference configures at model.py:111-115: same hyper-parameters, `param_groups` so
OneCycleLR drives `lr` exactly as in the reference, `state_dict()` / `load_state_dict()` in torch's format) whose `step()` is ONE
fused HIP kernel over the flat buffers (`rnnt_hip_adamw_step_ex`), guarded on device by the persistent recurrences' sticky
status word: if an LSTM kernel of this step gave up on an inter-workgroup wait, the update is skipped on device and the next
`step()` raises RnntHipError — no host synchronisation on the way (one 4-byte asynchronous read-back per step).

At world > 1 the status word is PART OF THE COLLECTIVE: the flat gradient buffer carries one extra slot behind the last
parameter, every rank writes (its word != 0) there before the all-reduce, and after the SUM the slot holds the number of ranks
whose recurrences gave up.  That slot — not the local word — is then the update kernel's guard and what `step()` reads back, so
ALL ranks skip the update and ALL ranks raise at their next `step()`; no rank walks into the next all-reduce alone.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from . import _lib


class FlatParams:
    def __init__(self, params: Iterable[torch.nn.Parameter], direct_grads: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise _lib.RnntHipError("FlatParams needs all parameters in float32 on one device (move the module first)")
        self.sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]  # keep every view 16-byte aligned
        self.offsets = [0]
        for n in self.sizes[:-1]:
            self.offsets.append(self.offsets[-1] + n)
        total = sum(self.sizes)
        self.n_param = total                       # elements the update kernel walks
        # + 4 floats (one 16-byte granule) behind the last parameter; element `n_param` of flat_grad is the STATUS SLOT of the
        # data-parallel exchange (see all_reduce_grads), the other three stay zero
        self.flat_param = torch.zeros(total + 4, device=dev)
        self.flat_grad = torch.zeros(total + 4, device=dev)
        self.direct_grads = bool(direct_grads)
        for p, off in zip(self.params, self.offsets):
            self.flat_param[off:off + p.numel()].copy_(p.data.reshape(-1))
        self.adopt()
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def view(self, flat: torch.Tensor, i: int) -> torch.Tensor:
        p, off = self.params[i], self.offsets[i]
        return flat[off:off + p.numel()].view_as(p)

    def adopt(self) -> None:
        """(Re)points every parameter's .data / .grad at its slice of the flat buffers.  A gradient found elsewhere
        (`zero_grad(set_to_none=True)` followed by a backward, a `.grad` assigned by hand) is copied in first."""
        for i, p in enumerate(self.params):
            pv, gv = self.view(self.flat_param, i), self.view(self.flat_grad, i)
            if p.data.data_ptr() != pv.data_ptr():
                if p.data.device != self.flat_param.device:
                    raise _lib.RnntHipError("a parameter moved to another device after the optimizer was built; build the "
                                            "optimizer after model.to(device)")
                pv.copy_(p.data)
                p.data = pv
            if p.grad is None:
                gv.zero_()   # no gradient arrived since it was dropped: the slice must not keep an older step's values
                p.grad = gv
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
                p.grad = gv
            p._rnnt_direct_grad = self.direct_grads

    def views_in_place(self) -> bool:
        esz = self.flat_param.element_size()
        p0, g0 = self.flat_param.data_ptr(), self.flat_grad.data_ptr()
        return all(p.grad is not None and p.data_ptr() == p0 + off * esz and p.grad.data_ptr() == g0 + off * esz
                   for p, off in zip(self.params, self.offsets))

    def zero_grad(self) -> None:
        if not self.views_in_place():
            self.adopt()
        self.flat_grad.zero_()

    def status_slot(self) -> torch.Tensor:
        """1-element view of flat_grad behind the last parameter: after `all_reduce_grads(status=...)` the number of ranks
        whose status word was raised (0.0 = every rank's recurrences completed)."""
        return self.flat_grad[self.n_param:self.n_param + 1]

    def all_reduce_grads(self, status: Optional[torch.Tensor] = None) -> float:
        """DDP semantics (train.py:45): SUM over ranks in ONE collective over the flat buffer; returns the 1/world factor
        that turns the sum into DDP's average (FlatAdamW folds it into the update kernel; `average_grads` applies it here).
        `status`: this rank's device status word (any integer tensor, element 0 is read); (word != 0) rides in the buffer's
        status slot, so that the SAME collective tells every rank whether any rank's gradients are invalid."""
        if self.world > 1:
            if not self.views_in_place():
                self.adopt()
            slot = self.status_slot()
            if status is not None:
                slot.copy_(status.reshape(-1)[:1].ne(0))   # on the gradients' stream, in front of the collective
            else:
                slot.zero_()
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
        return 1.0 / self.world

    def average_grads(self) -> None:
        scale = self.all_reduce_grads()
        if scale != 1.0:
            self.flat_grad.mul_(scale)

    def grad_bytes(self) -> int:
        return self.flat_grad.numel() * self.flat_grad.element_size()


class FlatAdamW(torch.optim.AdamW):
    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, direct_grads: bool = False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.flat = FlatParams([p for g in self.param_groups for p in g["params"]], direct_grads=direct_grads)
        self.flat_param, self.flat_grad = self.flat.flat_param, self.flat.flat_grad
        self.flat_m = torch.zeros_like(self.flat_param)
        self.flat_v = torch.zeros_like(self.flat_param)
        self._steps = 0
        self._grad_scale = 1.0
        self._reduced = False   # all_reduce_grads() ran since the last step(): the status slot of the collective is the guard
        self._point_state()
        self.world = self.flat.world
        self._status_host: Optional[torch.Tensor] = None   # pinned read-back slot of the device status word
        self._status_event = None

    # ---- state <-> flat moments ----------------------------------------------------------------------------------
    def _point_state(self) -> None:
        for i, p in enumerate(self.flat.params):
            st = self.state[p]
            if "step" not in st:
                st["step"] = torch.tensor(float(self._steps))
            st["exp_avg"] = self.flat.view(self.flat_m, i)
            st["exp_avg_sq"] = self.flat.view(self.flat_v, i)

    def load_state_dict(self, state_dict) -> None:
        """torch's format in, flat buffers out: loaded moments are copied into flat_m / flat_v, `state[p]` is re-pointed at the
        views and the step count (bias correction) is restored — a resumed run continues exactly where the saved one stopped."""
        super().load_state_dict(state_dict)
        steps = 0
        for i, p in enumerate(self.flat.params):
            st = self.state.get(p, {})
            if "exp_avg" in st:
                self.flat.view(self.flat_m, i).copy_(st["exp_avg"])
                self.flat.view(self.flat_v, i).copy_(st["exp_avg_sq"])
                steps = max(steps, int(float(st.get("step", 0))))
            else:
                self.flat.view(self.flat_m, i).zero_()
                self.flat.view(self.flat_v, i).zero_()
        self._steps = steps
        for p in self.flat.params:
            self.state[p]["step"] = torch.tensor(float(steps))
        self._point_state()

    # ---- gradients -----------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False) -> None:  # keep the views: autograd / the kernels accumulate in place
        self.flat.zero_grad()

    def all_reduce_grads(self) -> None:
        """One RCCL all-reduce (SUM) over the flat gradient buffer; the 1/world of DDP's average rides in the next step().
        At world > 1 this rank's LSTM status word travels in the same collective (FlatParams.all_reduce_grads)."""
        status = None
        if self.world > 1 and self.flat_param.is_cuda:
            from .ops import lstm_status_word
            status = lstm_status_word(self.flat_param.device)
        self._grad_scale = self.flat.all_reduce_grads(status)
        self._reduced = self.world > 1

    def grad_bytes(self) -> int:
        return self.flat.grad_bytes()

    # ---- status word of the persistent recurrences ----------------------------------------------------------------
    def _check_previous_status(self) -> None:
        if self._status_event is None:
            return
        self._status_event.synchronize()  # recorded a whole step ago: already complete, costs nothing
        self._status_event = None
        if int(self._status_host[0]) != 0:   # uint32 view: a raised word, or the bits of a positive rank count
            from .ops import lstm_status_word
            lstm_status_word(self.flat_param.device).zero_()
            # the device skipped that update: the host-side counters (bias correction of the NEXT update, state_dict's step) go
            # back to the number of updates actually applied
            self._steps -= 1
            for p in self.flat.params:
                self.state[p]["step"] -= 1
            where = "of this rank or a peer (the word travels in the gradient all-reduce)" if self.world > 1 else "of the previous step"
            raise _lib.RnntHipError(f"a persistent LSTM kernel {where} abandoned an inter-workgroup wait (4 s bound: "
                                    "its workgroups were not co-resident — is another kernel holding the CUs?).  That step's "
                                    "gradients were invalid; its AdamW update was skipped on device on every rank, parameters "
                                    "are intact and the step counters were rewound.")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) != 1:
            raise _lib.RnntHipError("FlatAdamW supports the reference's single parameter group (model.py:112)")
        if not self.flat_param.is_cuda:
            raise _lib.RnntHipError("FlatAdamW.step() runs on the MI355X only (rnnt_hip_adamw_step_ex); on CPU tensors this class "
                                    "provides the flat buffers and the collective, not an update")
        from .ops import _addr, _stream, lstm_status_word
        self._check_previous_status()
        if not self.flat.views_in_place():
            self.flat.adopt()  # e.g. zero_grad(set_to_none=True) by an outer loop: gradients found elsewhere are copied in
        g = self.param_groups[0]
        self._steps += 1
        b1, b2 = g["betas"]
        # guard of the update kernel (*guard != 0 as uint32 -> skip): the local sticky word, or — after a data-parallel exchange —
        # the collective's status slot (fp32 count of ranks with a raised word: 0.0 is all-zero bits, any count >= 1 is not), so
        # that every rank takes the same decision
        if self.world > 1 and not self._reduced:
            # gradients were exchanged by someone else (DistributedDataParallel's reducer under a Lightning trainer): every rank is
            # in step() now, so a 4-byte SUM of (word != 0) through the same slot keeps the ranks' decisions identical
            slot = self.flat.status_slot()
            slot.copy_(lstm_status_word(self.flat_param.device)[:1].ne(0))
            dist.all_reduce(slot, op=dist.ReduceOp.SUM)
            self._reduced = True
        if self._reduced:
            status = self.flat.status_slot().view(torch.int32)
        else:
            status = lstm_status_word(self.flat_param.device)[:1]
        _lib.check(_lib.lib().rnnt_hip_adamw_step_ex(_addr(self.flat_param), _addr(self.flat_grad), _addr(self.flat_m),
                                                     _addr(self.flat_v), self.flat.n_param, float(g["lr"]), float(b1),
                                                     float(b2), float(g["eps"]), float(g["weight_decay"]), self._steps,
                                                     float(self._grad_scale), _addr(status), _stream()), "rnnt_hip_adamw_step_ex")
        self._grad_scale = 1.0
        self._reduced = False
        if self._status_host is None:
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._status_host.copy_(status, non_blocking=True)
        self._status_event = torch.cuda.Event()
        self._status_event.record()
        for p in self.flat.params:
            self.state[p]["step"] += 1
        return loss
