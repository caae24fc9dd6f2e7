"""FlatAdamW — torch.optim.AdamW (what the reference configures at model.py:111-115) with every parameter, gradient and
moment of the module living in ONE flat fp32 buffer each:

  * `p.data` / `p.grad` become views into `flat_param` / `flat_grad`, so autograd accumulates straight into the flat
    gradient buffer, the data-parallel exchange is a single RCCL all-reduce over it (dist.py), and
  * `step()` is ONE fused HIP kernel (`rnnt_hip_adamw_step`) instead of torch's multi-tensor passes.

It stays a `torch.optim.AdamW` subclass: `param_groups` (so OneCycleLR drives `lr` exactly as in the reference),
`state_dict()` (per-parameter `exp_avg` / `exp_avg_sq` are views into the flat moments) and `zero_grad()` keep working.
"""
from typing import Iterable

import torch
import torch.distributed as dist

from . import _lib
from .ops import _addr, _stream


class FlatAdamW(torch.optim.AdamW):
    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if not ps:
            raise ValueError("no trainable parameters")
        dev = ps[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in ps):
            raise _lib.RnntHipError("FlatAdamW needs all parameters in float32 on one GPU (move the module first)")
        sizes = [(p.numel() + 3) // 4 * 4 for p in ps]  # keep every view 16-byte aligned
        total = sum(sizes)
        self.flat_param = torch.zeros(total, device=dev)
        self.flat_grad = torch.zeros(total, device=dev)
        self.flat_m = torch.zeros(total, device=dev)
        self.flat_v = torch.zeros(total, device=dev)
        off = 0
        for p, n in zip(ps, sizes):
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + k].view_as(p)
            p.grad = self.flat_grad[off:off + k].view_as(p)
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.flat_m[off:off + k].view_as(p),
                             "exp_avg_sq": self.flat_v[off:off + k].view_as(p)}
            off += n
        self._flat_params = ps
        self._steps = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def zero_grad(self, set_to_none: bool = False) -> None:  # keep the views: autograd accumulates in place
        self.flat_grad.zero_()

    def all_reduce_grads(self) -> None:
        """DDP semantics (train.py:45): SUM over ranks, x 1/world — one collective over the flat buffer."""
        if self.world > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
            self.flat_grad.mul_(1.0 / self.world)

    def grad_bytes(self) -> int:
        return self.flat_grad.numel() * 4

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        if len(self.param_groups) != 1:
            raise _lib.RnntHipError("FlatAdamW supports the reference's single parameter group (model.py:112)")
        self._steps += 1
        b1, b2 = g["betas"]
        _lib.check(_lib.lib().rnnt_hip_adamw_step(_addr(self.flat_param), _addr(self.flat_grad), _addr(self.flat_m),
                                                  _addr(self.flat_v), self.flat_param.numel(), float(g["lr"]), float(b1),
                                                  float(b2), float(g["eps"]), float(g["weight_decay"]), self._steps,
                                                  _stream()), "rnnt_hip_adamw_step")
        for p in self._flat_params:
            self.state[p]["step"] += 1
        return loss
