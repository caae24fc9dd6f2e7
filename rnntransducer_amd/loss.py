"""RNNTLoss — drop-in for the two loss modules the reference constructs at model.py:31,39
(`warprnnt_pytorch.RNNTLoss(blank, reduction="mean")` / `torchaudio.transforms.RNNTLoss(...)`) and calls at
model.py:57 as `loss(logits, targets, logit_lengths, target_lengths)`, on the HIP alpha/beta kernels.
Returns a 0-d tensor for "mean"/"sum" (the warp-transducer build returns shape (1,), model.py:83-88; documented
difference), or (B,) for reduction="none".
"""
import torch
import torch.nn as nn

from .ops import RnntLossFromLogitsFn


class RNNTLoss(nn.Module):
    def __init__(self, blank: int = 0, reduction: str = "mean"):
        super().__init__()
        if reduction not in ("mean", "sum", "none"):
            raise ValueError(f"reduction must be mean|sum|none, got {reduction!r}")
        self.blank, self.reduction = int(blank), reduction

    def forward(self, logits: torch.Tensor, targets: torch.Tensor, logit_lengths: torch.Tensor,
                target_lengths: torch.Tensor) -> torch.Tensor:
        if logits.dim() != 4:
            raise ValueError("logits must be (B, T, U+1, V)")
        nll = RnntLossFromLogitsFn.apply(logits, targets, logit_lengths, target_lengths, self.blank)
        if self.reduction == "mean":
            return nll.mean()
        if self.reduction == "sum":
            return nll.sum()
        return nll
