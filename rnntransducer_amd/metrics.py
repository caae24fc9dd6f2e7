"""Word / character error rates of the validation side (model.py:7,41-42,94-95 use torchmetrics' WordErrorRate and
CharErrorRate, which is not a dependency here).  Same definition: sum of Levenshtein distances over sum of reference
lengths, words = str.split(), characters = every character of the string including spaces.  Host-side integer work on a few
hundred short strings per validation run: nothing to accelerate.  `token_error_rate` is the same ratio on token-id
sequences (what validation_step returns when no tokenizer is attached)."""
from typing import Iterable, Sequence, Union

import torch


def edit_distance(pred: Sequence, ref: Sequence) -> int:
    """Levenshtein distance (substitution = insertion = deletion = 1), two-row DP."""
    if len(pred) == 0:
        return len(ref)
    prev = list(range(len(ref) + 1))
    for i, p in enumerate(pred, 1):
        cur = [i] + [0] * len(ref)
        for j, r in enumerate(ref, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (p != r))
        prev = cur
    return prev[-1]


def _rate(pairs: Iterable) -> torch.Tensor:
    errors = total = 0
    for p, r in pairs:
        errors += edit_distance(p, r)
        total += len(r)
    return torch.tensor(errors / total if total else 0.0)


def _as_list(x: Union[str, Sequence[str]]):
    return [x] if isinstance(x, str) else list(x)


def word_error_rate(preds: Union[str, Sequence[str]], target: Union[str, Sequence[str]]) -> torch.Tensor:
    preds, target = _as_list(preds), _as_list(target)
    if len(preds) != len(target):
        raise ValueError("preds and target need the same number of sentences")
    return _rate((p.split(), t.split()) for p, t in zip(preds, target))


def char_error_rate(preds: Union[str, Sequence[str]], target: Union[str, Sequence[str]]) -> torch.Tensor:
    preds, target = _as_list(preds), _as_list(target)
    if len(preds) != len(target):
        raise ValueError("preds and target need the same number of sentences")
    return _rate((list(p), list(t)) for p, t in zip(preds, target))


def token_error_rate(preds: Sequence, target: Sequence) -> torch.Tensor:
    """Edit distance over reference length on token-id sequences (tensors or lists)."""
    to_list = lambda s: s.tolist() if isinstance(s, torch.Tensor) else list(s)  # noqa: E731
    return _rate((to_list(p), to_list(t)) for p, t in zip(preds, target))
