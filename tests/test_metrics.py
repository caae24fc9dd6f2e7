"""WER / CER of the validation side against the known answers of the torchmetrics documentation examples (the reference uses
torchmetrics.WordErrorRate / CharErrorRate: model.py:7,41-42) and hand-checked edit distances."""
import torch

from rnntransducer_amd.metrics import char_error_rate, edit_distance, token_error_rate, word_error_rate

PREDS = ["this is the prediction", "there is an other sample"]
TARGET = ["this is the reference", "there is another one"]


def test_documented_examples():
    assert abs(word_error_rate(PREDS, TARGET).item() - 0.5) < 1e-6        # torchmetrics WordErrorRate docstring: tensor(0.5000)
    assert abs(char_error_rate(PREDS, TARGET).item() - 0.3415) < 5e-5    # torchmetrics CharErrorRate docstring: tensor(0.3415)


def test_edit_distance_and_token_rate():
    assert edit_distance("kitten", "sitting") == 3
    assert edit_distance([], [1, 2, 3]) == 3 and edit_distance([1, 2, 3], []) == 3 and edit_distance([], []) == 0
    assert edit_distance([1, 2, 3], [1, 2, 3]) == 0
    ter = token_error_rate([torch.tensor([1, 2, 3]), [4, 5]], [torch.tensor([1, 3]), [4, 5, 6, 7]])
    assert abs(ter.item() - (1 + 2) / (2 + 4)) < 1e-6
    assert word_error_rate("a b", "a b").item() == 0.0
    assert word_error_rate("", "a b").item() == 1.0
