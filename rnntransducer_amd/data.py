"""Synthetic KsponSpeech-shaped batches in the reference's 7-tuple layout (dataloader.py:16-49) and the
length-grouped rank sharding of datasampler.py:74-99 (sort by length descending, pad by wrap-around to a multiple
of the world size, deal `indices[rank::world]`).  Recipe: SURVEY.md §8(d)."""
from typing import List, Sequence

import torch


def synthetic_batch(B: int, T: int, U: int, V: int, n_mels: int = 80, ragged: bool = False, seed: int = 1234,
                    blank: int = 0, device="cpu", t_lengths=None):
    """(input_audios f32 (B,T,n_mels), audio_lengths list, tensor_audio_lengths i32 (B,), input_texts i64 (B,U+1),
    text_lengths list, targets i32 (B,U), target_lengths i32 (B,)) — frames beyond T_b are 0.0, labels in 1..V-1."""
    g = torch.Generator().manual_seed(seed)
    audios = torch.randn(B, T, n_mels, generator=g)
    if t_lengths is not None:  # lengths decided by the caller (e.g. a rank's share of a length-sorted global batch)
        t_list = [int(t) for t in t_lengths]
        assert len(t_list) == B and max(t_list) <= T and min(t_list) >= 1
        u_list = [max(1, min(U, round(U * t / T))) for t in t_list]
    elif ragged:
        t_list = torch.randint(max(1, T // 2), T + 1, (B,), generator=g).tolist()
        t_list[0] = T
        u_list = [max(1, round(U * t / T)) for t in t_list]
        u_list[0] = U
    else:
        t_list, u_list = [T] * B, [U] * B
    targets = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32)
    for b in range(B):
        audios[b, t_list[b]:] = 0.0
        targets[b, u_list[b]:] = blank
    texts = torch.cat([torch.full((B, 1), blank, dtype=torch.int64), targets.to(torch.int64)], dim=1)
    return (audios.to(device), t_list, torch.tensor(t_list, dtype=torch.int32, device=device), texts.to(device),
            [u + 1 for u in u_list], targets.to(device), torch.tensor(u_list, dtype=torch.int32, device=device))


def global_ragged_lengths(n: int, T: int, seed: int = 1234) -> List[int]:
    """Frame counts of a global batch of n KsponSpeech-shaped utterances: U{T/2..T}, at least one at T."""
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(max(1, T // 2), T + 1, (n,), generator=g).tolist()
    t[0] = T
    return t


def length_grouped_indices(lengths: Sequence[int], rank: int, world: int) -> List[int]:
    """datasampler.py:74-99 behaviour: descending length, wrap-pad to a multiple of world, rank-strided deal."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    pad = (-len(order)) % world
    order = order + order[:pad]
    return order[rank::world]


def collate_batch(batch, pad_token_id: int, n_mels: int):
    """The 7-tuple of dataloader.py:16-49 from a list of samples {"input_values": (T_i, n_mels) float tensor,
    "input_ids": label ids}: audio / texts / targets padded with pad_token_id, prediction-net input = [pad] + labels
    (int64, as the reference's torch.cat promotes it), targets int32, python-list lengths for the packers and IntTensor
    lengths for the loss."""
    from torch.nn.utils.rnn import pad_sequence
    input_audios = [s["input_values"] for s in batch]
    audio_lengths = [int(a.size(0)) for a in input_audios]
    targets = [torch.as_tensor(s["input_ids"], dtype=torch.int32) for s in batch]
    target_lengths = torch.tensor([len(s["input_ids"]) for s in batch], dtype=torch.int32)
    input_texts = [torch.cat([torch.full((1,), pad_token_id, dtype=torch.int64), t.to(torch.int64)]) for t in targets]
    text_lengths = [int(t.numel()) for t in input_texts]
    if input_audios[0].size(-1) != n_mels:
        raise ValueError(f"feature width {input_audios[0].size(-1)} != n_mels {n_mels} (dataloader.py:38)")
    return (pad_sequence(input_audios, batch_first=True, padding_value=pad_token_id), audio_lengths,
            torch.tensor(audio_lengths, dtype=torch.int32), pad_sequence(input_texts, batch_first=True, padding_value=pad_token_id),
            text_lengths, pad_sequence(targets, batch_first=True, padding_value=pad_token_id), target_lengths)


class AudioDataLoader(torch.utils.data.DataLoader):
    """dataloader.py:5-14: a DataLoader whose collate_fn emits the 7-tuple above."""

    def __init__(self, pad_token_id, bos_token_id, n_mels, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.pad_token_id, self.bos_token_id, self.n_mels = pad_token_id, bos_token_id, n_mels
        self.collate_fn = lambda batch: collate_batch(batch, self.pad_token_id, self.n_mels)
