"""Generates tests/golden/collate.npz from the REFERENCE's own dataloader.py::AudioDataLoader._collate_fn (dataloader.py:16-49).
Run ONLY in the build container:  python tests/golden/make_golden_collate.py     (data only; nothing of the reference is copied)"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from dataloader import AudioDataLoader  # noqa: E402  (the reference's)

g = torch.Generator().manual_seed(21)
frames, labels = [17, 9, 23, 1], [4, 2, 7, 1]
samples = [{"input_values": torch.randn(t, 80, generator=g), "input_ids": torch.randint(1, 72, (u,), generator=g).tolist()}
           for t, u in zip(frames, labels)]
loader = types.SimpleNamespace(pad_token_id=0, bos_token_id=1, n_mels=80)
out = AudioDataLoader._collate_fn(loader, samples)
names = ["input_audios", "audio_lengths", "tensor_audio_lengths", "input_texts", "text_lengths", "targets", "target_lengths"]
blob = {}
for i, s in enumerate(samples):
    blob[f"sample{i}/input_values"] = s["input_values"].numpy()
    blob[f"sample{i}/input_ids"] = np.array(s["input_ids"], np.int64)
for n, v in zip(names, out):
    blob["out/" + n] = v.numpy() if isinstance(v, torch.Tensor) else np.array(v, np.int64)
    blob["dtype/" + n] = np.array(str(v.dtype) if isinstance(v, torch.Tensor) else "list")
np.savez_compressed(os.path.join(HERE, "collate.npz"), **blob)
print({n: (tuple(v.shape), str(v.dtype)) if isinstance(v, torch.Tensor) else v for n, v in zip(names, out)})
