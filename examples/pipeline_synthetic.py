"""End-to-end on synthetic audio, every stage on the MI355X HIP path:
waveforms -> LogMelFrontend (norm, DFT-as-GEMM, mel, log1p) -> SpecAugment -> 7-tuple (collate) -> RNNTransducer.training_step
(fused joint + RNN-T loss) -> FlatAdamW / OneCycleLR -> validation_step (fused loss + on-device greedy search) -> error rate.

    python examples/pipeline_synthetic.py [--steps 300] [--batch 8] [--seconds 2.0]
"""
import argparse
import math
import os
import sys
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from rnntransducer_amd import LogMelFrontend, RNNTransducer, collate_batch, spec_augment  # noqa: E402


def synthetic_utterances(n, seconds, vocab, gen):
    """'Speech': one tone per label, 0.2 s each, in noise; label ids 1..vocab-1 without immediate repeats."""
    out = []
    for _ in range(n):
        n_lab = max(1, int(seconds / 0.2 * (0.6 + 0.4 * torch.rand((), generator=gen).item())))
        labels, prev = [], 0
        for _ in range(n_lab):
            k = int(torch.randint(1, vocab, (), generator=gen))
            k = k % (vocab - 1) + 1 if k == prev else k
            labels.append(k)
            prev = k
        t = torch.arange(int(0.2 * 16000)) / 16000.0
        wav = torch.cat([torch.sin(2 * math.pi * (200.0 + 90.0 * k) * t) for k in labels])
        out.append((wav + 0.05 * torch.randn(wav.numel(), generator=gen), labels))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--hidden", type=int, default=128)
    a = ap.parse_args(argv)
    dev, V = torch.device("cuda:0"), 32
    gen = torch.Generator().manual_seed(0)
    utts = synthetic_utterances(a.batch, a.seconds, V, gen)
    lens = [w.numel() for w, _ in utts]
    wav = torch.zeros(a.batch, max(lens))
    for b, (w, _) in enumerate(utts):
        wav[b, :w.numel()] = w
    frontend = LogMelFrontend().to(dev)
    feats, nframes = frontend(wav.to(dev), lens)                                  # (B, T, 80) on the GPU
    samples = [{"input_values": feats[b, :int(nframes[b])], "input_ids": utts[b][1]} for b in range(a.batch)]
    batch = collate_batch(samples, pad_token_id=0, n_mels=80)                     # dataloader.py:16-49's 7-tuple
    batch = tuple(x.to(dev) if isinstance(x, torch.Tensor) else x for x in batch)

    args = Namespace(learning_rate=3e-3, weight_decay=0.0, warmup_ratio=0.1, final_div_factor=10.0, total_steps=a.steps,
                     move_metrics_to_cpu=False)
    tn = dict(input_size=80, hidden_size=a.hidden, output_size=a.hidden, num_layers=1, dropout=0.0, bidirectional=True)
    pn = dict(embedding_size=V, pad_token_id=0, hidden_size=a.hidden, output_size=a.hidden, num_layers=1, dropout=0.0)
    torch.manual_seed(0)
    model = RNNTransducer(pn, tn, dict(num_classes=V), args).to(dev).train()
    cfg = model.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    first = None
    for step in range(a.steps):
        aug = spec_augment(batch[0], batch[2], freq_mask_param=8, time_mask_param=5)
        opt.zero_grad()
        loss = model.training_step((aug,) + batch[1:], step)["loss"]
        loss.backward()
        opt.step()
        sched.step()
        first = loss.item() if first is None else first
    out = model.validation_step(batch, 0)
    ep = model.validation_epoch_end([out])
    print(f"loss {first:.3f} -> {loss.item():.3f} after {a.steps} steps; validation loss {ep['val_loss'].item():.3f}, "
          f"token error rate {ep['val_ter'].item():.3f}")
    return first, loss.item(), ep


if __name__ == "__main__":
    main()
