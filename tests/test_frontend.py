"""Input side: LogMelFrontend on the GPU vs the CPU oracle (oracle/frontend_oracle.py: torch.stft in float64 +
the documented HTK filterbank), SpecAugment properties, and the oracle's own sanity on the CPU."""
import math

import numpy as np
import pytest
import torch

from oracle.frontend_oracle import log_mel, melscale_fbanks_htk64


def test_oracle_filterbank_and_tone():
    fb = melscale_fbanks_htk64(201, 0.0, 8000.0, 80, 16000)
    assert fb.shape == (201, 80) and fb.min() >= 0.0 and fb.max() <= 1.0 + 1e-12
    peaks = fb.argmax(0)
    assert np.all(np.diff(peaks) >= 0) and peaks[0] >= 1 and peaks[-1] <= 199  # ordered triangles inside (0, Nyquist)
    assert np.all(fb[0] == 0.0)                                                # DC belongs to no filter (f_min = 0)
    # a pure tone lights up the filter whose triangle contains its frequency
    t = np.arange(16000) / 16000.0
    f0 = 1000.0
    feats = log_mel(np.sin(2 * math.pi * f0 * t))
    assert feats.shape == (1 + 16000 // 160, 80)
    mel_of = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)  # noqa: E731
    centre = int(round(mel_of(f0) / mel_of(8000.0) * 81)) - 1
    assert abs(int(feats[50].argmax()) - centre) <= 1


def test_package_filterbank_equals_oracle_constants():
    from rnntransducer_amd.frontend import melscale_fbanks_htk
    got = melscale_fbanks_htk(201, 0.0, 8000.0, 80, 16000).double().numpy()
    assert np.abs(got - melscale_fbanks_htk64(201, 0.0, 8000.0, 80, 16000)).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("lengths", [[16000], [4000, 1600, 2399, 400, 241], [48000, 31999]])
def test_log_mel_matches_oracle(lengths):
    from rnntransducer_amd.frontend import LogMelFrontend
    g = torch.Generator().manual_seed(sum(lengths))
    B, Lmax = len(lengths), max(lengths)
    wav = torch.zeros(B, Lmax)
    for b, n in enumerate(lengths):  # speech-like dynamics: noise with a slow envelope, a tone and a DC offset
        t = torch.arange(n) / 16000.0
        wav[b, :n] = (0.3 * torch.randn(n, generator=g) * (0.5 + 0.5 * torch.sin(2 * math.pi * 3.0 * t)) + 0.2 * torch.sin(2 * math.pi * 440.0 * t) + 0.05)
        wav[b, n:] = 123.0  # garbage beyond the length must be ignored
    fe = LogMelFrontend().cuda()
    feats, nframes = fe(wav.cuda(), lengths)
    assert feats.shape == (B, 1 + Lmax // 160, 80) and nframes.tolist() == [1 + n // 160 for n in lengths]
    for b, n in enumerate(lengths):
        want = log_mel(wav[b, :n].numpy())
        got = feats[b, :nframes[b]].double().cpu().numpy()
        # fp32 DFT (split-bf16 GEMM, fp32 accumulate) vs float64: log1p compresses; absolute tolerance on the features
        assert np.abs(got - want).max() < 2e-4 * max(1.0, np.abs(want).max()), (b, np.abs(got - want).max())
        assert torch.all(feats[b, nframes[b]:] == 0)  # the collate's padding value (dataloader.py:40)
    # un-normalised variant
    raw, _ = LogMelFrontend(normalize=False).cuda()(wav.cuda(), lengths)
    want = log_mel(wav[0, :lengths[0]].numpy(), normalize=False)
    assert np.abs(raw[0, :nframes[0]].double().cpu().numpy() - want).max() < 2e-4 * max(1.0, np.abs(want).max())


@pytest.mark.gpu
def test_spec_augment_properties():
    from rnntransducer_amd.frontend import spec_augment
    gen = torch.Generator(device="cuda").manual_seed(3)
    feats = torch.ones(6, 50, 80, device="cuda")
    lens = torch.tensor([50, 40, 30, 20, 10, 5], dtype=torch.int32, device="cuda")
    out = spec_augment(feats, lens, freq_mask_param=15, time_mask_param=12, freq_mask_cnt=2, time_mask_cnt=2, generator=gen)
    assert out.shape == feats.shape and torch.all((out == 0) | (out == 1))
    for b in range(6):
        z = out[b] == 0
        mel_cols = z.all(0).sum().item()       # fully masked mel bins
        frm_rows = z.all(1).sum().item()       # fully masked frames
        assert mel_cols < 2 * 15 and frm_rows < 2 * 12
        assert not z[lens[b]:].all(1).any() or mel_cols == 80  # time masks stay inside the utterance
    assert torch.equal(spec_augment(feats, lens, 15, 12, 0, 0), feats)
