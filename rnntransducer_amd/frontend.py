"""LogMelFrontend — the reference's offline input side (datamodule.py:48-90) on the MI355X.

Per utterance the reference runs, on the host and once per dataset: `mean_var_norm` (:87-90) -> torchaudio
`MelSpectrogram(sample_rate, win_length = n_fft = ceil(sr * window_size_sec), hop_length = sr * window_stride_sec, n_mels)`
(:48-66; torchaudio defaults: hann window, center=True / reflect pad, power 2, HTK mel scale, no filterbank norm)
-> `log1p` (:67) -> SpecAugment (:74-85) -> transpose to (time, mel).  Here a padded batch of waveforms goes through
three launches (csrc/frontend.hip): normalise + reflect-pad, ONE GEMM against the windowed DFT basis with the frames
addressed in place, power -> mel -> log1p.  The output is what `dataloader.py:40` would have padded: (B, T_max, n_mels)
with zeros after each utterance's last frame, plus the frame counts.

torchaudio is not a dependency: the filterbank below restates torchaudio.functional.melscale_fbanks(norm=None,
mel_scale="htk") from its documentation, and the tests pin the STFT against torch.stft.
"""
import math
from typing import Sequence, Tuple, Union

import torch
import torch.nn as nn

from . import _lib
from .networks.encoder import lengths_to_device
from .ops import _addr, _need_gpu, _stream, gemm


def melscale_fbanks_htk(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """(n_freqs, n_mels) triangular filters, HTK mel scale, no area normalisation (torchaudio's defaults for MelSpectrogram)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    hz_to_mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)  # noqa: E731
    m_pts = torch.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0).float()


class LogMelFrontend(nn.Module):
    def __init__(self, sample_rate: int = 16000, window_size_sec: float = 0.025, window_stride_sec: float = 0.01,
                 n_mels: int = 80, normalize: bool = True):
        super().__init__()
        self.sample_rate = sample_rate
        self.n_fft = int(math.ceil(sample_rate * window_size_sec))       # datamodule.py:51-54: win_length = n_fft
        self.hop = int(sample_rate * window_stride_sec)                   # :56
        self.n_mels, self.normalize = n_mels, normalize
        if self.n_fft % 16 != 0 or self.hop % 4 != 0:
            raise ValueError("n_fft must be a multiple of 16 and hop a multiple of 4 samples (GEMM K-tile / 16-byte row starts)")
        self.n_bins = self.n_fft // 2 + 1
        n = torch.arange(self.n_fft, dtype=torch.float64)
        window = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / self.n_fft)   # torch.hann_window(periodic=True)
        ang = 2.0 * math.pi * torch.arange(self.n_bins, dtype=torch.float64).unsqueeze(1) * n.unsqueeze(0) / self.n_fft
        basis = torch.cat([torch.cos(ang) * window, -torch.sin(ang) * window], 0)   # (2*n_bins, n_fft): [re | im]
        self.register_buffer("basis", basis.float(), persistent=False)
        self.register_buffer("fb", melscale_fbanks_htk(self.n_bins, 0.0, sample_rate / 2.0, n_mels, sample_rate), persistent=False)

    def num_frames(self, n_samples: int) -> int:
        return 1 + n_samples // self.hop

    @torch.no_grad()
    def forward(self, wav: torch.Tensor, lengths: Union[Sequence[int], torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        """wav (B, L_max) float32 on the GPU (anything beyond lengths[b] is ignored), lengths in samples ->
        (features (B, T_max, n_mels) float32 with zeros after each utterance's last frame, frame counts (B,) int32)."""
        _need_gpu(wav, self.basis)
        if wav.dim() != 2 or wav.dtype != torch.float32:
            raise ValueError("wav must be (B, L_max) float32")
        wav = wav.contiguous()
        B, Lmax = wav.shape
        lens = lengths_to_device(lengths, wav.device)
        P = self.n_fft // 2
        F = self.num_frames(Lmax)
        Lp = (Lmax + 2 * P + 3) // 4 * 4
        padded = torch.empty(B, Lp, device=wav.device, dtype=torch.float32)
        L = _lib.lib()
        _lib.check(L.rnnt_hip_frontend_norm_pad(_addr(wav), Lmax, _addr(lens), B, P, Lp, 1 if self.normalize else 0,
                                                _addr(padded), _stream()), "rnnt_hip_frontend_norm_pad")
        spec = torch.empty(B * F, 2 * self.n_bins, device=wav.device, dtype=torch.float32)
        gemm(B * F, 2 * self.n_bins, self.n_fft, padded, self.basis, spec, a_div=F, a_so=Lp, a_si=self.hop)
        nframes = (1 + torch.div(lens, self.hop, rounding_mode="floor")).to(torch.int32)
        nframes = torch.where(lens > 0, nframes, torch.zeros_like(nframes))
        out = torch.empty(B, F, self.n_mels, device=wav.device, dtype=torch.float32)
        _lib.check(L.rnnt_hip_power_mel_log1p(_addr(spec), B * F, self.n_bins, _addr(self.fb), self.n_mels, _addr(nframes), F,
                                              _addr(out), _stream()), "rnnt_hip_power_mel_log1p")
        return out, nframes


def spec_augment(feats: torch.Tensor, frame_lengths: torch.Tensor, freq_mask_param: int, time_mask_param: int,
                 freq_mask_cnt: int = 1, time_mask_cnt: int = 1, generator: torch.Generator = None) -> torch.Tensor:
    """SpecAugment as datamodule.py:74-85 applies it per utterance (torchaudio FrequencyMasking / TimeMasking: width
    ~ U[0, param), start ~ U[0, size - width), masked with 0), on a padded (B, T, n_mels) batch: every utterance draws its own
    masks and time masks stay inside its own frame count.  Random by construction: not parity-pinnable, property-tested."""
    B, T, M = feats.shape
    dev = feats.device
    out = feats.clone()
    rnd = lambda: torch.rand(B, device=dev, generator=generator)  # noqa: E731
    mel = torch.arange(M, device=dev).view(1, 1, M)
    frm = torch.arange(T, device=dev).view(1, T, 1)
    for _ in range(freq_mask_cnt):
        width = rnd() * freq_mask_param
        start = rnd() * (M - width)
        lo, hi = start.long().view(B, 1, 1), (start.long() + width.long()).view(B, 1, 1)
        out = out.masked_fill((mel >= lo) & (mel < hi), 0.0)
    size = frame_lengths.to(dev).float()
    for _ in range(time_mask_cnt):
        width = torch.minimum(rnd() * time_mask_param, size)
        start = rnd() * (size - width)
        lo, hi = start.long().view(B, 1, 1), (start.long() + width.long()).view(B, 1, 1)
        out = out.masked_fill((frm >= lo) & (frm < hi), 0.0)
    return out
