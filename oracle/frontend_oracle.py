"""CPU restatement of the reference's input side for ONE utterance (TEST INFRASTRUCTURE ONLY, like everything under oracle/):
datamodule.py:87-90 (mean_var_norm, numpy population variance) -> :48-66 MelSpectrogram -> :67 log1p -> (time, mel).

torchaudio (the reference's MelSpectrogram) is not importable here.  Pinning: the STFT is torch.stft itself -- the call
torchaudio.transforms.Spectrogram makes (center=True, pad_mode="reflect", periodic hann window, onesided, power 2, not
normalised) -- in float64; the HTK filterbank follows the torchaudio.functional.melscale_fbanks documentation
(norm=None, mel_scale="htk", f_min=0, f_max=sr/2) -> parity of the filterbank constants is UNPINNED by the reference."""
import math

import numpy as np
import torch


def melscale_fbanks_htk64(n_freqs, f_min, f_max, n_mels, sample_rate):
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_min, m_max = 2595.0 * math.log10(1.0 + f_min / 700.0), 2595.0 * math.log10(1.0 + f_max / 700.0)
    f_pts = 700.0 * (10.0 ** (np.linspace(m_min, m_max, n_mels + 2) / 2595.0) - 1.0)
    f_diff = np.diff(f_pts)
    slopes = f_pts[None, :] - all_freqs[:, None]
    return np.maximum(0.0, np.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))


def log_mel(wav: np.ndarray, sample_rate=16000, window_size_sec=0.025, window_stride_sec=0.01, n_mels=80, normalize=True) -> np.ndarray:
    """(L,) samples -> (1 + L // hop, n_mels) float64"""
    x = np.asarray(wav, dtype=np.float64)
    if normalize:
        x = (x - x.mean()) / np.sqrt(x.var() + 1e-7)
    n_fft = int(math.ceil(sample_rate * window_size_sec))
    hop = int(sample_rate * window_stride_sec)
    window = torch.hann_window(n_fft, periodic=True, dtype=torch.float64)
    X = torch.stft(torch.from_numpy(x), n_fft, hop_length=hop, win_length=n_fft, window=window, center=True, pad_mode="reflect",
                   normalized=False, onesided=True, return_complex=True)
    power = (X.real ** 2 + X.imag ** 2).numpy()                        # (n_bins, frames)
    fb = melscale_fbanks_htk64(n_fft // 2 + 1, 0.0, sample_rate / 2.0, n_mels, sample_rate)
    return np.log1p(power.T @ fb)
