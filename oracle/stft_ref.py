"""Oracle: espnet2 ``Stft`` / ``STFTEncoder`` / ``STFTDecoder`` behaviour.

TEST INFRASTRUCTURE - never imported by the product path.

Reference call sites: ``baseline_code/models/bsrnn.py:14-25,37,40`` (encoder /
decoder construction and calls), ``baseline_code/flow_model.py:26-42,134-146``
(exponent spec transform).  The arithmetic itself lives in espnet==202412
(``espnet2/layers/stft.py``, ``espnet2/enh/encoder/stft_encoder.py``,
``espnet2/enh/decoder/stft_decoder.py``), absent from /root/reference, so this
is a restatement on stock ``torch.stft`` / ``torch.istft`` (parity unpinned by
the reference; cross-checked against the plain-numpy DFT below).
"""
import math

import numpy as np
import torch


def reconfig_for_fs(n_fft, hop, fs, default_fs):
    """STFTEncoder._reconfig_for_fs: scale n_fft / win / hop by fs // default_fs."""
    if fs is None:
        return n_fft, hop
    fs = int(fs)
    return n_fft * fs // default_fs, hop * fs // default_fs


def stft(x, n_fft, hop, window="hann", ilens=None):
    """espnet ``Stft.forward``: returns complex [B, T, F] (and olens).

    center=True, pad_mode="reflect", normalized=False, onesided=True, periodic
    Hann window of win_length=n_fft (``window=None`` -> rectangular).  Frames
    t >= olens[b] are zeroed when ``ilens`` is given.
    """
    win = torch.hann_window(n_fft, dtype=x.dtype) if window == "hann" else torch.ones(n_fft, dtype=x.dtype)
    X = torch.stft(x, n_fft, hop, n_fft, win, center=True, pad_mode="reflect",
                   normalized=False, onesided=True, return_complex=True)
    X = X.transpose(1, 2)  # [B, T, F]
    olens = None
    if ilens is not None:
        ilens = torch.as_tensor(ilens)
        olens = (ilens + 2 * (n_fft // 2) - n_fft) // hop + 1
        t = torch.arange(X.shape[1])[None, :]
        X = X.masked_fill((t >= olens[:, None])[..., None], 0.0)
    return X, olens


def istft(X, n_fft, hop, length, window="hann"):
    """espnet ``Stft.inverse``: X complex [B, T, F] -> wav [B, length]."""
    win = torch.hann_window(n_fft, dtype=X.real.dtype) if window == "hann" else torch.ones(n_fft, dtype=X.real.dtype)
    return torch.istft(X.transpose(1, 2), n_fft, hop, n_fft, win, center=True,
                       normalized=False, onesided=True, length=int(length))


def spec_transform(X, kind, factor=0.15, exponent=0.5):
    """STFTEncoder.spec_transform_func (flow model uses 'exponent')."""
    if kind in (None, "none"):
        return X
    if kind == "exponent":
        if exponent != 1:
            X = X.abs() ** exponent * torch.exp(1j * X.angle())
        return X * factor
    if kind == "log":
        return torch.log1p(X.abs()) * torch.exp(1j * X.angle()) * factor
    raise ValueError(kind)


def spec_back(X, kind, factor=0.15, exponent=0.5):
    """STFTDecoder.spec_back."""
    if kind in (None, "none"):
        return X
    if kind == "exponent":
        X = X / factor
        if exponent != 1:
            X = X.abs() ** (1.0 / exponent) * torch.exp(1j * X.angle())
        return X
    if kind == "log":
        X = X / factor
        return torch.expm1(X.abs()) * torch.exp(1j * X.angle())
    raise ValueError(kind)


# ---------------------------------------------------------------------------
# Independent plain-numpy restatement (float64 direct DFT), used by the tests to
# check that the torch-based oracle above means what the docstring says.
# ---------------------------------------------------------------------------
def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_numpy(x, n_fft, hop, window="hann"):
    x = np.asarray(x, dtype=np.float64)
    B, L = x.shape
    p = n_fft // 2
    xp = np.pad(x, ((0, 0), (p, p)), mode="reflect")
    T = L // hop + 1
    w = hann_periodic(n_fft) if window == "hann" else np.ones(n_fft)
    n = np.arange(n_fft)
    k = np.arange(n_fft // 2 + 1)
    D = np.exp(-2j * np.pi * np.outer(n, k) / n_fft)  # [n, k]
    out = np.zeros((B, T, len(k)), dtype=np.complex128)
    for t in range(T):
        out[:, t] = (xp[:, t * hop:t * hop + n_fft] * w) @ D
    return out


def istft_numpy(X, n_fft, hop, length, window="hann"):
    X = np.asarray(X, dtype=np.complex128)
    B, T, F = X.shape
    w = hann_periodic(n_fft) if window == "hann" else np.ones(n_fft)
    frames = np.fft.irfft(X, n=n_fft, axis=-1) * w  # [B, T, n]
    tot = n_fft + hop * (T - 1)
    y = np.zeros((B, tot))
    env = np.zeros(tot)
    for t in range(T):
        y[:, t * hop:t * hop + n_fft] += frames[:, t]
        env[t * hop:t * hop + n_fft] += w * w
    p = n_fft // 2
    y = y[:, p:p + length]
    env = env[p:p + length]
    out = np.zeros((B, length))
    out[:, :y.shape[1]] = y / env[:y.shape[1]]
    return out
