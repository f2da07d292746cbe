"""Seeded PESQ test pairs shared by the CPU (oracle) and GPU (HIP vs oracle) tests: speech-like signals with silent edges and
pauses (VAD / utterance logic), additive noise over a range of SNRs, constant and piecewise delays (crude / fine alignment,
utterance splitting), a dropout (bad-interval realignment) and a silent reference (NO_UTTERANCES_DETECTED)."""
import numpy as np


def speech_like(rng, L, fs):
    n = rng.standard_normal(L)
    k = np.fft.rfftfreq(L)
    x = np.fft.irfft(np.fft.rfft(n) / (1.0 - 0.95 * np.exp(-2j * np.pi * k)), n=L)
    x /= x.std()
    t = np.arange(L) / fs
    x *= 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi))
    e = min(int(0.4 * fs), L // 4)
    x[:e] *= 1e-3
    x[L - e:] *= 1e-3
    return 0.9 * x / np.abs(x).max()


def shift(x, d):
    if d == 0:
        return x
    return np.concatenate([np.zeros(d), x[:-d]]) if d > 0 else np.concatenate([x[-d:], np.zeros(-d)])


CASES = [   # (fs, mode, seconds, snr_db | None, delay, variant)
    (8000, "nb", 4.0, None, 0, ""), (8000, "nb", 4.0, 15.0, 0, ""), (8000, "nb", 4.0, -5.0, 0, ""),
    (8000, "nb", 5.0, 12.0, 123, "pause"), (8000, "nb", 3.0, 20.0, -57, ""),
    (16000, "wb", 4.0, 20.0, 0, ""), (16000, "wb", 4.0, 5.0, 37, ""), (16000, "wb", 6.0, 10.0, -300, "pause"),
    (16000, "wb", 4.0, 25.0, 0, "jump"), (16000, "wb", 4.0, 30.0, 0, "dropout"), (16000, "wb", 2.0, None, 0, "silent"),
]


def make_case(i):
    fs, mode, seconds, snr, delay, variant = CASES[i]
    rng = np.random.default_rng(100 + i)
    L = int(seconds * fs)
    clean = speech_like(rng, L, fs)
    if variant == "pause":
        clean[L // 2 - fs // 3:L // 2 + fs // 3] *= 1e-3
    if variant == "silent":         # one 0.1 s burst: shorter than the minimum utterance -> NO_UTTERANCES_DETECTED
        clean = np.zeros(L)
        clean[L // 2:L // 2 + fs // 10] = 0.3 * rng.standard_normal(fs // 10)
    deg = clean.copy()
    if variant == "jump":           # the second half arrives 20 ms late: the utterance must be split
        d = int(0.02 * fs)
        deg = np.concatenate([clean[:L // 2], np.zeros(d), clean[L // 2:L - d]])
    if variant == "dropout":
        deg[L // 2:L // 2 + fs // 8] = 0.0
    if snr is not None:
        p = (clean ** 2).mean() if clean.any() else 1e-3
        deg = deg + rng.standard_normal(L) * np.sqrt(p / 10 ** (snr / 10))
    if variant == "silent":
        deg = deg + 1e-4 * rng.standard_normal(L)
    return fs, mode, clean.astype(np.float32), shift(deg, delay).astype(np.float32)


TRACE_KEYS = ("crude_delay", "n_utterances", "utt_start", "utt_end", "utt_delay", "start_frame", "stop_frame", "bad_intervals")
