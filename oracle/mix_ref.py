"""Oracle: the numpy / scipy DSP subset of the reference's on-the-fly simulator, float64.

TEST INFRASTRUCTURE - never imported by the product path.

Restates ``simulation/simulate_data_from_param.py``: ``filter_designs`` :25-54, ``mix_noise`` :95-126,
``add_reverberation`` :220-230, ``clipping`` :255-276, ``packet_loss`` :333-341, the high-pass ``filtfilt`` call :461 and
the final joint peak normalisation :576-584.  The reference module itself cannot be imported here (espnet2, soundfile,
librosa are absent), but everything it calls for these functions except ``espnet2.train.preprocessor
.detect_non_silence`` is numpy / scipy, which ARE present, so this file calls the same library routines
(``scipy.signal.convolve`` / ``filtfilt`` / ``firwin2``, ``np.quantile`` / ``np.clip`` / ``np.pad(mode="wrap")``).
``detect_non_silence`` / ``framing`` are restated from espnet 202412: PARITY UNPINNED for that one function.
"""
import numpy as np
import scipy.signal


def framing(x, frame_length=512, frame_shift=256, centered=True, padded=True):
    if centered:
        pad = [(0, 0)] * (x.ndim - 1) + [(frame_length // 2, frame_length // 2)]
        x = np.pad(x, pad, mode="constant", constant_values=0)
    if padded:
        nadd = (-(x.shape[-1] - frame_length) % frame_shift) % frame_length
        x = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, nadd)], mode="constant", constant_values=0)
    n = (x.shape[-1] - frame_length) // frame_shift + 1
    idx = np.arange(frame_length)[None, :] + frame_shift * np.arange(n)[:, None]
    return x[..., idx]


def detect_non_silence(x, threshold=0.01, frame_length=1024, frame_shift=512, window="boxcar"):
    if x.shape[-1] < frame_length:
        return np.full(x.shape, True, dtype=bool)
    framed = framing(x.astype(np.float64), frame_length, frame_shift, centered=False, padded=True)
    framed = framed * scipy.signal.get_window(window, frame_length)
    power = (framed ** 2).mean(axis=-1)
    mean_power = np.mean(power, axis=-1, keepdims=True)
    if np.all(mean_power == 0):
        return np.full(x.shape, True, dtype=bool)
    det = power / mean_power > threshold
    det = np.broadcast_to(det[..., None], det.shape + (frame_shift,)).reshape(*det.shape[:-1], -1)
    return np.pad(det, [(0, 0)] * (x.ndim - 1) + [(0, x.shape[-1] - det.shape[-1])], mode="edge")


def align_noise(noise, len_speech, offset):
    """the wrap / crop branch of mix_noise (:108-119) with the offset the host drew."""
    len_noise = noise.shape[-1]
    if len_noise < len_speech:
        return np.pad(noise, [(0, 0), (offset, len_speech - len_noise - offset)], mode="wrap")
    if len_noise > len_speech:
        return noise[:, offset:offset + len_speech]
    return noise


def mix_noise(speech, noise, snr, offset=0):
    noise = align_noise(noise, speech.shape[-1], offset)
    ps = (speech[detect_non_silence(speech)] ** 2).mean()
    pn = (noise[detect_non_silence(noise)] ** 2).mean()
    scale = 10 ** (-snr / 20) * np.sqrt(ps) / np.sqrt(max(pn, 1e-10))
    noise = scale * noise
    return speech + noise, noise


def add_reverberation(speech, rir):
    return scipy.signal.convolve(speech, rir, mode="full")[:, :speech.shape[1]]


def filter_designs(fs, cutoff=70, transition_width=15, attenuation=10):
    nyq = 0.5 * fs
    stop = cutoff - transition_width
    if stop < 0:
        stop, transition_width = 0, cutoff
    norm_stop, norm_pass = stop / nyq, min(cutoff, nyq) / nyq
    numtaps = max(int((attenuation * fs) / (22 * transition_width)), 101)
    if numtaps % 2 == 0:
        numtaps += 1
    return scipy.signal.firwin2(numtaps, freq=[0, norm_stop, norm_pass, 1.0], gain=[0, 0, 1, 1])


def high_pass(speech, fs):
    return scipy.signal.filtfilt(filter_designs(fs), 1.0, speech.flatten()).reshape(speech.shape)


def clipping(speech, min_quantile=0.0, max_quantile=0.9):
    mn, mx = np.quantile(speech, np.array([min_quantile, max_quantile]), axis=-1, keepdims=False)
    return np.stack([np.clip(speech[i], mn[i], mx[i]) for i in range(speech.shape[0])], axis=0)


def packet_loss(speech, fs, indices, packet_duration_ms=20):
    speech = speech.copy()
    for idx in indices:
        speech[:, idx * packet_duration_ms * fs // 1000:(idx + 1) * packet_duration_ms * fs // 1000] = 0
    return speech


def joint_peak_normalise(speech, noisy, noise, target=0.9):
    scale = target / max(np.max(np.abs(noisy)), np.max(np.abs(speech)), np.max(np.abs(noise)), 1e-6)
    return speech * scale, noisy * scale, noise * scale


# ---- resampy (librosa res_type "kaiser_best" / "kaiser_fast") ---------------------------------------------------------------------
# librosa.resample(y, orig_sr, target_sr, res_type="kaiser_best" | "kaiser_fast") = resampy.resample(y, orig_sr, target_sr, filter=res_type)
# followed by fix_length to ceil(n * ratio) (simulate_data_from_param.py:233-252, the two resampy branches of the bandwidth limitation).
# resampy is NOT in the image and ships its two filters as pre-computed tables; they are the output of its own
# `filters.sinc_window(num_zeros, precision, window=kaiser(beta), rolloff)` with the parameters its documentation states:
#   kaiser_best: 64 zero crossings, 2^9 table samples per crossing, beta 14.769656459379492, roll-off 0.9475937167399596
#   kaiser_fast: 16 zero crossings, 2^9 table samples per crossing, beta 8.555504641634386,  roll-off 0.85
# Restated from the published algorithm (Smith's band-limited interpolation with a linearly interpolated filter table,
# resampy/interpn.py `_resample_loop`); PARITY UNPINNED against the package.
RESAMPY_FILTERS = {"kaiser_best": (64, 9, 14.769656459379492, 0.9475937167399596),
                   "kaiser_fast": (16, 9, 8.555504641634386, 0.85)}


def resampy_filter(name):
    """resampy.filters.sinc_window: the right wing of the windowed sinc on 2^precision samples per zero crossing."""
    num_zeros, precision, beta, rolloff = RESAMPY_FILTERS[name]
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def resampy_resample(x, sr_orig, sr_new, name):
    """resampy.resample(x, sr_orig, sr_new, filter=name) for a 1-D float64 signal (vectorised restatement of _resample_loop)."""
    x = np.asarray(x, dtype=np.float64)
    sample_ratio = float(sr_new) / sr_orig
    n_out = int(x.shape[0] * sample_ratio)
    interp_win, num_table = resampy_filter(name)
    if sample_ratio < 1:
        interp_win = sample_ratio * interp_win
    interp_delta = np.diff(interp_win, append=interp_win[-1])
    scale = min(1.0, sample_ratio)
    t_out = np.arange(n_out) * (1.0 / sample_ratio)
    index_step = int(scale * num_table)
    nwin, n_orig = interp_win.shape[0], x.shape[0]
    y = np.zeros(n_out)
    taps = nwin // index_step + 2
    for c0 in range(0, n_out, 4096):
        tr = t_out[c0:c0 + 4096]
        n = tr.astype(np.int64)
        frac = scale * (tr - n)
        for wing in (0, 1):
            if wing:
                frac = scale - frac
            index_frac = frac * num_table
            offset = index_frac.astype(np.int64)
            eta = index_frac - offset
            cnt = np.minimum(n + 1 if wing == 0 else n_orig - n - 1, (nwin - offset) // index_step)
            i = np.arange(taps)[None, :]
            ok = i < cnt[:, None]
            idx = np.where(ok, offset[:, None] + i * index_step, 0)
            w = interp_win[idx] + eta[:, None] * interp_delta[idx]
            xi = np.where(ok, (n[:, None] - i) if wing == 0 else (n[:, None] + i + 1), 0)
            y[c0:c0 + 4096] += np.where(ok, w * x[xi], 0.0).sum(1)
    return y


def bandwidth_limitation_resampy(x, fs, fs_new, name):
    """bandwidth_limitation(x[None], fs, fs_new, res_type=name)[0] (simulate_data_from_param.py:233-252)."""
    import math
    x = np.asarray(x, dtype=np.float64)

    def librosa_resample(y, orig, target):
        out = resampy_resample(y, orig, target, name)
        n = int(math.ceil(len(y) * float(target) / orig))
        return out[:n] if len(out) >= n else np.pad(out, (0, n - len(out)))
    down = librosa_resample(x, fs, fs_new)
    up = librosa_resample(down, fs_new, fs)
    return up[:len(x)] if len(up) >= len(x) else np.pad(up, (0, len(x) - len(up)))
