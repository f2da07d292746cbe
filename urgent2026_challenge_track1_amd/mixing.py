"""On-device dynamic mixing: the batched counterpart of the reference's per-utterance simulator functions
(``simulation/simulate_data_from_param.py``: ``mix_noise`` :95-126, ``add_reverberation`` :220-230, high-pass :29-56,461,
``clipping`` :255-276, ``packet_loss`` :333-341, final peak normalisation :576-584) on the HIP kernels of csrc/mix.hip.

Signals are f32 ``[B, L]`` CUDA tensors with per-utterance ``lens`` (int32 CUDA); the random draws of the recipe (SNR,
noise offset, packet indices, quantiles: ``dataset.py:232-278``) stay on the host and come in as small tensors.  The
FIR taps of the high-pass are designed on the host with the same ``scipy.signal.firwin2`` call as the reference.
"""
import functools
import os

import torch

from . import _lib, ops
from ._lib import call
from .ops import stream_ptr


def _upload(v, dtype, dev):
    """small per-batch parameter lists -> device.  Through pinned memory and non-blocking: a pageable `.to(device)` makes the
    host wait for everything queued on the stream before it, and the train step behind the simulator then starts with an
    empty queue (20 such waits per batch cost 20 ms of a 200 ms step)."""
    if torch.is_tensor(v) and v.device.type == "cuda":
        return v.to(dtype).contiguous()
    t = torch.as_tensor(v, dtype=dtype).contiguous()
    if torch.device(dev).type != "cuda":
        return t
    return t.pin_memory().to(dev, non_blocking=True)


def _i32(v, dev):
    return _upload(v, torch.int32, dev)


def _f32(v, dev):
    return _upload(v, torch.float32, dev)


@functools.lru_cache(maxsize=None)
def _device_taps(fs, dev):
    taps = filter_designs(int(fs))
    return _f32(taps, dev), _i32([len(taps)], dev), len(taps)


def nonsilence_power(x, lens):
    ops.require_cuda(x)
    B, L = x.shape
    hop = torch.empty(B * ((L + 511) // 512), dtype=torch.float64, device=x.device)
    power = torch.empty(B, dtype=torch.float64, device=x.device)
    call("nonsilence_power", x, _i32(lens, x.device), B, x.stride(0), 0.01, hop, power, stream_ptr())
    return power


def mix_noise(speech, noise_raw, noise_lens, lens, snr_db, offsets=None):
    """-> (noisy, scaled noise); ``offsets`` = the ``rng.integers`` draw of :109/:117 per utterance (0 if equal lengths)."""
    ops.require_cuda(speech, noise_raw)
    B, L = speech.shape
    dev = speech.device
    offsets = torch.zeros(B, dtype=torch.int32) if offsets is None else offsets
    noise, noisy = torch.empty_like(speech), torch.empty_like(speech)
    scratch = torch.empty(B * ((L + 511) // 512) + 2 * B, dtype=torch.float64, device=dev)
    call("mix_noise", speech, noise_raw, _i32(noise_lens, dev), noise_raw.stride(0), _i32(lens, dev), _i32(offsets, dev),
         _f32(snr_db, dev), B, speech.stride(0), noise, noisy, scratch, stream_ptr())
    return noisy, noise


FFT_CONV_MIN_TAPS = int(os.environ.get("URSE_FFT_CONV_MIN_TAPS", "4096"))    # longest RIR of the batch from which the FFT form is used
_conv_ws = {}


def add_reverberation(speech, lens, rir, rir_lens):
    """one RIR per utterance, ``scipy.signal.convolve(..., "full")[:, :L]``.  Direct form (``urse_fir_full``: L x taps multiply-adds,
    1.3 ms per call at 4 s x 1 s @ 48 kHz) for short filters; from ``FFT_CONV_MIN_TAPS`` taps on, the library's power-of-two FFTs
    (``urse_fft_convolve``, float32 like scipy's own FFT path on float32 input: within 1e-5 of the output's peak of the exact result)."""
    ops.require_cuda(speech, rir)
    B, L = speech.shape
    dev = speech.device
    host_taps = None if torch.is_tensor(rir_lens) else [int(n) for n in rir_lens]
    if host_taps is not None and max(host_taps) >= FFT_CONV_MIN_TAPS and L + max(host_taps) - 1 <= (1 << 20) and speech.stride(1) == 1 \
            and rir.stride(1) == 1:
        import ctypes
        lib = _lib.load()
        mt = max(host_taps)
        nb = ctypes.c_int64()
        if lib.urse_fft_convolve_workspace_bytes(B, L, mt, ctypes.addressof(nb)) != 0:
            raise _lib.UrseError(lib.urse_last_error().decode())
        wkey = (dev, torch.cuda.current_stream(dev).cuda_stream)      # one workspace per stream: a main-stream call must not share it with staged batches
        ws = _conv_ws.get(wkey)
        if ws is None or ws.numel() < nb.value:
            ws = _conv_ws[wkey] = torch.empty(nb.value, device=dev, dtype=torch.uint8)
        out = torch.empty_like(speech)
        call("fft_convolve", speech, _i32(lens, dev), B, speech.stride(0), rir, _i32(host_taps, dev), rir.stride(0), 1, out, L, mt,
             ws, ws.numel(), stream_ptr())
        ws.record_stream(torch.cuda.current_stream(dev))
        return out
    out = torch.empty_like(speech)
    call("fir_full", speech, _i32(lens, speech.device), speech.shape[0], speech.stride(0), rir, _i32(rir_lens, rir.device),
         rir.stride(0), 1, out, stream_ptr())
    return out


@functools.lru_cache(maxsize=None)
def filter_designs(fs, cutoff=70, transition_width=15, attenuation=10):
    """high-pass FIR taps exactly as ``filter_designs`` (:25-54): same firwin2 call, host side."""
    from scipy.signal import firwin2
    nyq = 0.5 * fs
    stop = cutoff - transition_width
    if stop < 0:
        stop, transition_width = 0, cutoff
    numtaps = max(int((attenuation * fs) / (22 * transition_width)), 101)
    if numtaps % 2 == 0:
        numtaps += 1
    return firwin2(numtaps, freq=[0, stop / nyq, min(cutoff, nyq) / nyq, 1.0], gain=[0, 0, 1, 1])


def high_pass(speech, lens, fs):
    """``filtfilt(high_pass_taps[fs], 1.0, x)`` (:461)."""
    ops.require_cuda(speech)
    B, L = speech.shape
    dev = speech.device
    taps_dev, nt_dev, nt = _device_taps(int(fs), str(dev))
    lds = L + 7 * nt
    scratch = torch.empty(2 * B * lds, dtype=torch.float32, device=dev)
    out = torch.empty_like(speech)
    call("filtfilt_fir", speech, _i32(lens, dev), B, speech.stride(0), taps_dev, nt_dev, nt, out, scratch, lds, stream_ptr())
    return out


def clipping(speech, lens, min_quantile, max_quantile):
    """in place; returns the per-utterance (min, max) thresholds."""
    ops.require_cuda(speech)
    B = speech.shape[0]
    dev = speech.device
    bounds = torch.empty(B, 2, dtype=torch.float32, device=dev)
    call("quantile_clip", speech, _i32(lens, dev), B, speech.stride(0), _f32(min_quantile, dev), _f32(max_quantile, dev), bounds,
         stream_ptr())
    return bounds


def packet_loss(speech, fs, packet_loss_indices, packet_duration_ms=20):
    """in place; ``packet_loss_indices`` = one list of packet numbers per utterance (:337-339)."""
    ops.require_cuda(speech)
    segs = [[b, idx * packet_duration_ms * fs // 1000, (idx + 1) * packet_duration_ms * fs // 1000]
            for b, lst in enumerate(packet_loss_indices) for idx in lst]
    if segs:
        call("zero_segments", speech, speech.stride(0), _i32(segs, speech.device), len(segs), stream_ptr())
    return speech


def joint_peak_normalise(speech, noisy, noise, target=0.9):
    """in place on all three (:576-584)."""
    ops.require_cuda(speech, noisy, noise)
    B = speech.shape[0]
    scratch = torch.empty(B, dtype=torch.int32, device=speech.device)
    call("joint_peak_scale", speech, noisy, noise, B, speech.stride(0), float(target), scratch, stream_ptr())
    return speech, noisy, noise


def early_rir_stop(rir, fs, early_rir_sec=0.05, level_ratio=1e-1):
    """index after which ``estimate_early_rir`` (simulation/rir_utils.py:4-20,24-61) zeroes a single-channel RIR: first
    sample above ``level_ratio * max|h|`` (searched up to the maximum) + ``early_rir_sec * fs``.  Host side: RIRs are
    read on the host and this is one argmax per file."""
    import numpy as np
    h = np.abs(np.asarray(rir).reshape(-1))
    mi = int(np.argmax(h))
    start = int(np.argmax(h[:mi + 1] > level_ratio * h[mi]))
    return start + int(early_rir_sec * fs)


def simulate_batch(speech, lens, noise_raw, noise_lens, fs, snr_db, noise_offsets=None, rir=None, rir_lens=None,
                   rir_early_stops=None, highpass=True, clip_quantiles=None, packet_loss_indices=None):
    """The supported subset of ``process_one_sample`` (simulate_data_from_param.py:440-590) for a batch that shares
    ``fs``: high-pass(speech) -> [reverberate: noisy = speech * rir, speech = speech * early rir] -> additive noise at
    ``snr_db`` -> clipping / packet loss on the noisy signal -> joint peak normalisation to 0.9.
    ``clip_quantiles`` = (min [B], max [B]) with NaN rows meaning "no clipping" is not supported: pass None to skip;
    codec / bandwidth-limitation / wind-noise augmentations need ffmpeg / librosa and stay out of scope.
    -> (speech, noisy, fs) like the reference's on-the-fly return (:586-587), plus the scaled noise."""
    if highpass:
        speech = high_pass(speech, lens, fs)
    noisy = speech
    if rir is not None:
        noisy = add_reverberation(speech, lens, rir, rir_lens)
        early = rir_lens if rir_early_stops is None else [min(int(a), int(b)) for a, b in zip(rir_lens, rir_early_stops)]
        speech = add_reverberation(speech, lens, rir, early)
    noisy, noise = mix_noise(noisy, noise_raw, noise_lens, lens, snr_db, noise_offsets)
    if clip_quantiles is not None:
        clipping(noisy, lens, clip_quantiles[0], clip_quantiles[1])
    if packet_loss_indices is not None:
        packet_loss(noisy, fs, packet_loss_indices)
    if speech.data_ptr() == noisy.data_ptr():
        speech = speech.clone()
    joint_peak_normalise(speech, noisy, noise)
    return speech, noisy, fs, noise


def bandwidth_limitation_polyphase(speech, fs, fs_new):
    """``bandwidth_limitation(x, fs, fs_new, res_type="polyphase")`` (simulate_data_from_param.py:233-252): librosa's polyphase
    branch is ``scipy.signal.resample_poly(y, target // gcd, orig // gcd)`` followed by ``fix_length`` to ``ceil(n * ratio)``,
    applied down and back up, cropped to the input length.  Runs on the batched polyphase kernel (`metrics._poly_resample`:
    scipy's default Kaiser(5.0) design).  The other three resamplers the reference draws: `bandwidth_limitation_fft` (scipy),
    `bandwidth_limitation_resampy` (kaiser_best / kaiser_fast)."""
    import math
    from .metrics import _poly_resample
    if fs == fs_new:
        return speech
    L = speech.shape[1]

    def fix(y, n):
        if y.shape[1] >= n:
            return y[:, :n].contiguous()
        out = torch.zeros(y.shape[0], n, device=y.device, dtype=y.dtype)
        out[:, :y.shape[1]] = y
        return out
    down = fix(_poly_resample(speech, fs, fs_new), int(math.ceil(L * fs_new / fs)))
    up = fix(_poly_resample(down, fs_new, fs), int(math.ceil(down.shape[1] * fs / fs_new)))
    return fix(up, L)


def _fft_resample_span(L, fs, fs_new):
    """longest transform of the down / up round trip of `scipy.signal.resample`: the up leg's input has ceil(L * r) samples and its
    output ceil(ceil(L * r) / r), which can exceed L by one."""
    import math
    r = float(fs_new) / float(fs)
    n1 = int(math.ceil(L * r))
    return max(L, n1, int(math.ceil(n1 / r)))


_fft_plans = {}        # (device, length) -> plan tensor (chirp + transformed conjugate chirp); a handful of MB each, LRU of 16
_fft_ws = {}


def _fft_plan(n, dev):
    import ctypes
    key = (dev, int(n))
    if key in _fft_plans:
        _fft_plans[key] = _fft_plans.pop(key)          # (most recently used last)
        return _fft_plans[key]
    lib = _lib.load()
    ne, nt = ctypes.c_int64(), ctypes.c_int64()
    if lib.urse_fft_resample_plan_elems(int(n), ctypes.addressof(ne), ctypes.addressof(nt)) != 0:
        raise _lib.UrseError(lib.urse_last_error().decode())
    plan = torch.empty(ne.value, 2, device=dev, dtype=torch.float32)
    tmp = torch.empty(nt.value, 2, device=dev, dtype=torch.float32)
    call("fft_resample_plan", plan, tmp, int(n), stream_ptr())
    tmp.record_stream(torch.cuda.current_stream(dev))
    ev = torch.cuda.Event()               # the plan enters a cache other streams read without an event: built once, waited for once
    ev.record(torch.cuda.current_stream(dev))
    ev.synchronize()
    _fft_plans[key] = plan
    while len(_fft_plans) > 16:
        _fft_plans.pop(next(iter(_fft_plans)))
    return plan


FFT_RESAMPLE_MAX = 1 << 19


def _fft_resample(x, num):
    """``scipy.signal.resample(x, num, axis=1)`` for real f32 [P, n]: the spectrum is truncated / zero-padded to ``num`` bins (the
    Nyquist bin doubled when it is cut, halved when it is introduced) and transformed back.  The two transforms have the
    utterance's own, arbitrary length: Bluestein transforms on the library's power-of-two FFTs (csrc/fft_any.hip,
    ``urse_fft_resample``; no rocFFT / torch.fft on the product path)."""
    import ctypes
    ops.require_cuda(x)
    x = x.contiguous().float()
    P, nx = x.shape
    num = int(num)
    dev = x.device
    pa, pb = _fft_plan(nx, dev), _fft_plan(num, dev)
    lib = _lib.load()
    nb = ctypes.c_int64()
    if lib.urse_fft_resample_workspace_bytes(P, nx, num, ctypes.addressof(nb)) != 0:
        raise _lib.UrseError(lib.urse_last_error().decode())
    wkey = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _fft_ws.get(wkey)
    if ws is None or ws.numel() < nb.value:
        ws = _fft_ws[wkey] = torch.empty(nb.value, device=dev, dtype=torch.uint8)
    y = torch.empty(P, num, device=dev, dtype=torch.float32)
    call("fft_resample", x, x.stride(0) if P > 1 else nx, y, num, pa, pb, ws, ws.numel(), P, nx, num, stream_ptr())   # (a size-1 dim may carry stride 0)
    ws.record_stream(torch.cuda.current_stream(dev))
    return y


def bandwidth_limitation_fft(speech, fs, fs_new):
    """``bandwidth_limitation(x, fs, fs_new, res_type="scipy")`` (simulate_data_from_param.py:233-252): librosa's scipy / fft branch is
    ``scipy.signal.resample(y, ceil(n * ratio))``, applied down and back up, cropped to the input length."""
    import math
    if fs == fs_new:
        return speech
    L = speech.shape[1]
    down = _fft_resample(speech, int(math.ceil(L * float(fs_new) / fs)))
    up = _fft_resample(down, int(math.ceil(down.shape[1] * float(fs) / fs_new)))
    return up[:, :L].contiguous()


# resampy's two filters (absent package: the tables are rebuilt from the parameters its documentation states; unpinned)
RESAMPY_FILTERS = {"kaiser_best": (64, 9, 14.769656459379492, 0.9475937167399596), "kaiser_fast": (16, 9, 8.555504641634386, 0.85)}
_resampy_tables = {}


def _resampy_table(name, sample_ratio, dev):
    """(win, delta, nwin, num_table) on the device: resampy.filters.sinc_window(num_zeros, precision, kaiser(beta), rolloff), scaled by
    the ratio when downsampling, and its first differences (interp_delta of resampy.resample)."""
    import numpy as np
    key = (name, float(sample_ratio) if sample_ratio < 1 else 1.0, dev)
    if key not in _resampy_tables:
        num_zeros, precision, beta, rolloff = RESAMPY_FILTERS[name]
        num_bits = 2 ** precision
        n = num_bits * num_zeros
        win = np.kaiser(2 * n + 1, beta)[n:] * (rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True)))
        if sample_ratio < 1:
            win = sample_ratio * win
        delta = np.diff(win, append=win[-1])
        _resampy_tables[key] = (ops.upload(torch.from_numpy(win), dev, cached=True), ops.upload(torch.from_numpy(delta), dev, cached=True), len(win), num_bits)
    return _resampy_tables[key]


def _resampy_resample(x, sr_orig, sr_new, name):
    """``resampy.resample(x, sr_orig, sr_new, filter=name, axis=1)`` for f32 [P, n] (``urse_resample_table``)."""
    ops.require_cuda(x)
    x = x.contiguous().float()
    P, n = x.shape
    sample_ratio = float(sr_new) / sr_orig
    n_out = int(n * sample_ratio)
    win, delta, nwin, num_table = _resampy_table(name, sample_ratio, x.device)
    scale = min(1.0, sample_ratio)
    y = torch.empty(P, max(n_out, 1), device=x.device, dtype=torch.float32)
    if n_out > 0:
        call("resample_table", x, x.stride(0) if P > 1 else n, y, y.shape[1], win, delta, nwin, P, n, n_out, 1.0 / sample_ratio, scale,
             num_table, int(scale * num_table), stream_ptr())
    return y[:, :n_out]


def bandwidth_limitation_resampy(speech, fs, fs_new, res_type):
    """``bandwidth_limitation(x, fs, fs_new, res_type="kaiser_best" | "kaiser_fast")`` (simulate_data_from_param.py:233-252):
    librosa.resample = resampy.resample + fix_length to ceil(n * ratio), down and back up, cropped to the input length."""
    import math
    if fs == fs_new:
        return speech
    L = speech.shape[1]

    def fix(y, n):
        if y.shape[1] >= n:
            return y[:, :n].contiguous()
        out = torch.zeros(y.shape[0], n, device=y.device, dtype=y.dtype)
        out[:, :y.shape[1]] = y
        return out
    down = fix(_resampy_resample(speech, fs, fs_new, res_type), int(math.ceil(L * float(fs_new) / fs)))
    up = fix(_resampy_resample(down, fs_new, fs, res_type), int(math.ceil(down.shape[1] * float(fs) / fs_new)))
    return fix(up, L)


def simulate_recipes(speech, lens, noise_raw, noise_lens, rir, rir_lens, rir_early_stops, fs, recipes, skipped=None):
    """``process_one_sample(on_the_fly=True)`` (simulate_data_from_param.py:440-590) for a batch of raw sources and the
    recipes ``dataset.draw_recipe`` drew for them (one fs per batch) -> (speech, noisy) f32 [B, L].

    Utterances without an RIR convolve with a unit impulse (exact identity).  ``clipping`` / ``packet_loss`` are applied in
    each recipe's own order: pass p handles every utterance's p-th augmentation, the others ride along with identity
    parameters (quantiles 0 / 1, no packets).  ``bandwidth_limitation`` is applied with all four resamplers the reference draws
    (polyphase, scipy / FFT, and - round 3 - resampy's kaiser_best / kaiser_fast, restated without the package); ``codec``
    (ffmpeg) and the wind-noise side-chain compressor (ffmpeg) have no device implementation: the recipe still DRAWS them (so the
    random stream matches the reference) but they are not applied - wind noise is mixed additively at its drawn SNR - and each
    omission is counted in ``skipped``."""
    ops.require_cuda(speech, noise_raw)
    B, L = speech.shape
    dev = speech.device
    skipped = {} if skipped is None else skipped

    def count(name):
        skipped[name] = skipped.get(name, 0) + 1
    host_lens = [int(n) for n in lens]
    lens = _i32(lens, dev)                      # one upload; every stage below takes the device copy
    if recipes[0].get("highpass", True):
        speech = high_pass(speech, lens, fs)
    noisy = speech
    if rir is not None and any(n > 0 for n in rir_lens):
        full = [int(n) if n > 0 else 1 for n in rir_lens]
        early = [min(int(n), int(e)) if n > 0 else 1 for n, e in zip(rir_lens, rir_early_stops)]
        # rows without an RIR are all zero: make them unit impulses
        rir = rir.clone()
        rir[:, 0].add_(_f32([0.0 if n > 0 else 1.0 for n in rir_lens], dev))
        noisy = add_reverberation(speech, lens, rir, full)
        speech = add_reverberation(speech, lens, rir, early)
    for r in recipes:
        if r.get("wind"):
            count("wind_noise")
    noisy, noise = mix_noise(noisy, noise_raw, noise_lens, lens, [float(r["snr"]) for r in recipes],
                             torch.tensor([int(r.get("noise_offset", 0)) for r in recipes], dtype=torch.int32))
    todo = []
    for r in recipes:
        mine = []
        for a in r.get("order", []):
            if a in ("clipping", "packet_loss"):
                mine.append(a)
            elif a == "bandwidth_limitation" and r["params"][a]["res_type"] in ("polyphase", "scipy", "kaiser_best", "kaiser_fast", "none") and not (
                    r["params"][a]["res_type"] == "scipy" and _fft_resample_span(int(r["length"]), fs, r["params"][a]["fs_new"]) > FFT_RESAMPLE_MAX):      # (> 10.9 s at 48 kHz)
                mine.append(a)
            else:
                count(a)
        todo.append(mine)
    for p in range(max([len(t) for t in todo] or [0])):
        for b in range(B):          # bandwidth limitation: per utterance, its own rate pair and resampler
            if len(todo[b]) > p and todo[b][p] == "bandwidth_limitation" and recipes[b]["params"]["bandwidth_limitation"]["fs_new"] != fs:
                n = host_lens[b]
                bw = recipes[b]["params"]["bandwidth_limitation"]
                if bw["res_type"] in RESAMPY_FILTERS:
                    noisy[b:b + 1, :n] = bandwidth_limitation_resampy(noisy[b:b + 1, :n].contiguous(), fs, bw["fs_new"], bw["res_type"])
                    continue
                limit = bandwidth_limitation_fft if bw["res_type"] == "scipy" else bandwidth_limitation_polyphase
                noisy[b:b + 1, :n] = limit(noisy[b:b + 1, :n].contiguous(), fs, bw["fs_new"])
        lo = [recipes[b]["params"]["clipping"]["min_quantile"] if len(todo[b]) > p and todo[b][p] == "clipping" else 0.0
              for b in range(B)]
        hi = [recipes[b]["params"]["clipping"]["max_quantile"] if len(todo[b]) > p and todo[b][p] == "clipping" else 1.0
              for b in range(B)]
        if any(len(todo[b]) > p and todo[b][p] == "clipping" for b in range(B)):
            clipping(noisy, lens, lo, hi)
        idx = [recipes[b]["params"]["packet_loss"]["packet_loss_indices"]
               if len(todo[b]) > p and todo[b][p] == "packet_loss" else [] for b in range(B)]
        if any(idx):
            packet_loss(noisy, fs, idx)
    if speech.data_ptr() == noisy.data_ptr():
        speech = speech.clone()
    joint_peak_normalise(speech, noisy, noise)
    return speech, noisy
