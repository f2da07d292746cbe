"""Batched intrusive SE metrics on the GPU, behind the reference's function surface.

``evaluation_metrics/calculate_intrusive_se_metrics.py``: ``estoi_metric(ref, inf, fs)`` (:37-48),
``sdr_metric(ref, inf)`` (:90-109), ``pesq_metric`` (:52-88), ``process_one_pair`` / ``main`` (:114-169: scp in,
``{METRIC}.scp`` lines ``"uid value"`` + ``RESULTS.txt`` ``"METRIC: mean:.4f"`` out).  The batched entry points
(``estoi_batch`` / ``sdr_batch``) take ``[P, L]`` device tensors; the per-pair functions wrap them.
"""
import math

import numpy as np
import torch

from ._lib import call, require_cuda, stream_ptr

STOI_FS = 10000
_cache = {}


def _resample_plan(fs, L, device):
    """host-side filter design of pystoi.utils._resample_window_oct + scipy.signal.resample_poly's padding rule."""
    key = (fs, L, device)
    if key in _cache:
        return _cache[key]
    g = math.gcd(STOI_FS, fs)
    up, down = STOI_FS // g, fs // g
    cutoff = 1.0 / (2 * max(up, down))
    roll = cutoff / 10
    rej = 60.0
    Lh = int(math.ceil((rej - 8) / (28.714 * roll)))
    t = np.arange(-Lh, Lh + 1)
    h = np.kaiser(2 * Lh + 1, 0.1102 * (rej - 8.7)) * (2 * up * cutoff * np.sinc(2 * cutoff * t))
    h = h / h.sum() * up
    half = (len(h) - 1) // 2
    n_pre_pad = down - half % down
    n_pre_remove = (half + n_pre_pad) // down
    n_out = L * up // down + (1 if (L * up) % down else 0)
    hp = np.concatenate([np.zeros(n_pre_pad), h])
    plan = (torch.from_numpy(hp).to(device), len(hp), up, down, n_pre_remove, n_out)
    _cache[key] = plan
    return plan


def _tw512(device):
    key = ("tw512", device)
    if key not in _cache:
        j = np.arange(512)
        a = -2.0 * np.pi * j / 512.0
        _cache[key] = torch.from_numpy(np.stack([np.cos(a), np.sin(a)], 1).astype(np.float32)).to(device)
    return _cache[key]


def resample_to_10k(x, fs):
    """pystoi.utils.resample_oct(x, 10000, fs) for a batch f32 [P, L]."""
    require_cuda(x)
    x = x.contiguous().float()
    P, L = x.shape
    if fs == STOI_FS:
        return x
    hp, hlen, up, down, npr, n_out = _resample_plan(int(fs), L, x.device)
    y = torch.empty(P, n_out, device=x.device, dtype=torch.float32)
    call("resample_poly", x, y, hp, hlen, P, L, n_out, up, down, npr, stream_ptr())
    return y


def estoi_batch(ref, inf, fs):
    """ESTOI of P pairs: ref/inf f32 [P, L] (same length) at sampling rate fs -> f32 [P]."""
    require_cuda(ref, inf)
    assert ref.shape == inf.shape and ref.dim() == 2
    x, y = resample_to_10k(ref, fs), resample_to_10k(inf, fs)
    P, L = x.shape
    dev = x.device
    nfr = max(1, (L - 256 + 127) // 128 if L > 256 else 0)
    out = torch.empty(P, device=dev, dtype=torch.float32)
    ws_x, ws_y = torch.empty(P, L, device=dev), torch.empty(P, L, device=dev)
    tob_x, tob_y = torch.empty(P, 15, nfr, device=dev), torch.empty(P, 15, nfr, device=dev)
    lens = torch.empty(P, device=dev, dtype=torch.int32)
    call("estoi_batch", x, y, out, ws_x, ws_y, tob_x, tob_y, lens, _tw512(dev), P, L, stream_ptr())
    return out


def sdr_batch(ref, inf, clamp_db=50.0):
    """SDR (dB) of P single-source pairs f32 [P, L] -> f32 [P]."""
    require_cuda(ref, inf)
    assert ref.shape == inf.shape and ref.dim() == 2
    ref, inf = ref.contiguous().float(), inf.contiguous().float()
    P, L = ref.shape
    dev = ref.device
    out = torch.empty(P, device=dev, dtype=torch.float32)
    acf = torch.empty(P, 512, device=dev, dtype=torch.float64)
    xc = torch.empty(P, 512, device=dev, dtype=torch.float64)
    norms = torch.empty(P, 2, device=dev, dtype=torch.float64)
    call("sdr_batch", ref, inf, out, acf, xc, norms, P, L, float(clamp_db), stream_ptr())
    return out


PESQ_TRACE = 286      # include/urse.h URSE_PESQ_TRACE


SOXR_HQ_BITS = 20        # soxr.h: SOXR_HQ = "20-bit" quality


def soxr_hq_spec(fs_in, fs_out):
    """The low-pass SPECIFICATION libsoxr's HQ recipe asks of its filters (soxr.c `soxr_quality_spec`, quality 4): stop-band
    rejection 20 bits x 6.0206 dB = 120.4 dB, stop band from the Nyquist frequency of the lower rate, pass band up to
    1 - 0.05 / TO_3dB(rej) = 0.9136 of it (TO_3dB(a) = (1.6e-6 a - 7.5e-4) a + 0.646), linear phase.  -> (f_pass, f_stop, att_dB) in Hz."""
    rej = SOXR_HQ_BITS * 20.0 * math.log10(2.0)
    to3db = (1.6e-6 * rej - 7.5e-4) * rej + 0.646
    nyq = 0.5 * min(fs_in, fs_out)
    return (1.0 - 0.05 / to3db) * nyq, nyq, rej


def _design(kind, up, down, fs_in):
    """host filter design -> (taps f64 scaled by `up`, half length); the polyphase kernel evaluates it exactly (f64 accumulate)."""
    from scipy.signal import firwin
    mx = max(up, down)
    if kind == "scipy":         # scipy.signal.resample_poly's default: Kaiser(5.0), 20 * max(up, down) + 1 taps, cutoff at the lower Nyquist
        half = 10 * mx
        return firwin(2 * half + 1, 1.0 / mx, window=("kaiser", 5.0)) * up, half
    # "soxr_hq": ONE Kaiser-windowed sinc meeting soxr HQ's specification (cutoff in the middle of the transition band, as soxr's
    # lsx_design_lpf places it); soxr itself cascades several stages designed to the same total specification, so the two agree to
    # the specification's own ripple (2^-20) below f_pass and both reject >= 120 dB above f_stop; they may differ inside the
    # 0.9136-1.0 x Nyquist transition band.  Not bit-equal to libsoxr (not reproducible).
    fs_work = float(fs_in) * up
    f_pass, f_stop, att = soxr_hq_spec(fs_in, fs_in * up / down)
    dw = 2.0 * math.pi * (f_stop - f_pass) / fs_work
    att += 2.0               # design margin: Kaiser's length estimate lands 1.3 dB short at the very band edges for the 80 k-tap filters
    half = int(math.ceil((att - 7.95) / (2.285 * dw) / 2.0))
    h = firwin(2 * half + 1, (f_pass + f_stop) / fs_work, window=("kaiser", 0.1102 * (att - 8.7)))
    return h * up, half


def _poly_resample(x, fs_in, fs_out, design="scipy"):
    """``scipy.signal.resample_poly(x, up, down, window=h)`` on the batched polyphase kernel, f32 [P, L] -> f32 [P, ceil(L up / down)].
    design "scipy": its default Kaiser(5.0) filter (librosa's res_type="polyphase", the bandwidth-limitation augmentation);
    design "soxr_hq": a filter built to libsoxr's HQ specification (`_design`), the stand-in for ``soxr.resample`` in
    pesq_metric (:69-70) and for ``librosa.resample(res_type="soxr_hq")`` in the simulator's read_audio
    (simulate_data_from_param.py:350-352)."""
    require_cuda(x)
    x = x.contiguous().float()
    P, L = x.shape
    g = math.gcd(int(fs_in), int(fs_out))
    up, down = int(fs_out) // g, int(fs_in) // g
    key = ("poly", design, up, down, int(fs_in) if design != "scipy" else 0, x.device)     # (not L: the filter does not depend on it)
    if key not in _cache:
        from . import ops
        h, half = _design(design, up, down, int(fs_in))
        n_pre_pad = down - half % down
        hp = np.concatenate([np.zeros(n_pre_pad), h])
        _cache[key] = (ops.upload(torch.from_numpy(hp), x.device, cached=True), len(hp), (half + n_pre_pad) // down)
    hp, hlen, npr = _cache[key]
    n_out = L * up // down + (1 if (L * up) % down else 0)
    y = torch.empty(P, n_out, device=x.device, dtype=torch.float32)
    call("resample_poly", x, y, hp, hlen, P, L, n_out, up, down, npr, stream_ptr())
    return y


def resample_soxr_hq(x, fs_in, fs_out):
    return _poly_resample(x, fs_in, fs_out, design="soxr_hq")


def pesq_batch(ref, inf, fs, mode=None, lens=None, return_trace=False, max_pairs_per_launch=2048):
    """PESQ (MOS-LQO) of P pairs f32 [P, L] -> f32 [P]; NaN where pesq.pesq returns NO_UTTERANCES_DETECTED (pesq_metric maps
    that to None / nan, :81-88, :160-162).  mode None = the rule of pesq_metric: 'nb' at 8 kHz, 'wb' at 16 kHz, higher rates
    are resampled to 16 kHz first.  return_trace: also the integer outputs of the alignment stages, int32 [P, 286]."""
    require_cuda(ref, inf)
    assert ref.shape == inf.shape and ref.dim() == 2
    fs = int(fs)
    if mode is None:
        if fs == 8000:
            mode = "nb"
        elif fs >= 16000:
            mode = "wb"
        else:
            raise ValueError("sample rate must be 8000 or 16000+ for PESQ evaluation, but got %d" % fs)
    if fs > 16000:
        ref, inf = resample_soxr_hq(ref, fs, 16000), resample_soxr_hq(inf, fs, 16000)
        if lens is not None:
            lens = (torch.as_tensor(lens).long() * 16000 + fs - 1) // fs
        fs = 16000
    ref, inf = ref.contiguous().float(), inf.contiguous().float()
    P, L = ref.shape
    dev = ref.device
    if lens is not None:
        lens = torch.as_tensor(lens).to(device=dev, dtype=torch.int32).contiguous()
    mos = torch.empty(P, device=dev, dtype=torch.float32)
    raw = torch.empty(P, device=dev, dtype=torch.float32)
    trace = torch.zeros(P, PESQ_TRACE, device=dev, dtype=torch.int32)
    import ctypes
    nbytes = ctypes.c_int64()
    chunk = min(P, max_pairs_per_launch)
    lib = __import__("urgent2026_challenge_track1_amd._lib", fromlist=["load"]).load()
    if lib.urse_pesq_workspace_bytes(chunk, L, fs, ctypes.addressof(nbytes)) != 0:
        raise RuntimeError(lib.urse_last_error().decode())
    key = ("pesq_ws", dev)
    if key not in _cache or _cache[key].numel() < nbytes.value:
        _cache[key] = torch.empty(nbytes.value, device=dev, dtype=torch.uint8)
    ws = _cache[key]
    for p0 in range(0, P, chunk):
        n = min(chunk, P - p0)
        call("pesq_batch", ref[p0:], inf[p0:], ref.stride(0), None if lens is None else lens[p0:], n, L, fs,
             1 if mode == "wb" else 0, mos[p0:], raw[p0:], trace[p0:], ws, ws.numel(), stream_ptr())
    return (mos, raw, trace) if return_trace else mos


_side_streams = {}


def score_batch(ref, inf, fs, names=("PESQ", "ESTOI"), slice_pairs=256, pesq_pairs_per_launch=2048):
    """All requested metrics of P pairs f32 [P, L] -> {name: f32 [P]} (device tensors).  PESQ keeps the stream it is called on (one
    workgroup per pair, latency-bound: ~13 ms per pair, four pairs per CU); ESTOI and SDR - short bandwidth-bound kernels - run
    in `slice_pairs` slices on a second stream BESIDE it, in the issue slots the PESQ workgroups leave idle; the caller's stream
    waits for both before it returns.  (calculate_intrusive_se_metrics.py:114-132 scores one pair per pool task.)"""
    require_cuda(ref, inf)
    dev = ref.device
    out = {}
    side_names = [n for n in names if n in ("ESTOI", "SDR")]
    cur = torch.cuda.current_stream(dev)
    side = None
    if side_names and "PESQ" in names:
        side = _side_streams.get(dev)
        if side is None:
            side = _side_streams[dev] = torch.cuda.Stream(dev)
        side.wait_stream(cur)                                   # the inputs are ready on the caller's stream
    with torch.cuda.stream(side) if side is not None else torch.cuda.stream(cur):
        parts = {n: [] for n in side_names}
        for i in range(0, ref.shape[0], slice_pairs):
            r, e = ref[i:i + slice_pairs], inf[i:i + slice_pairs]
            if "ESTOI" in parts:
                parts["ESTOI"].append(estoi_batch(r, e, fs))
            if "SDR" in parts:
                parts["SDR"].append(sdr_batch(r, e))
        for n in side_names:
            out[n] = torch.cat(parts[n]) if parts[n] else torch.empty(0, device=dev)
    if "PESQ" in names:
        out["PESQ"] = pesq_batch(ref, inf, fs, max_pairs_per_launch=pesq_pairs_per_launch)
    if side is not None:
        cur.wait_stream(side)
        for t in (ref, inf):
            t.record_stream(side)
        for n in side_names:
            out[n].record_stream(cur)
    return out


# ---- the reference's per-pair functions ----------------------------------------------------------------------
def _dev(x, device="cuda"):
    return torch.as_tensor(np.asarray(x, dtype=np.float32)).reshape(1, -1).to(device)


def estoi_metric(ref, inf, fs=16000):
    return float(estoi_batch(_dev(ref), _dev(inf), fs)[0])


def pesq_metric(ref, inf, fs=8000):
    """-> MOS-LQO, or None when no utterance is detected (calculate_intrusive_se_metrics.py:52-88)."""
    assert np.shape(ref) == np.shape(inf)
    v = float(pesq_batch(_dev(ref), _dev(inf), fs)[0])
    return None if math.isnan(v) else v


def sdr_metric(ref, inf):
    assert np.shape(ref) == np.shape(inf)
    return float(sdr_batch(_dev(ref), _dev(inf))[0])
