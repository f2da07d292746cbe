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


# ---- the reference's per-pair functions ----------------------------------------------------------------------
def _dev(x, device="cuda"):
    return torch.as_tensor(np.asarray(x, dtype=np.float32)).reshape(1, -1).to(device)


def estoi_metric(ref, inf, fs=16000):
    return float(estoi_batch(_dev(ref), _dev(inf), fs)[0])


def sdr_metric(ref, inf):
    assert np.shape(ref) == np.shape(inf)
    return float(sdr_batch(_dev(ref), _dev(inf))[0])
