"""Oracle: intrusive metrics of ``evaluation_metrics/calculate_intrusive_se_metrics.py`` on CPU.

TEST INFRASTRUCTURE - never imported by the product path.

``estoi_metric`` (:37-48) -> ``pystoi.stoi(ref, inf, fs_sig=fs, extended=True)`` (pystoi==0.4.1, requirements.txt:2);
``sdr_metric`` (:90-109) -> ``fast_bss_eval.bss_eval_sources(ref, inf, compute_permutation=False, clamp_db=50)``.
Neither package is on disk: both are restated here from their published algorithms on numpy +
``scipy.signal.resample_poly`` (the routine pystoi itself calls).  Parity with the real packages: unpinned by the
reference (no vectors); self-consistency checks live in tests/test_metrics_oracle.py.
"""
import numpy as np
from scipy.signal import resample_poly

FS = 10000
N_FRAME = 256
NFFT = 512
NUMBAND = 15
MINFREQ = 150
N = 30
DYN_RANGE = 40
EPS = np.finfo("float").eps


def resample_window_oct(p, q):
    """pystoi.utils._resample_window_oct: Kaiser-windowed sinc, Octave compatible."""
    g = np.gcd(p, q)
    p, q = p // g, q // g
    stopband_cutoff_f = 1.0 / (2 * max(p, q))
    roll_off_width = stopband_cutoff_f / 10
    rejection_db = 60.0
    L = int(np.ceil((rejection_db - 8) / (28.714 * roll_off_width)))
    t = np.arange(-L, L + 1)
    ideal = 2 * p * stopband_cutoff_f * np.sinc(2 * stopband_cutoff_f * t)
    beta = 0.1102 * (rejection_db - 8.7)
    return np.kaiser(2 * L + 1, beta) * ideal, p, q


def resample_oct(x, p, q):
    h, pr, qr = resample_window_oct(p, q)
    return resample_poly(x, pr, qr, window=h / np.sum(h))


def thirdoct(fs, nfft, num_bands, min_freq):
    f = np.linspace(0, fs, nfft + 1)[: nfft // 2 + 1]
    k = np.arange(num_bands).astype(float)
    cf = np.power(2.0 ** (1.0 / 3), k) * min_freq
    freq_low = min_freq * np.power(2.0, (2 * k - 1) / 6)
    freq_high = min_freq * np.power(2.0, (2 * k + 1) / 6)
    obm = np.zeros((num_bands, len(f)))
    lo, hi = [], []
    for i in range(len(cf)):
        fl = int(np.argmin(np.square(f - freq_low[i])))
        fh = int(np.argmin(np.square(f - freq_high[i])))
        obm[i, fl:fh] = 1
        lo.append(fl)
        hi.append(fh)
    return obm, cf, lo, hi


# pystoi's frame loops are ``range(0, len(x) - framelen, hop)`` (which records the package's code: "last full frame
# excluded - keep the off-by-one"): a signal of exactly framelen + k * hop samples has k frames, not k + 1.  That is the reading this
# oracle and the kernels implement.  LAST_FRAME_INCLUSIVE = True switches the oracle to the other reading (``len - framelen + 1``)
# so that tests/test_metrics_gpu.py::test_estoi_frame_range_at_the_boundary_length can show, on a length = 256 mod 128 fixture,
# which one the kernels follow and what the difference is worth (one frame; a few 1e-4 of ESTOI on a 2.6 s signal).
LAST_FRAME_INCLUSIVE = False


def _frames(x, framelen, hop):
    w = np.hanning(framelen + 2)[1:-1]
    idx = list(range(0, len(x) - framelen + (1 if LAST_FRAME_INCLUSIVE else 0), hop))
    return np.array([w * x[i:i + framelen] for i in idx]).reshape(len(idx), framelen)


def _overlap_and_add(frames, hop):
    n, fl = frames.shape
    out = np.zeros((n - 1) * hop + fl if n > 0 else 0)
    for i in range(n):
        out[i * hop:i * hop + fl] += frames[i]
    return out


def remove_silent_frames(x, y, dyn_range, framelen, hop):
    xf, yf = _frames(x, framelen, hop), _frames(y, framelen, hop)
    energies = 20 * np.log10(np.linalg.norm(xf, axis=1) + EPS)
    mask = (np.max(energies) - dyn_range - energies) < 0
    return _overlap_and_add(xf[mask], hop), _overlap_and_add(yf[mask], hop), mask


def stft_tob(x, obm):
    fr = _frames(x, N_FRAME, N_FRAME // 2)
    spec = np.fft.rfft(fr, n=NFFT).T            # [257, M]
    return np.sqrt(np.matmul(obm, np.square(np.abs(spec))))


def row_col_normalize(x):
    """x [J, 15, 30]; the EPS * randn dither of pystoi (2.2e-16) is omitted."""
    xn = x - np.mean(x, axis=-1, keepdims=True)
    xn = xn / np.sqrt(np.sum(np.square(xn), axis=-1, keepdims=True))
    xn = xn - np.mean(xn, axis=1, keepdims=True)
    xn = xn / np.sqrt(np.sum(np.square(xn), axis=1, keepdims=True))
    return xn


def estoi(x, y, fs_sig):
    """pystoi.stoi(x, y, fs_sig, extended=True); x = clean, y = processed."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    assert x.shape == y.shape
    if fs_sig != FS:
        x = resample_oct(x, FS, fs_sig)
        y = resample_oct(y, FS, fs_sig)
    x, y, _ = remove_silent_frames(x, y, DYN_RANGE, N_FRAME, N_FRAME // 2)
    obm = thirdoct(FS, NFFT, NUMBAND, MINFREQ)[0]
    x_tob, y_tob = stft_tob(x, obm), stft_tob(y, obm)
    M = x_tob.shape[1]
    if M < N:
        return 1e-5            # pystoi warns "Not enough STFT frames" and returns 1e-5
    xs = np.array([x_tob[:, m - N:m] for m in range(N, M + 1)])
    ys = np.array([y_tob[:, m - N:m] for m in range(N, M + 1)])
    xn, yn = row_col_normalize(xs), row_col_normalize(ys)
    return float(np.sum(xn * yn / N) / xn.shape[0])


def sdr(ref, est, filter_length=512, clamp_db=50.0):
    """fast_bss_eval.bss_eval_sources for one source: SDR in dB (SIR = inf, SAR = SDR)."""
    ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    est = np.asarray(est, dtype=np.float64).reshape(-1)
    ref = ref / max(np.linalg.norm(ref), 1e-6)
    est = est / max(np.linalg.norm(est), 1e-6)
    n = 2 ** int(np.ceil(np.log2(len(ref) + filter_length - 1)))
    R, E = np.fft.rfft(ref, n), np.fft.rfft(est, n)
    acf = np.fft.irfft(R.conj() * R, n)[:filter_length]
    xcorr = np.fft.irfft(R.conj() * E, n)[:filter_length]
    idx = np.abs(np.arange(filter_length)[:, None] - np.arange(filter_length)[None, :])
    sol = np.linalg.solve(acf[idx], xcorr)
    coh = float(np.dot(xcorr, sol))
    e = 10.0 ** (-clamp_db / 10.0)
    eps = e / (1.0 + e)
    coh = min(max(coh, eps), 1.0 - eps)
    return 10.0 * np.log10(coh / (1.0 - coh))


# ---- soxr HQ specification resampler (stand-in for soxr.resample / librosa res_type="soxr_hq") -----------------------------------
def soxr_hq_design(fs_in, fs_out):
    """One Kaiser-windowed sinc built to the specification of libsoxr's HQ recipe (soxr.c soxr_quality_spec, quality 4: 20-bit =
    120.4 dB rejection from the lower Nyquist frequency, pass band to 1 - 0.05 / TO_3dB(rej) = 0.9136 of it, linear phase,
    cutoff mid-transition as lsx_design_lpf places it).  -> (h with unity DC gain, up, down).  soxr's own cascade is not reproducible
    bit-wise; what is pinned here is the specification, checked in tests/test_oracle.py."""
    from math import gcd
    from scipy.signal import firwin
    g = gcd(int(fs_in), int(fs_out))
    up, down = int(fs_out) // g, int(fs_in) // g
    rej = 20 * 20.0 * np.log10(2.0)
    to3db = (1.6e-6 * rej - 7.5e-4) * rej + 0.646
    nyq = 0.5 * min(fs_in, fs_out)
    f_pass, f_stop = (1.0 - 0.05 / to3db) * nyq, nyq
    fs_work = float(fs_in) * up
    dw = 2.0 * np.pi * (f_stop - f_pass) / fs_work
    att = rej + 2.0          # design margin: Kaiser's length estimate lands 1.3 dB short at the very band edges for the 80 k-tap filters
    half = int(np.ceil((att - 7.95) / (2.285 * dw) / 2.0))
    h = firwin(2 * half + 1, (f_pass + f_stop) / fs_work, window=("kaiser", 0.1102 * (att - 8.7)))
    return h, up, down


def resample_soxr_hq_spec(x, fs_in, fs_out):
    h, up, down = soxr_hq_design(fs_in, fs_out)
    return resample_poly(np.asarray(x, dtype=np.float64), up, down, window=h)     # (scipy scales the taps by `up` itself)
