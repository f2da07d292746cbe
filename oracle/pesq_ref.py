"""Oracle: PESQ (ITU-T P.862, P.862.1 / P.862.2 mappings) as ``pesq.pesq(fs, ref, deg, mode)`` computes it.

TEST INFRASTRUCTURE - never imported by the product path.

Reference call site: ``evaluation_metrics/calculate_intrusive_se_metrics.py:52-88`` (``pesq_metric``: 'nb' at 8 kHz, 'wb' at
16 kHz, > 16 kHz resampled to 16 kHz first; ``PesqError.RETURN_VALUES``: NO_UTTERANCES_DETECTED -> None).  The arithmetic
lives in ``pesq==0.0.4`` (Cython over the ITU-T P.862 reference C code), which is neither in the image nor under
/root/reference: PARITY UNPINNED.  This file restates the published algorithm stage by stage (names follow the reference C
code: ``fix_power_level``, ``apply_filter``, ``input_filter``, ``apply_VAD``, ``crude_align``, ``id_searchwindows``,
``time_align``, ``id_utterances``, ``utterance_split`` / ``split_align``, ``pesq_psychoacoustic_model``); the tables and what
could not be restated digit for digit are in ``pesq_tables.py`` (8 kHz tables: verified by their internal redundancy; 16 kHz:
the seven bands above 4 kHz reconstructed).  Arithmetic is float64 where the C code is float32 (``precision="f32"`` rounds every
stored buffer to float32 as the C code's `float` arrays do, see `q`): the INTEGER outputs of the
alignment stages (crude delay, utterance boundaries, per-utterance delays, bad intervals, frame counts) are what the GPU
path must reproduce exactly (``trace`` returns them), the MOS to floating-point tolerance.

Wrapper behaviour restated from the Python package: float signals are put on the 16-bit scale (x 32768, both
divided by their common peak first when it exceeds 1); fs must be 8000 or 16000; 'nb' uses the IRS receive filter and the
P.862.1 mapping, 'wb' the P.862.2 IIR input filter and mapping.  (The level alignment that follows makes the result
independent of that scale.)
"""
import numpy as np

from . import pesq_tables as T

SEARCHBUFFER = 75
DATAPADDING_MSECS = 320
MAXNUTTERANCES = 50
MINSPEECHLGTH = 4
JOINSPEECHLGTH = 50
MINUTTLENGTH = 50
TARGET_AVG_POWER = 1e7
CRITERIUM_FOR_SILENCE_OF_5_SAMPLES = 500.0
NUMBER_OF_PSQM_FRAMES_PER_SYLLABE = 20
D_POW_F, D_POW_S, D_POW_T = 2.0, 6.0, 2.0
A_POW_F, A_POW_S, A_POW_T = 1.0, 6.0, 2.0
D_WEIGHT, A_WEIGHT = 0.1, 0.0309
THRESHOLD_BAD_FRAMES = 30.0
ZWICKER_POWER = 0.23
NO_UTTERANCES_DETECTED = -1

# Storage precision.  The ITU-T reference code keeps every buffer (signals, VAD, correlations, histograms, Bark densities,
# loudness, frame disturbances) in C `float`.  `q` rounds an array or scalar to float32 (kept in a float64 container) at
# every point where that code stores into such a buffer when PRECISION == "f32"; with "f64" (default) it is the identity.
# Expression-level arithmetic stays float64 in both: the f32 variant restates the code's STORAGE rounding - 1e-7 relative
# perturbations at ~25 points per pair - which is what can move an integer decision (an argmax, a threshold crossing).
# tests/test_pesq_cpu.py requires both variants to give the same integers on every seeded pair, tests/test_pesq_gpu.py
# requires the kernels to equal both.
PRECISION = "f64"


def q(a):
    if PRECISION == "f64":
        return a
    if np.isscalar(a):
        return float(np.float32(a))
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def nextpow2(x):
    n = 1
    while n < x:
        n *= 2
    return n


TABLE_VARIANT = {}        # scripts/pesq_band_sweep.py: parameters of the reconstructed part of the 16 kHz Bark table


class Ctx:
    def __init__(self, fs, mode):
        assert fs in (8000, 16000) and mode in ("nb", "wb") and not (mode == "wb" and fs != 16000)
        self.fs, self.mode = fs, mode
        self.ds = 32 if fs == 8000 else 64                 # Downsample
        self.align_nfft = 512 if fs == 8000 else 1024
        self.pad = DATAPADDING_MSECS * (fs // 1000)
        self.iir = T.INIIR_HSOS_8K if fs == 8000 else T.INIIR_HSOS_16K
        self.wb_iir = T.WB_INIIR_HSOS_8K if fs == 8000 else T.WB_INIIR_HSOS_16K
        self.tb = T.tables(fs, **TABLE_VARIANT) if (TABLE_VARIANT and fs == 16000) else T.tables(fs)


def interpolate(freq, curve):
    c = np.asarray(curve, dtype=np.float64)
    f, g = c[:, 0], c[:, 1]
    if freq <= f[0]:
        lo, hi = 0, 1
    elif freq >= f[-1]:
        lo, hi = len(f) - 2, len(f) - 1
    else:
        hi = 1
        while f[hi] < freq:
            hi += 1
        lo = hi - 1
    return ((freq - f[lo]) * g[hi] + (f[hi] - freq) * g[lo]) / (f[hi] - f[lo])


def apply_filter(c, data, nsamples, curve):
    """FFT-domain filter over the whole signal (zero-padded to a power of two), gain relative to 1 kHz."""
    sb = SEARCHBUFFER * c.ds
    n = nsamples - 2 * sb + c.pad
    p2 = nextpow2(n)
    x = np.zeros(p2)
    x[:n] = data[sb:sb + n]
    X = np.fft.rfft(x)
    ref_gain = interpolate(1000.0, curve)
    res = c.fs / p2
    fac = np.array([10.0 ** ((interpolate(i * res, curve) - ref_gain) / 20.0) for i in range(p2 // 2 + 1)])
    y = np.fft.irfft(X * fac, p2)
    data[sb:sb + n] = q(y[:n])


def iir_sos(x, sos):
    """cascade of direct-form-II biquads {b0, b1, b2, a1, a2}, in place."""
    from scipy.signal import lfilter
    for b0, b1, b2, a1, a2 in sos:
        x[:] = q(lfilter([b0, b1, b2], [1.0, a1, a2], x))


def pow_of(x, start, stop, divisor):
    seg = np.asarray(x[start:stop], dtype=np.float64)
    return float(np.dot(seg, seg)) / divisor


def fix_power_level(c, data, nsamples, max_nsamples):
    sb = SEARCHBUFFER * c.ds
    tmp = data.copy()
    apply_filter(c, tmp, nsamples, T.ALIGN_FILTER_DB)
    p = pow_of(tmp, sb, nsamples - sb + c.pad, max_nsamples - 2 * sb + c.pad)
    if p > 0.0:                      # (an all-zero signal: the C code scales by inf; here it is left alone)
        data[:nsamples] = q(data[:nsamples] * q(np.sqrt(TARGET_AVG_POWER / p)))


def dc_block(c, data, nsamples):
    ofs = SEARCHBUFFER * c.ds
    seg = data[ofs:nsamples - ofs]
    seg -= seg.sum() / nsamples
    ramp = (0.5 + np.arange(c.ds)) / c.ds
    data[ofs:ofs + c.ds] *= ramp
    data[nsamples - ofs - c.ds:nsamples - ofs] *= ramp[::-1]
    data[:] = q(data)


def apply_vad(c, data, nsamples):
    ds = c.ds
    nw = nsamples // ds
    vad = q((data[:nw * ds].reshape(nw, ds) ** 2).sum(1) / ds)
    level_thresh = vad.sum() / nw
    level_min = vad.max()
    level_min = level_min * 1.0e-4 if level_min > 0.0 else 1.0
    vad = np.maximum(vad, level_min)
    for _ in range(12):
        sel = vad <= level_thresh
        level_noise, std_noise = 0.0, 0.0
        if sel.any():
            level_noise = vad[sel].mean()
            std_noise = np.sqrt(((vad[sel] - level_noise) ** 2).sum() / sel.sum())
        level_thresh = 1.001 * (level_noise + 2.0 * std_noise)
    above = vad > level_thresh
    length = int(above.sum())
    level_sig = vad[above].sum() / length if length > 0 else 0.0
    if length == 0:
        level_thresh = -1.0
    level_noise = vad[~above].sum() / (nw - length) if length < nw else 1.0
    vad = np.where(vad <= level_thresh, -vad, vad)
    vad[0] = vad[nw - 1] = -level_min
    start = finish = 0
    for i in range(1, nw):
        if vad[i] > 0.0 and vad[i - 1] <= 0.0:
            start = i
        if vad[i] <= 0.0 and vad[i - 1] > 0.0:
            finish = i
            if finish - start <= MINSPEECHLGTH:
                vad[start:finish] = -vad[start:finish]
    if level_sig >= level_noise * 1000.0:
        for i in range(1, nw):
            if vad[i] > 0.0 and vad[i - 1] <= 0.0:
                start = i
            if vad[i] <= 0.0 and vad[i - 1] > 0.0:
                finish = i
                if vad[start:finish].sum() < 3.0 * level_thresh * (finish - start):
                    vad[start:finish] = -vad[start:finish]
    start = finish = 0
    for i in range(1, nw):
        if vad[i] > 0.0 and vad[i - 1] <= 0.0:
            start = i
            if finish > 0 and start - finish <= JOINSPEECHLGTH:
                vad[finish:start] = level_min
        if vad[i] <= 0.0 and vad[i - 1] > 0.0:
            finish = i
    start = 0
    for i in range(1, nw):
        if vad[i] > 0.0 and vad[i - 1] <= 0.0:
            start = i
    if start == 0:
        vad = np.abs(vad)
        vad[0] = vad[nw - 1] = -level_min
    i = 3
    while i < nw - 2:
        if vad[i] > 0.0 and vad[i - 2] <= 0.0:
            vad[i - 2] = vad[i] * 0.1
            vad[i - 1] = vad[i] * 0.3
            i += 1
        if vad[i] <= 0.0 and vad[i - 1] > 0.0:
            vad[i] = vad[i - 1] * 0.3
            vad[i + 1] = vad[i - 1] * 0.1
            i += 3
        i += 1
    vad = np.maximum(vad, 0.0)
    if level_thresh <= 0.0:
        level_thresh = level_min
    logvad = np.where(vad <= level_thresh, 0.0, np.log(np.maximum(vad, 1e-300) / level_thresh))
    return q(vad), q(logvad)


def fftn_xcorr(x1, x2):
    """y[k] = sum_i x1[i] x2[k - (n1 - 1) + i], k = 0 .. n1 + n2 - 2 (FFT of twice the next power of two)."""
    n1, n2 = len(x1), len(x2)
    nx = nextpow2(max(n1, n2))
    a = np.fft.rfft(x1[::-1], 2 * nx)
    b = np.fft.rfft(x2, 2 * nx)
    return q(np.fft.irfft(a * b, 2 * nx)[:n1 + n2 - 1])


class Err:
    def __init__(self):
        self.nutt = 0
        self.crude_delay = 0
        self.search_start = [0] * MAXNUTTERANCES
        self.search_end = [0] * MAXNUTTERANCES
        self.delay_est = [0] * MAXNUTTERANCES
        self.delay = [0] * MAXNUTTERANCES
        self.delay_conf = [0.0] * MAXNUTTERANCES
        self.start = [0] * MAXNUTTERANCES
        self.end = [0] * MAXNUTTERANCES


def _cdiv(a, b):
    """C integer division (truncation towards zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def crude_align(c, ref, deg, e, utt_id):
    ds = c.ds
    nd_all = deg["n"] // ds
    if utt_id == -1:
        nr, nd, startr, startd = ref["n"] // ds, nd_all, 0, 0
    else:
        if utt_id == MAXNUTTERANCES:
            est, k = e.delay_est[MAXNUTTERANCES - 1], MAXNUTTERANCES - 1
        else:
            est, k = e.crude_delay, utt_id
        startr = e.search_start[k]
        startd = startr + _cdiv(est, ds)
        if startd < 0:
            startr = _cdiv(-est, ds)
            startd = 0
        nr = e.search_end[k] - startr
        nd = nr
        if startd + nd > nd_all:
            nd = nd_all - startd
    i_max, mx = nr - 1, 0.0
    if nr > 1 and nd > 1:
        y = fftn_xcorr(ref["logvad"][startr:startr + nr], deg["logvad"][startd:startd + nd])
        k = int(np.argmax(y))
        if y[k] > 0.0:
            i_max, mx = k, y[k]
    lag = (i_max - nr + 1) * ds
    if utt_id == -1:
        e.crude_delay = lag
    elif utt_id == MAXNUTTERANCES:
        e.delay[MAXNUTTERANCES - 1] = lag + e.delay_est[MAXNUTTERANCES - 1]
    else:
        e.delay_est[utt_id] = lag + e.crude_delay


def _utt_scan(c, ref, deg, e, on_start, on_end):
    ds = c.ds
    vad = ref["vad"]
    n = ref["n"] // ds
    del_deg_start = MINUTTLENGTH - _cdiv(e.crude_delay, ds)
    del_deg_end = _cdiv(deg["n"] - e.crude_delay, ds) - MINUTTLENGTH
    num, flag, this_start = 0, 0, 0
    for i in range(n):
        v = vad[i]
        if v > 0.0 and flag == 0:
            flag, this_start = 1, i
            on_start(num, i)
        if (v == 0.0 or i == n - 1) and flag == 1:
            flag = 0
            on_end(num, i, n)
            if i - this_start >= MINUTTLENGTH and this_start < del_deg_end and i > del_deg_start:
                num += 1
                if num >= MAXNUTTERANCES - 1:       # the arrays hold MAXNUTTERANCES entries, the last is scratch
                    break
    return num


def id_searchwindows(c, ref, deg, e):
    def on_start(k, i):
        e.search_start[k] = max(i - SEARCHBUFFER, 0)

    def on_end(k, i, n):
        e.search_end[k] = min(i + SEARCHBUFFER, n - 1)
    e.nutt = _utt_scan(c, ref, deg, e, on_start, on_end)


def _frame_xcorr_hist(c, ref, deg, startr, startd, window):
    n = c.align_nfft
    x1 = np.fft.rfft(ref["data"][startr:startr + n] * window)
    x2 = np.fft.rfft(deg["data"][startd:startd + n] * window)
    x = q(np.abs(np.fft.irfft(np.conj(x1) * x2, n)))
    v_max = q(x.max() * 0.99)
    return x, v_max


def time_align(c, ref, deg, e, utt_id):
    n, ds = c.align_nfft, c.ds
    est = e.delay_est[utt_id]
    window = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(n) / n))
    h = np.zeros(n)
    startr = e.search_start[utt_id] * ds
    startd = startr + est
    if startd < 0:
        startr, startd = -est, 0
    while startd + n <= deg["n"] and startr + n // 4 <= e.search_end[utt_id] * ds:
        x, v_max = _frame_xcorr_hist(c, ref, deg, startr, startd, window)
        h[x > v_max] = q(h[x > v_max] + q(v_max ** 0.125))
        startr += n // 4
        startd += n // 4
    hsum = h.sum()
    kernel = n // 64
    x2 = np.zeros(n)
    x2[0] = 1.0
    for k in range(1, kernel):
        x2[k] = x2[n - k] = 1.0 - k / kernel
    sm = np.abs(np.fft.irfft(np.fft.rfft(h) * np.fft.rfft(x2), n))
    hh = q(sm / hsum) if hsum > 0.0 else np.zeros(n)
    i_max = int(np.argmax(hh))
    v_max = hh[i_max]
    if v_max <= 0.0:
        i_max, v_max = 0, 0.0
    if i_max >= n // 2:
        i_max -= n
    e.delay[utt_id] = est + i_max
    e.delay_conf[utt_id] = float(v_max)


def id_utterances(c, ref, deg, e):
    ds = c.ds

    def on_start(k, i):
        e.start[k] = i

    def on_end(k, i, n):
        e.end[k] = i
    _utt_scan(c, ref, deg, e, on_start, on_end)
    n = ref["n"] // ds
    nu = e.nutt
    e.start[0] = SEARCHBUFFER
    e.end[nu - 1] = n - SEARCHBUFFER
    for k in range(1, nu):
        mid = (e.start[k] + e.end[k - 1]) // 2
        e.start[k] = e.end[k - 1] = mid
    this_start = e.start[0] * ds + e.delay[0]
    if this_start < SEARCHBUFFER * ds:
        e.start[0] = SEARCHBUFFER + _cdiv(ds - 1 - e.delay[0], ds)
    last_end = e.end[nu - 1] * ds + e.delay[nu - 1]
    if last_end > deg["n"] - SEARCHBUFFER * ds:
        e.end[nu - 1] = _cdiv(deg["n"] - e.delay[nu - 1], ds) - SEARCHBUFFER
    for k in range(1, nu):
        this_start = e.start[k] * ds + e.delay[k]
        last_end = e.end[k - 1] * ds + e.delay[k - 1]
        if this_start < last_end:
            mid = _cdiv(this_start + last_end, 2)
            e.start[k] = _cdiv(ds - 1 + mid - e.delay[k], ds)
            e.end[k - 1] = _cdiv(mid - e.delay[k - 1], ds)


def split_align(c, ref, deg, e, utt_start, speech_start, speech_end, utt_end, delay_est, delay_conf):
    n, ds = c.align_nfft, c.ds
    utt_len = speech_end - speech_start
    test = MAXNUTTERANCES - 1
    window = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(n) / n))
    kernel = n // 64
    delta = n // (4 * ds)
    step = int((0.801 * utt_len + 40 * delta - 1) / (40 * delta)) * delta
    pad = max(utt_len // 10, 75)
    bps = [speech_start + pad]
    while True:
        bps.append(bps[-1] + step)
        if not (bps[-1] <= speech_end - pad and len(bps) - 1 < 40):
            break
    nb = len(bps) - 1
    best = dict(dc1=0.0, dc2=0.0)
    if nb <= 0:
        return best
    ed1, ed2 = [0] * nb, [0] * nb
    for bp in range(nb):
        e.delay_est[test], e.search_start[test], e.search_end[test] = delay_est, utt_start, bps[bp]
        crude_align(c, ref, deg, e, MAXNUTTERANCES)
        ed1[bp] = e.delay[test]
        e.delay_est[test], e.search_start[test], e.search_end[test] = delay_est, bps[bp], utt_end
        crude_align(c, ref, deg, e, MAXNUTTERANCES)
        ed2[bp] = e.delay[test]
    tri = np.array([kernel - abs(k) for k in range(1 - kernel, kernel)], dtype=np.float64)

    def accumulate(h, startr, startd):
        x, v_max = _frame_xcorr_hist(c, ref, deg, startr, startd, window)
        n_max = q(v_max ** 0.125 / kernel)
        add = 0.0
        for cnt in np.nonzero(x > v_max)[0]:
            add += n_max * kernel
            idx = (cnt + np.arange(1 - kernel, kernel) + n) % n
            np.add.at(h, idx, n_max * tri)
            h[idx] = q(h[idx])
        return add

    def peak(h, hsum, est):
        i_max = int(np.argmax(h))
        v_max = h[i_max]
        if v_max <= 0.0:
            i_max, v_max = 0, 0.0
        if i_max >= n // 2:
            i_max -= n
        return est + i_max, (v_max / hsum if hsum > 0.0 else 0.0)
    d1, dc1 = [0] * nb, [-2.0] * nb
    while True:
        bp = 0
        while bp < nb and dc1[bp] > -2.0:
            bp += 1
        if bp >= nb:
            break
        est = ed1[bp]
        h, hsum = np.zeros(n), 0.0
        startr = utt_start * ds
        startd = startr + est
        if startd < 0:
            startr, startd = -est, 0
        while True:
            while startd + n <= deg["n"] and startr + n // 4 <= bps[bp] * ds:
                hsum += accumulate(h, startr, startd)
                startr += n // 4
                startd += n // 4
            d1[bp], dc1[bp] = peak(h, hsum, est)
            nxt = None
            while bp < nb - 1:
                bp += 1
                if ed1[bp] == est and dc1[bp] <= -2.0:
                    nxt = bp
                    break
            if nxt is None:
                break
    d2 = [0] * nb
    dc2 = [(-2.0 if dc1[bp] > delay_conf else 0.0) for bp in range(nb)]
    while True:
        bp = nb - 1
        while bp >= 0 and dc2[bp] > -2.0:
            bp -= 1
        if bp < 0:
            break
        est = ed2[bp]
        h, hsum = np.zeros(n), 0.0
        startr = utt_end * ds - n
        startd = startr + est
        if startd + n > deg["n"]:
            startd = deg["n"] - n
            startr = startd - est
        while True:
            while startd >= 0 and startr + n * 3 // 4 >= bps[bp] * ds:
                hsum += accumulate(h, startr, startd)
                startr -= n // 4
                startd -= n // 4
            d2[bp], dc2[bp] = peak(h, hsum, est)
            nxt = None
            while bp > 0:
                bp -= 1
                if ed2[bp] == est and dc2[bp] <= -2.0:
                    nxt = bp
                    break
            if nxt is None:
                break
    for bp in range(nb):
        if abs(d2[bp] - d1[bp]) >= ds and dc1[bp] + dc2[bp] > best["dc1"] + best["dc2"] and dc1[bp] > delay_conf and \
                dc2[bp] > delay_conf:
            best = dict(ed1=ed1[bp], d1=d1[bp], dc1=dc1[bp], ed2=ed2[bp], d2=d2[bp], dc2=dc2[bp], bp=bps[bp])
    return best


def utterance_split(c, ref, deg, e):
    ds = c.ds
    vad = ref["vad"]
    k = 0
    while k < e.nutt and e.nutt < MAXNUTTERANCES:
        us, ue, conf = e.start[k], e.end[k], e.delay_conf[k]
        ss = us
        while ss < ue and vad[ss] <= 0.0:
            ss += 1
        se = ue
        while se > us and vad[se] <= 0.0:
            se -= 1
        se += 1
        if se - ss >= 200:
            b = split_align(c, ref, deg, e, us, ss, se, ue, e.delay_est[k], conf)
            if b["dc1"] > conf and b["dc2"] > conf:
                for s in range(e.nutt - 1, k, -1):
                    e.delay_est[s + 1], e.delay[s + 1], e.delay_conf[s + 1] = e.delay_est[s], e.delay[s], e.delay_conf[s]
                    e.start[s + 1], e.end[s + 1] = e.start[s], e.end[s]
                    e.search_start[s + 1], e.search_end[s + 1] = e.start[s], e.end[s]
                e.nutt += 1
                e.delay_est[k], e.delay[k], e.delay_conf[k] = b["ed1"], b["d1"], b["dc1"]
                e.delay_est[k + 1], e.delay[k + 1], e.delay_conf[k + 1] = b["ed2"], b["d2"], b["dc2"]
                e.search_start[k + 1], e.search_end[k + 1] = e.search_start[k], e.search_end[k]
                if b["d2"] < b["d1"]:
                    e.start[k], e.end[k], e.start[k + 1], e.end[k + 1] = us, b["bp"], b["bp"], ue
                else:
                    half = _cdiv(b["d2"] - b["d1"], 2 * ds)
                    e.start[k], e.end[k], e.start[k + 1], e.end[k + 1] = us, b["bp"] + half, b["bp"] - half, ue
                if (e.start[k] - SEARCHBUFFER) * ds + b["d1"] < 0:
                    e.start[k] = SEARCHBUFFER + _cdiv(ds - 1 - b["d1"], ds)
                if e.end[k + 1] * ds + b["d2"] > deg["n"] - SEARCHBUFFER * ds:
                    e.end[k + 1] = _cdiv(deg["n"] - b["d2"], ds) - SEARCHBUFFER
                continue
        k += 1


# ---- perceptual model ------------------------------------------------------------------------------------------------
def _pitch_pow_dens(c, data, start, whanning):
    tb = c.tb
    nf = 8 * c.ds
    spec = np.abs(np.fft.rfft(data[start:start + nf] * whanning)[:nf // 2]) ** 2
    spec[0] = 0.0
    edges = np.concatenate([[0], np.cumsum(tb["nr"])])
    out = np.add.reduceat(q(spec), edges[:-1])
    return q(out * tb["pow_corr"] * tb["sp"])


def _total_audible(c, ppd, factor):
    th = factor * c.tb["abs_thresh"][1:]
    h = ppd[1:]
    return float(h[h > th].sum())


def _loudness(c, ppd):
    tb = c.tb
    cb = tb["centre_bark"]
    h = np.where(cb < 4.0, 6.0 / (cb + 2.0), 1.0)
    h = np.minimum(h, 2.0) ** 0.15
    zp = ZWICKER_POWER * h
    th = tb["abs_thresh"]
    ld = np.where(ppd > th, (th / 0.5) ** zp * ((0.5 + 0.5 * ppd / th) ** zp - 1.0), 0.0)
    return q(ld * tb["sl"])


def _pseudo_lp(c, x, p):
    w = c.tb["width_bark"][1:]
    tot = w.sum()
    r = ((np.abs(x[1:]) * w) ** p).sum() / tot
    return r ** (1.0 / p) * tot


def _disturbances(c, ppd_ref, ppd_deg):
    lr, ld = _loudness(c, ppd_ref), _loudness(c, ppd_deg)
    d = ld - lr
    m = 0.25 * np.minimum(ld, lr)
    d = np.where(d > m, d - m, np.where(d < -m, d + m, 0.0))
    fd = _pseudo_lp(c, d, D_POW_F)
    ratio = (ppd_deg + 50.0) / (ppd_ref + 50.0)
    h = ratio ** 1.2
    h = np.where(h > 12.0, 12.0, h)
    h = np.where(h < 3.0, 0.0, h)
    return q(fd), q(_pseudo_lp(c, q(d * h), A_POW_F))


def _lpq_weight(start_frame, stop_frame, p_syl, p_time, fd, tw):
    res, tot = 0.0, 0.0
    s = start_frame
    while s <= stop_frame:
        seg = fd[s:min(s + NUMBER_OF_PSQM_FRAMES_PER_SYLLABE, stop_frame + 1)]
        r = ((seg ** p_syl).sum() / NUMBER_OF_PSQM_FRAMES_PER_SYLLABE) ** (1.0 / p_syl)
        res += (tw[s - start_frame] * r) ** p_time
        tot += tw[s - start_frame] ** p_time
        s += NUMBER_OF_PSQM_FRAMES_PER_SYLLABE // 2
    return (res / tot) ** (1.0 / p_time)


def compute_delay(start, stop, search_range, s1, s2):
    n = stop - start
    p2 = nextpow2(2 * n)
    pw1 = pow_of(s1, start, stop, stop - start) * n / p2
    pw2 = pow_of(s2, start, stop, stop - start) * n / p2
    if pw1 <= 1e-6 or pw2 <= 1e-6:
        return 0, 0.0
    norm = np.sqrt(pw1 * pw2)
    x1 = np.fft.rfft(np.abs(s1[start:stop]), p2) / p2
    x2 = np.fft.rfft(np.abs(s2[start:stop]), p2)
    y = q(np.fft.irfft(np.conj(x1) * x2, p2))
    best, mx = 0, 0.0
    for i in list(range(-search_range, 0)) + list(range(0, search_range)):
        h = abs(y[i % p2]) / norm
        if h > mx:
            mx, best = h, i
    return best, mx


def psychoacoustic_model(c, ref, deg, e, trace):
    ds, fs, tb = c.ds, c.fs, c.tb
    nb = tb["nb"]
    sb = SEARCHBUFFER * ds
    maxn = max(ref["n"], deg["n"])
    nf = 8 * ds
    half = nf // 2
    whanning = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(nf) / nf))
    rd, dd = ref["data"], deg["data"]
    skip_start = 0
    while True:
        s5 = np.abs(rd[sb + skip_start:sb + skip_start + 5]).sum()
        if s5 < CRITERIUM_FOR_SILENCE_OF_5_SAMPLES:
            skip_start += 1
        if not (s5 < CRITERIUM_FOR_SILENCE_OF_5_SAMPLES and skip_start < maxn // 2):
            break
    skip_end = 0
    while True:
        hi = maxn - sb + c.pad - 1 - skip_end
        s5 = np.abs(rd[hi - 4:hi + 1]).sum()
        if s5 < CRITERIUM_FOR_SILENCE_OF_5_SAMPLES:
            skip_end += 1
        if not (s5 < CRITERIUM_FOR_SILENCE_OF_5_SAMPLES and skip_end < maxn // 2):
            break
    start_frame = skip_start // half
    total_frames = (maxn - 2 * sb + c.pad) // half - 1
    stop_frame = total_frames - skip_end // half
    nfr = stop_frame + 1
    ppd_ref, ppd_deg = np.zeros((nfr, nb)), np.zeros((nfr, nb))
    silent = np.zeros(nfr, dtype=bool)

    def delay_at(sample):
        u = e.nutt - 1
        while u >= 0 and e.start[u] * ds > sample:
            u -= 1
        return e.delay[u] if u >= 0 else e.delay[0]
    for f in range(nfr):
        s_ref = sb + f * half
        ppd_ref[f] = _pitch_pow_dens(c, rd, s_ref, whanning)
        s_deg = s_ref + delay_at(s_ref)
        if s_deg > 0 and s_deg + nf < maxn + c.pad:
            ppd_deg[f] = _pitch_pow_dens(c, dd, s_deg, whanning)
        silent[f] = _total_audible(c, ppd_ref[f], 1e2) < 1e7
    th100 = 100.0 * tb["abs_thresh"]
    act = ~silent
    avg_ref = np.where(ppd_ref[act] > th100, ppd_ref[act], 0.0).sum(0) / total_frames
    avg_deg = np.where(ppd_deg[act] > th100, ppd_deg[act], 0.0).sum(0) / total_frames
    x = np.clip((avg_deg + 1000.0) / (avg_ref + 1000.0), 0.01, 100.0)
    ppd_ref[:] = q(ppd_ref * q(x)[None, :])
    fd, fda = np.zeros(nfr), np.zeros(nfr)
    total_power_ref = np.zeros(nfr)

    def scale_and_disturb(frames, first_pass):
        old = 1.0
        for f in frames:
            ta_ref, ta_deg = _total_audible(c, ppd_ref[f], 1.0), _total_audible(c, ppd_deg[f], 1.0)
            if first_pass:
                total_power_ref[f] = ta_ref
            sc = (ta_ref + 5e3) / (ta_deg + 5e3)
            if f > 0:
                sc = 0.2 * old + 0.8 * sc
            old = sc
            sc = min(max(sc, 3e-4), 5.0)
            ppd_deg[f] = q(ppd_deg[f] * q(sc))
            a, b = _disturbances(c, ppd_ref[f], ppd_deg[f])
            if first_pass:
                fd[f], fda[f] = a, b
            else:
                fd[f], fda[f] = min(fd[f], a), min(fda[f], b)
    scale_and_disturb(range(nfr), True)
    bad_frame = bool((fd > THRESHOLD_BAD_FRAMES).any())
    skipped = np.zeros(nfr, dtype=bool)
    for u in range(1, e.nutt):
        frame1 = int(np.floor(((e.start[u] - SEARCHBUFFER) * ds + e.delay[u]) / half))
        j = int(np.floor((e.end[u - 1] - SEARCHBUFFER) * ds + e.delay[u - 1])) // half
        jump = e.delay[u] - e.delay[u - 1]
        frame1 = max(min(frame1, j), 0)
        if jump < -half:
            frame2 = int((e.start[u] - SEARCHBUFFER) * ds + max(0, abs(jump))) // half + 1
            for f in range(frame1, frame2 + 1):
                if f < stop_frame:
                    skipped[f] = True
                    fd[f] = fda[f] = 0.0
    nn = c.pad + maxn
    tweaked = np.zeros(nn)
    idx = np.arange(sb, nn - sb)
    starts = np.array([e.start[u] * ds for u in range(e.nutt)])
    which = np.searchsorted(starts, idx, side="right") - 1
    delays = np.array([e.delay[u] for u in range(e.nutt)])
    dl = np.where(which >= 0, delays[np.maximum(which, 0)], delays[0])
    j = np.clip(idx + dl, sb, nn - sb - 1)
    tweaked[idx] = dd[j]
    bad_intervals = []
    if bad_frame:
        is_bad = fd > THRESHOLD_BAD_FRAMES
        is_bad[0] = False
        smeared = np.zeros(nfr, dtype=bool)
        for f in range(2, stop_frame - 2):
            smeared[f] = is_bad[f - 2:f + 1].any() and is_bad[f:f + 3].any()
        f = 0
        while f <= stop_frame:
            while f <= stop_frame and not smeared[f]:
                f += 1
            if f <= stop_frame:
                st = f
                while f <= stop_frame and smeared[f]:
                    f += 1
                if f <= stop_frame and f - st >= 5:
                    bad_intervals.append([st, f])
        srange = 4 * nf
        for bi in bad_intervals:
            st_s = bi[0] * half + sb
            sp_s = bi[1] * half + nf + sb
            ns = sp_s - st_s
            r = np.zeros(2 * srange + ns)
            r[srange:srange + ns] = rd[st_s:st_s + ns]
            jj = np.clip(st_s - srange + np.arange(2 * srange + ns), sb, maxn - sb + c.pad - 1)
            d = tweaked[jj]
            dly, corr = compute_delay(0, 2 * srange + ns, srange, r, d)
            bi += [st_s, sp_s, dly if corr >= 0.5 else 0]
        if bad_intervals:
            doubly = tweaked[:maxn + c.pad].copy()
            for st, sp, st_s, sp_s, dly in bad_intervals:
                ii = np.arange(st_s, sp_s)
                doubly[ii] = tweaked[np.clip(ii + dly, 0, maxn - 1)]
            for st, sp, st_s, sp_s, dly in bad_intervals:
                for f in range(st, sp):
                    ppd_deg[f] = _pitch_pow_dens(c, doubly, sb + f * half, whanning)
                scale_and_disturb(range(st, sp), False)
    tw = np.ones(nfr)
    if nfr > 1000:
        n = (maxn - 2 * sb) // half - 1
        twf = min((n - 1000.0) / 5500.0, 0.5)
        tw = (1.0 - twf) + twf * np.arange(nfr) / n
    h = ((total_power_ref + 1e5) / 1e7) ** 0.04
    fd = q(np.minimum(fd / h, 45.0))
    fda = q(np.minimum(fda / h, 45.0))
    d_ind = _lpq_weight(start_frame, stop_frame, D_POW_S, D_POW_T, fd, tw)
    a_ind = _lpq_weight(start_frame, stop_frame, A_POW_S, A_POW_T, fda, tw)
    trace.update(start_frame=start_frame, stop_frame=stop_frame, bad_intervals=[list(map(int, b)) for b in bad_intervals],
                 d_indicator=d_ind, a_indicator=a_ind, n_skipped=int(skipped.sum()))
    return 4.5 - D_WEIGHT * d_ind - A_WEIGHT * a_ind


def _load(c, x, scale):
    """SIGNAL_INFO.data: SEARCHBUFFER * Downsample zeros, the samples on the 16-bit scale, then zeros (the package's
    wrapper multiplies float input by 32768)."""
    sb = SEARCHBUFFER * c.ds
    n = len(x) + 2 * sb
    data = np.zeros(n + c.pad + 4 * c.align_nfft)
    data[sb:sb + len(x)] = q(np.asarray(x, dtype=np.float64) * scale)
    return dict(data=data, n=n)


def pesq(fs, ref, deg, mode="wb", return_trace=False, precision="f64"):
    """-> MOS-LQO (P.862.1 for 'nb', P.862.2 for 'wb'), or NO_UTTERANCES_DETECTED; with return_trace also the integer
    outputs of the alignment stages.  precision "f32": buffers rounded to float32 where the ITU code stores C floats (`q`)."""
    global PRECISION
    old, PRECISION = PRECISION, precision
    try:
        return _pesq(fs, ref, deg, mode, return_trace)
    finally:
        PRECISION = old


def _pesq(fs, ref, deg, mode, return_trace):
    c = Ctx(fs, mode)
    peak = max(float(np.max(np.abs(ref))), float(np.max(np.abs(deg))), 1.0)
    r, d = _load(c, ref, 32768.0 / peak), _load(c, deg, 32768.0 / peak)
    maxn = max(r["n"], d["n"])
    fix_power_level(c, r["data"], r["n"], maxn)
    fix_power_level(c, d["data"], d["n"], maxn)
    for s in (r, d):
        if mode == "wb":
            seg = s["data"][:s["n"] + c.pad]
            iir_sos(seg, c.wb_iir)
        else:
            apply_filter(c, s["data"], s["n"], T.STANDARD_IRS_FILTER_DB)
    model = [s["data"].copy() for s in (r, d)]
    for s in (r, d):
        dc_block(c, s["data"], s["n"])
        seg = s["data"][:s["n"] + c.pad]
        iir_sos(seg, c.iir)
        s["vad"], s["logvad"] = apply_vad(c, s["data"], s["n"])
    e = Err()
    crude_align(c, r, d, e, -1)
    id_searchwindows(c, r, d, e)
    trace = dict(crude_delay=int(e.crude_delay), n_search_windows=e.nutt)
    if e.nutt < 1:
        return (NO_UTTERANCES_DETECTED, trace) if return_trace else NO_UTTERANCES_DETECTED
    for u in range(e.nutt):
        crude_align(c, r, d, e, u)
        time_align(c, r, d, e, u)
    trace["delay_est"] = [int(v) for v in e.delay_est[:e.nutt]]
    trace["delay_first"] = [int(v) for v in e.delay[:e.nutt]]
    id_utterances(c, r, d, e)
    utterance_split(c, r, d, e)
    trace.update(n_utterances=e.nutt, utt_start=[int(v) for v in e.start[:e.nutt]], utt_end=[int(v) for v in e.end[:e.nutt]],
                 utt_delay=[int(v) for v in e.delay[:e.nutt]])
    r["data"], d["data"] = model
    raw = psychoacoustic_model(c, r, d, e, trace)
    if mode == "nb":
        mos = 0.999 + 4.0 / (1.0 + np.exp(-1.4945 * raw + 4.6607))
    else:
        mos = 0.999 + 4.0 / (1.0 + np.exp(-1.3669 * raw + 3.8224))
    trace["raw"] = float(raw)
    return (float(mos), trace) if return_trace else float(mos)
