"""ITU-T P.862 / P.862.2 constant tables of the PESQ oracle (oracle/pesq_ref.py).

TEST INFRASTRUCTURE - never imported by the product path (the kernels carry their own copy in csrc/pesq_tables.h, written
by scripts/gen_pesq_tables.py from this module and committed).

PROVENANCE.  ``pesq==0.0.4`` (Cython over the ITU-T P.862 reference C code, the package behind
``evaluation_metrics/calculate_intrusive_se_metrics.py:9,76-86``) is not in the image and its source is not under
/root/reference; there is no network.  The tables below are RESTATED FROM THE PUBLISHED STANDARD'S REFERENCE CODE
(``pesqpar.h``).  They carry their own check: the standard tabulates, per Bark band, the band's centre and width in Bark and
in Hz, the number of FFT bins it collects and a power-density correction, and those columns are redundant -
``centre_bark = cumsum(width_bark) - width_bark / 2``, ``correction = width_hz / (width_bark * bins)``, and the bins
(31.25 Hz apart) falling between the cumulative Hz edges reproduce the bin counts.  ``check_redundancy`` asserts all of it to
the printed precision for the 42 narrow-band (8 kHz) bands: a mis-remembered digit breaks a relation.  PARITY: UNPINNED
against the package itself (absent), pinned by that redundancy for the 8 kHz tables.

16 kHz (P.862.2, 49 bands): bands 0-40 are the narrow-band ones.  For bands 41-48 the bin counts (12, 12, 15, 16, 18, 21, 25,
20: they sum, with the rest, to the 256 bins of the 512-point frame) and the absolute thresholds are restated; their Bark /
Hz widths could NOT be restated digit for digit and are RECONSTRUCTED here: Bark widths continue the smooth width sequence
(cubic through bands 20-41), Hz edges continue the Bark->Hz curve of the narrow-band edges and are held inside the
one-bin-wide interval the bin counts leave them, the last band ends at 8 kHz.  Wide-band scores therefore carry an
uncertainty of a few percent in the power-density correction of the seven bands above 4 kHz.  This is stated wherever a
wide-band PESQ number is reported.
"""
import numpy as np

SP_8K, SL_8K = 2.764344e-5, 1.866055e-1
SP_16K, SL_16K = 6.910853e-6, 1.866055e-1

NR_OF_HZ_BANDS_8K = [1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 2, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4, 3, 4, 5, 4, 5, 6, 6,
                     7, 8, 9, 9, 11]
CENTRE_OF_BAND_BARK_8K = [
    0.078672, 0.316341, 0.636559, 0.961246, 1.290450, 1.624217, 1.962597, 2.305636, 2.653383, 3.005889, 3.363201, 3.725371,
    4.092449, 4.464486, 4.841533, 5.223642, 5.610866, 6.003256, 6.400869, 6.803755, 7.211971, 7.625571, 8.044611, 8.469146,
    8.899232, 9.334927, 9.776288, 10.223374, 10.676242, 11.134952, 11.599563, 12.070135, 12.546731, 13.029408, 13.518232,
    14.013264, 14.514566, 15.022202, 15.536238, 16.056736, 16.583761, 17.117382]
CENTRE_OF_BAND_HZ_8K = [
    7.867213, 31.634144, 63.655895, 96.124611, 129.044968, 162.421738, 196.259659, 230.563568, 265.338348, 300.588867,
    336.320129, 372.537140, 409.244934, 446.448578, 484.568604, 526.600586, 570.303833, 619.423340, 672.121643, 728.525696,
    785.675964, 846.835693, 909.691650, 977.063293, 1049.861694, 1129.635986, 1217.257568, 1312.109497, 1412.501465,
    1517.999390, 1628.894165, 1746.194336, 1871.568848, 2008.776123, 2158.979248, 2326.743164, 2513.787109, 2722.488770,
    2952.586670, 3205.835449, 3492.679932, 3820.219238]
WIDTH_OF_BAND_BARK_8K = [
    0.157344, 0.317994, 0.322441, 0.326934, 0.331474, 0.336061, 0.340697, 0.345381, 0.350114, 0.354897, 0.359729, 0.364611,
    0.369544, 0.374529, 0.379565, 0.384653, 0.389794, 0.394989, 0.400236, 0.405538, 0.410894, 0.416306, 0.421773, 0.427297,
    0.432877, 0.438514, 0.444209, 0.449962, 0.455774, 0.461645, 0.467577, 0.473569, 0.479621, 0.485736, 0.491912, 0.498151,
    0.504454, 0.510819, 0.517250, 0.523745, 0.530308, 0.536934]
WIDTH_OF_BAND_HZ_8K = [
    15.734426, 31.799433, 32.244064, 32.693359, 33.147385, 33.606140, 34.069702, 34.538116, 35.011429, 35.489655, 35.972870,
    36.461121, 36.954407, 37.452911, 40.269653, 42.311859, 45.992554, 51.348511, 55.040527, 56.775208, 58.699402, 62.445862,
    64.820923, 69.195374, 76.745667, 84.016235, 90.825684, 97.931152, 103.348877, 107.801880, 113.552246, 121.490601,
    130.420410, 143.431763, 158.486816, 176.872803, 198.314697, 219.549561, 240.600098, 268.702393, 306.060059, 349.937012]
POW_DENS_CORRECTION_FACTOR_8K = [
    100.000000, 99.999992, 100.000000, 100.000008, 100.000008, 100.000015, 99.999992, 99.999969, 50.000027, 100.000000,
    99.999969, 100.000015, 99.999947, 100.000061, 53.047077, 110.000046, 117.991989, 65.000000, 68.760147, 69.999931,
    71.428818, 75.000038, 76.843384, 80.968781, 88.646126, 63.864388, 68.155350, 72.547775, 75.584831, 58.379192, 80.950836,
    64.135651, 54.384785, 73.821884, 64.437073, 59.176456, 65.521278, 61.399822, 58.144047, 57.004543, 64.126297, 59.248363]
ABS_THRESH_POWER_8K = [
    51286152.0, 2454709.500, 70794.593750, 4897.788574, 1174.897705, 389.045166, 104.712860, 45.708820, 17.782795, 9.772372,
    4.897789, 3.090296, 1.905461, 1.258925, 0.977237, 0.724436, 0.562341, 0.457088, 0.389045, 0.331131, 0.295121, 0.269153,
    0.257040, 0.251189, 0.251189, 0.251189, 0.251189, 0.263027, 0.288403, 0.309030, 0.338844, 0.371535, 0.398107, 0.436516,
    0.467735, 0.489779, 0.501187, 0.501187, 0.512861, 0.524807, 0.524807, 0.524807]

NR_OF_HZ_BANDS_16K_TAIL = [12, 12, 15, 16, 18, 21, 25, 20]                       # bands 41..48
ABS_THRESH_POWER_16K_TAIL = [0.524807, 0.512861, 0.478630, 0.426580, 0.371535, 0.363078, 0.416869, 0.537032]

# level alignment band-pass and IRS receive characteristic (dB over Hz): pesqio / pesqmain tables
ALIGN_FILTER_DB = [(0.0, -500), (50.0, -500), (100.0, -500), (125.0, -500), (160.0, -500), (200.0, -500), (250.0, -500),
                   (300.0, -500), (350.0, 0), (400.0, 0), (500.0, 0), (600.0, 0), (630.0, 0), (800.0, 0), (1000.0, 0),
                   (1250.0, 0), (1600.0, 0), (2000.0, 0), (2500.0, 0), (3000.0, 0), (3250.0, 0), (3500.0, -500),
                   (4000.0, -500), (5000.0, -500), (6300.0, -500), (8000.0, -500)]
STANDARD_IRS_FILTER_DB = [(0, -200), (50, -40), (100, -20), (125, -12), (160, -6), (200, 0), (250, 4), (300, 6), (350, 8),
                          (400, 10), (500, 11), (600, 12), (700, 12), (800, 12), (1000, 12), (1300, 12), (1600, 12),
                          (2000, 12), (2500, 12), (3000, 12), (3250, 12), (3500, 4), (4000, -200), (5000, -200),
                          (6300, -200), (8000, -200)]

# second-order sections {b0, b1, b2, a1, a2} of the alignment pre-filters and of the wide-band input filter
INIIR_HSOS_8K = [
    (0.885535424, -0.885535424, 0.000000000, -0.771070709, 0.000000000),
    (0.895092588, 1.292907193, 0.449260174, 1.268869037, 0.442025372),
    (4.049527940, -7.865190042, 3.815662102, -1.746859852, 0.786305963),
    (0.500002353, -0.500002353, 0.000000000, 0.000000000, 0.000000000),
    (0.565002834, -0.241585934, -0.306009671, 0.259688659, 0.249979657),
    (2.115237288, 0.919935084, 1.141240051, -1.587313419, 0.665935315),
    (0.912224584, -0.224397719, -0.641121413, -0.246029464, -0.556720590),
    (0.444617727, -0.307589321, 0.141638062, -0.996391149, 0.502251622)]
INIIR_HSOS_16K = [
    (0.325631521, -0.086782860, -0.238848661, -1.079416490, 0.434583902),
    (0.403961804, -0.556985881, 0.153024077, -0.415115835, 0.696590244),
    (4.736162769, 3.287251046, 1.753289019, -1.859599046, 0.876284034),
    (0.365373469, 0.000000000, 0.000000000, -0.634626531, 0.000000000),
    (0.884811506, 0.000000000, 0.000000000, -0.256725271, 0.141536777),
    (0.723593055, -1.447186099, 0.723593044, -1.129587469, 0.657232737),
    (1.644910855, -1.817280902, 1.249658063, -1.778403899, 0.801724355),
    (0.633692689, -0.284644314, -0.319789663, 0.000000000, 0.000000000),
    (1.032763031, 0.268428979, 0.602913323, 0.000000000, 0.000000000),
    (1.001616361, -0.823749013, 0.439731942, -0.885778255, 0.000000000),
    (0.752472096, -0.375388990, 0.188977609, -0.077258216, 0.247230734),
    (1.023700575, 0.001661628, 0.521284240, -0.183867259, 0.354324187)]
WB_INIIR_HSOS_16K = [(2.740826, -5.4816519, 2.740826, -1.9444777, 0.94597794)]
WB_INIIR_HSOS_8K = [(2.6657628, -5.3315255, 2.6657628, -1.8890331, 0.89487434)]


def check_redundancy():
    """the relations between the columns of the 8 kHz table (see the module docstring); raises on a broken one."""
    nr, cb, wb, wh, pc = (np.array(a, dtype=np.float64) for a in (NR_OF_HZ_BANDS_8K, CENTRE_OF_BAND_BARK_8K,
                                                                  WIDTH_OF_BAND_BARK_8K, WIDTH_OF_BAND_HZ_8K,
                                                                  POW_DENS_CORRECTION_FACTOR_8K))
    assert len(nr) == 42 and nr.sum() == 128
    eb = np.concatenate([[0.0], np.cumsum(wb)])
    assert np.abs((eb[:-1] + eb[1:]) / 2 - cb).max() < 5e-6
    assert np.abs(wh / wb / nr / pc - 1).max() < 5e-6
    eh = np.concatenate([[0.0], np.cumsum(wh)])
    bins = np.arange(128) * 31.25
    assert [int(((bins >= eh[k]) & (bins < eh[k + 1])).sum()) for k in range(42)] == [int(v) for v in nr]
    assert abs(eh[-1] - 4000.0) < 0.5
    assert abs(SP_8K / 4 - SP_16K) < 1e-11
    assert len(ABS_THRESH_POWER_8K) == 42 and len(CENTRE_OF_BAND_HZ_8K) == 42
    return True


def _tables_16k(edge_pos=None, bark_scale=1.0):
    """49-band table: bands 0..40 of the 8 kHz table, bands 41..48 reconstructed (module docstring).
    edge_pos / bark_scale parametrise the RECONSTRUCTED part for the sensitivity sweep (scripts/pesq_band_sweep.py): edge_pos
    in (0, 1) puts every free Hz edge at that position inside the one-bin interval the bin counts admit (None = the default
    Bark->Hz continuation); bark_scale scales the reconstructed Bark widths of bands 42-48."""
    wb8, wh8 = np.array(WIDTH_OF_BAND_BARK_8K), np.array(WIDTH_OF_BAND_HZ_8K)
    nr = NR_OF_HZ_BANDS_8K[:41] + NR_OF_HZ_BANDS_16K_TAIL
    assert sum(nr) == 256
    k = np.arange(20, 42)
    wfit = np.polyfit(k, wb8[20:42], 3)
    wb = np.concatenate([wb8[:42], bark_scale * np.polyval(wfit, np.arange(42, 49))])
    eb8 = np.concatenate([[0.0], np.cumsum(wb8)])[:42]          # lower Bark edges of bands 0..41
    eh8 = np.concatenate([[0.0], np.cumsum(wh8)])[:42]          # lower Hz edges of bands 0..41
    # Bark -> Hz of the band edges above 2 kHz: log f is close to quadratic in z there; continue it
    sel = eh8 > 2000.0
    ffit = np.polyfit(eb8[sel], np.log(eh8[sel]), 2)
    f_of_z = lambda z: float(np.exp(np.polyval(ffit, z)))
    edges_hz, edges_bark = [eh8[41]], [eb8[41]]
    first_bin = sum(nr[:41])
    for band in range(41, 49):
        zb = edges_bark[-1] + wb[band]
        first_bin += nr[band]
        if band == 48:
            hz = 8000.0
            # last band: cut at 8 kHz as the 8 kHz table's last band is cut at 4 kHz; its Bark width from the Hz-per-Bark
            # slope of the two bands below, continued geometrically
            s1 = (edges_hz[-1] - edges_hz[-2]) / wb[band - 1]
            s0 = (edges_hz[-2] - edges_hz[-3]) / wb[band - 2]
            zb = edges_bark[-1] + (hz - edges_hz[-1]) / (s1 * s1 / s0)
            wb[band] = zb - edges_bark[-1]
        else:
            lo, hi = (first_bin - 1) * 31.25, first_bin * 31.25      # the edge lies between the last bin in and the first out
            hz = f_of_z(zb)
            if not lo < hz < hi:
                hz = 0.5 * (lo + hi)
            if edge_pos is not None:
                hz = lo + edge_pos * (hi - lo)
        edges_hz.append(hz)
        edges_bark.append(zb)
    wh = np.concatenate([wh8[:41], np.diff(edges_hz)])
    ebark = np.concatenate([[0.0], np.cumsum(wb)])
    cb = (ebark[:-1] + ebark[1:]) / 2
    pc = np.concatenate([np.array(POW_DENS_CORRECTION_FACTOR_8K[:41]), wh[41:] / wb[41:] / np.array(nr[41:])])
    at = np.array(ABS_THRESH_POWER_8K[:41] + ABS_THRESH_POWER_16K_TAIL)
    cb[:41] = CENTRE_OF_BAND_BARK_8K[:41]
    return dict(nb=49, nr=np.array(nr), centre_bark=cb, width_bark=wb, width_hz=wh, pow_corr=pc, abs_thresh=at,
                sp=SP_16K, sl=SL_16K)


def tables(fs, **variant):
    check_redundancy()
    if variant:
        assert fs == 16000
        return _tables_16k(**variant)
    if fs == 8000:
        return dict(nb=42, nr=np.array(NR_OF_HZ_BANDS_8K), centre_bark=np.array(CENTRE_OF_BAND_BARK_8K),
                    width_bark=np.array(WIDTH_OF_BAND_BARK_8K), width_hz=np.array(WIDTH_OF_BAND_HZ_8K),
                    pow_corr=np.array(POW_DENS_CORRECTION_FACTOR_8K), abs_thresh=np.array(ABS_THRESH_POWER_8K), sp=SP_8K,
                    sl=SL_8K)
    if fs == 16000:
        return _tables_16k()
    raise ValueError(fs)
