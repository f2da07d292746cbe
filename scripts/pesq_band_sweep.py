"""How much can the RECONSTRUCTED part of the 16 kHz Bark table (bands 41-48: Hz edges free inside one 31.25 Hz bin each, Bark
widths a smooth continuation) move a wide-band PESQ score?  Scores a set of seeded wide-band pairs (noise 0-30 dB SNR, one
low-passed, one with high-frequency noise only - the case that leans hardest on the bands above 4 kHz) with the oracle at the
default table and at the extremes the bin counts admit; prints and stores the spread (profiles/r03_pesq_band_sweep.json), which
bench.py quotes in metrics_bench.pesq_note.  CPU only (oracle)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pesq_ref  # noqa: E402
from tests import pesq_cases  # noqa: E402

VARIANTS = {"default": {}, "edges_low": dict(edge_pos=0.02), "edges_high": dict(edge_pos=0.98), "edges_mid": dict(edge_pos=0.5),
            "bark_-3%": dict(bark_scale=0.97), "bark_+3%": dict(bark_scale=1.03),
            "edges_low_bark_+3%": dict(edge_pos=0.02, bark_scale=1.03), "edges_high_bark_-3%": dict(edge_pos=0.98, bark_scale=0.97)}


def pairs():
    fs, out = 16000, []
    for i, snr in enumerate((0.0, 5.0, 10.0, 15.0, 20.0, 30.0)):
        rng = np.random.default_rng(900 + i)
        c = pesq_cases.speech_like(rng, 4 * fs, fs)
        out.append(("white noise %g dB" % snr, c, c + rng.standard_normal(len(c)) * np.sqrt((c ** 2).mean() / 10 ** (snr / 10))))
    rng = np.random.default_rng(990)
    c = pesq_cases.speech_like(rng, 4 * fs, fs)
    X = np.fft.rfft(c)
    X[int(len(X) * 4500 / 8000):] = 0                    # enhanced signal lost everything above 4.5 kHz
    out.append(("low-passed at 4.5 kHz", c, np.fft.irfft(X, len(c))))
    n = rng.standard_normal(len(c))
    N = np.fft.rfft(n)
    N[:int(len(N) * 4200 / 8000)] = 0                    # noise only above 4.2 kHz
    n = np.fft.irfft(N, len(c))
    out.append(("noise above 4.2 kHz, 10 dB", c, c + n * np.sqrt((c ** 2).mean() / (n ** 2).mean() / 10.0)))
    return fs, out


def main():
    fs, ps = pairs()
    res = {}
    for name, var in VARIANTS.items():
        pesq_ref.TABLE_VARIANT = var
        res[name] = [pesq_ref.pesq(fs, c, d, "wb") for _, c, d in ps]
    pesq_ref.TABLE_VARIANT = {}
    base = np.array(res["default"])
    spread = {k: float(np.abs(np.array(v) - base).max()) for k, v in res.items() if k != "default"}
    out = {"pairs": [p[0] for p in ps], "mos_default": base.tolist(), "max_abs_delta_mos_per_variant": spread,
           "max_abs_delta_mos": max(spread.values()),
           "per_pair_max_delta": np.abs(np.array([v for k, v in res.items() if k != "default"]) - base).max(0).tolist()}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r03_pesq_band_sweep.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
