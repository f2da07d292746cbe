"""Batch contract, data-parallel shard rule, dynamic-mixing producer and audio I/O of the training loop.

Same surface as ``baseline_code/dataset.py`` (``read_kv_scp`` :79-86, ``read_source_scp`` :89-101,
``PreSimulatedDataset`` :104-151, ``DynamicMixingDataset`` :154-335, ``GroupedBatchSampler`` :338-401, ``collate_fn``
:404-441, ``AudioDataModule`` :444-524) with one structural difference: ``DynamicMixingDataset`` does not run the
simulator on the host.  A worker only DRAWS the recipe (same ``np.random`` call sequence as ``run_simulation`` :232-278
+ ``generate_data_param.process_one_sample`` :294-418, so the same seed gives the same recipe) and reads the raw
sources; ``collate_dynamic`` stacks them, and the DSP (high-pass, RIR, SNR mixing, clipping, packet loss, peak
normalisation) runs batched on the GPU in ``mixing.simulate_batch`` when the trainer calls ``materialise`` on the batch.

What must agree with the reference for identical shards is the order of RNG calls, not the text: it is pinned by
``tests/golden/ref_mix.npz`` (batches and recipes produced by the reference's own classes, ``make_golden_mix.py``).
``soundfile`` is not available here: WAV and FLAC are decoded by ``audio_io`` (own RIFF / FLAC readers).
"""
import random
from collections import defaultdict

import math

import numpy as np
import torch
from torch.utils.data import BatchSampler, DataLoader

from .audio_io import audio_frames, read_audio, write_audio  # noqa: F401  (re-exported: inference / metrics use them)


# ------------------------------------------------------------------------------------------------------------------
# scp tables
# ------------------------------------------------------------------------------------------------------------------
def _scp_rows(path, n_fields):
    with open(path, "r") as f:
        for line in f:
            parts = line.strip().split()
            if len(parts) != n_fields:
                raise ValueError("%s: expected %d fields, got %r" % (path, n_fields, line))
            yield parts


def read_kv_scp(scp):
    """``uid value`` lines -> dict (duplicate uids are an error)."""
    table = {}
    for uid, value in _scp_rows(scp, 2):
        if uid in table:
            raise AssertionError(uid)
        table[uid] = value
    return table


def read_source_scp(scp):
    """``uid fs path`` lines -> ({fs: {uid: path}}, {fs: [uid]}, {uid: path})."""
    by_fs, flat = defaultdict(dict), {}
    for uid, fs, path in _scp_rows(scp, 3):
        if uid in by_fs[int(fs)]:
            raise AssertionError((uid, fs))
        by_fs[int(fs)][uid] = path
        flat[uid] = path
    return by_fs, {fs: list(d) for fs, d in by_fs.items()}, flat


# ------------------------------------------------------------------------------------------------------------------
# datasets
# ------------------------------------------------------------------------------------------------------------------
class PreSimulatedDataset(torch.utils.data.Dataset):
    """(clean [1,T], noisy [1,T], fs, T) from the four tables of a simulated set; ``max_duration`` (in SAMPLES, quirk
    C.4) crops both signals at one ``random.randint`` offset."""

    def __init__(self, clean_speech, noisy_speech, utt2fs, speech_length, max_duration=-1):
        self.clean_speech, self.noisy_speech = read_kv_scp(clean_speech), read_kv_scp(noisy_speech)
        self.utt2fs = {u: int(v) for u, v in read_kv_scp(utt2fs).items()}
        self.speech_length = {u: int(v) for u, v in read_kv_scp(speech_length).items()}
        self.uid = list(self.clean_speech)
        self.max_duration = max_duration
        sizes = {len(self.clean_speech), len(self.noisy_speech), len(self.utt2fs), len(self.speech_length)}
        assert len(sizes) == 1, "the four tables must list the same utterances"

    def __len__(self):
        return len(self.uid)

    def get_srs(self):
        return [self.utt2fs[u] for u in self.uid]

    def get_source_length(self):
        cap = self.max_duration if self.max_duration > 0 else None
        return [self.speech_length[u] if cap is None else min(self.speech_length[u], cap) for u in self.uid]

    def __getitem__(self, index):
        uid = self.uid[index]
        pair = []
        for table in (self.clean_speech, self.noisy_speech):
            wav, fs = read_audio(table[uid])
            assert fs == self.utt2fs[uid], (uid, fs)
            pair.append(wav)
        clean, noisy = pair
        T = clean.shape[1]
        if 0 < self.max_duration < T:
            start = random.randint(0, T - self.max_duration)
            clean, noisy = (w[:, start:start + self.max_duration] for w in (clean, noisy))
        return clean, noisy, fs, clean.shape[1]


class SyntheticPairDataset(torch.utils.data.Dataset):
    """low-passed noise x 4 Hz envelope with 0.4 s near-silent edges (clean), + white noise at
    U(-5, 20) dB SNR (noisy), both peak-normalised to 0.9.  Deterministic per (seed, index)."""

    def __init__(self, n_items, fs_list=(48000,), seconds=4.0, seed=2024, vary_length=False):
        self.n, self.fs_list, self.seconds, self.seed, self.vary = n_items, list(fs_list), seconds, seed, vary_length

    def _len(self, i):
        fs = self.fs_list[i % len(self.fs_list)]
        L = int(self.seconds * fs)
        if self.vary:
            L = int(L * (0.6 + 0.4 * ((i * 2654435761) % 1000) / 1000.0))
        return fs, L

    def get_source_length(self):
        return [self._len(i)[1] for i in range(self.n)]

    def get_srs(self):
        return [self._len(i)[0] for i in range(self.n)]

    def __len__(self):
        return self.n

    @staticmethod
    def speech_like(rng, L, fs):
        n = rng.standard_normal(L)
        k = np.fft.rfftfreq(L)
        clean = np.fft.irfft(np.fft.rfft(n) / (1.0 - 0.95 * np.exp(-2j * np.pi * k)), n=L)
        clean /= clean.std()
        t = np.arange(L) / fs
        clean *= 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi))
        edge = min(int(0.4 * fs), L // 4)
        clean[:edge] *= 1e-3
        clean[L - edge:] *= 1e-3
        return clean * (0.9 / np.abs(clean).max())

    def __getitem__(self, i):
        fs, L = self._len(i)
        rng = np.random.default_rng(self.seed * 1000003 + i)
        clean = self.speech_like(rng, L, fs)
        snr = rng.uniform(-5.0, 20.0)
        noise = rng.standard_normal(L)
        noise *= np.sqrt((clean ** 2).mean() / ((noise ** 2).mean() * 10 ** (snr / 10)))
        noisy = clean + noise
        sc = 0.9 / max(np.abs(noisy).max(), np.abs(clean).max())
        return (clean * sc).astype(np.float32)[None], (noisy * sc).astype(np.float32)[None], fs, L


# ------------------------------------------------------------------------------------------------------------------
# dynamic mixing: recipe on the host, DSP on the device
# ------------------------------------------------------------------------------------------------------------------
class SimulationConfigs:
    """the probabilities / ranges of ``baseline_code/dataset.py:20-76`` (values are the contract; layout is ours)."""
    snr_low_bound, snr_high_bound = -5.0, 20.0
    reuse_noise, reuse_rir = True, True
    prob_wind_noise = 0.05
    prob_reverberation = 0.5
    wind_noise_config = dict(threshold=[0.1, 0.3], ratio=[1, 20], attack=[5, 100], release=[5, 100], sc_gain=[0.8, 1.2],
                             clipping_threshold=[0.85, 1.0], clipping_chance=0.75, wind_noise_snr_low_bound=-10.0,
                             wind_noise_snr_high_bound=15.0)
    num_augmentations = {0: 0.25, 1: 0.40, 2: 0.20, 3: 0.15}
    augmentations = dict(
        bandwidth_limitation=dict(weight=1.0, resample_methods="random"),
        clipping=dict(weight=1.0, clipping_min_quantile=[0.0, 0.1], clipping_max_quantile=[0.9, 1.0]),
        codec=dict(weight=1.0, config=[dict(format="mp3", encoder=None, qscale=[1, 10]),
                                       dict(format="ogg", encoder=["vorbis"], qscale=[-1, 10])]),
        packet_loss=dict(weight=1.0, packet_duration_ms=20, max_continuous_packet_loss=10, packet_loss_rate=[0.05, 0.25]),
    )
    augmentations_name = list(augmentations)


BANDWIDTH_RATES = (8000, 16000, 22050, 24000, 32000, 44100, 48000)            # generate_data_param.py:14
BANDWIDTH_METHODS = ("kaiser_best", "kaiser_fast", "scipy", "polyphase")       # generate_data_param.py:16-26


def _pick_source(fs, table, rs):
    """``select_sample`` (generate_data_param.py:421-455) for the on-the-fly case (nothing is marked used): a sample at
    ``fs`` if there is one, else one at a higher rate found by walking a shuffled list of the rates."""
    if table.get(fs):
        return rs.choice(list(table[fs]))
    rates = list(table)
    rs.shuffle(rates)
    for other in rates:
        if other > fs and table[other]:
            return rs.choice(list(table[other]))
    return None


def _lost_packets(n_samples, fs, cfg, rs):
    """``packet_loss`` index draw (generate_data_param.py:58-91)."""
    dur_ms = n_samples / fs * 1000
    n_packets = int(dur_ms // cfg["packet_duration_ms"])
    rate = rs.uniform(*cfg["packet_loss_rate"])
    n_lost = int(round(rate * dur_ms / cfg["packet_duration_ms"], 0))
    cap = cfg["max_continuous_packet_loss"]
    runs = []
    for _ in range(n_lost):
        runs.append(rs.randint(1, cap))
        if n_lost - sum(runs) <= cap:
            runs.append(n_lost - sum(runs))
            break
    starts = rs.choice(range(n_packets), len(runs), replace=False)
    lost = []
    for s, n in zip(starts, runs):
        lost += list(range(s, s + n))
    return list(set(lost))       # (built exactly like this in the reference: the order of the list is the set's)


def draw_recipe(speech_length, fs, noise_table, rir_table, wind_table, cfg=SimulationConfigs, rs=np.random):
    """One mixing recipe, drawn with the reference's sequence of ``np.random`` calls: wind-noise coin, number and choice
    of augmentations (re-drawn while wind noise meets clipping), noise sample, [wind-noise parameters,] SNR, RIR coin and
    sample, per-augmentation parameters.  Returns the reference's ``info`` fields plus the parsed parameters."""
    names = list(cfg.augmentations)
    w = np.array([cfg.augmentations[a]["weight"] for a in names], dtype=float)
    w = w / w.sum()
    wind = rs.random() < cfg.prob_wind_noise
    n_aug = rs.choice(list(cfg.num_augmentations), p=list(cfg.num_augmentations.values()))
    chosen = []
    if n_aug:
        chosen = rs.choice(names, p=w, size=n_aug, replace=False)
        while wind and "clipping" in chosen:
            chosen = rs.choice(names, p=w, size=n_aug, replace=False)
    text, params = "", {}
    if wind:
        noise_uid = _pick_source(fs, wind_table, rs)
        wc = cfg.wind_noise_config
        vals = [rs.uniform(*wc[k]) for k in ("threshold", "ratio", "attack", "release", "sc_gain", "clipping_threshold")]
        clip = rs.random() < wc["clipping_chance"]
        text = ("wind_noise(threshold=%s,ratio=%s,attack=%s,release=%s,sc_gain=%s,clipping=%s,clipping_threshold=%s)/"
                % (vals[0], vals[1], vals[2], vals[3], vals[4], clip, vals[5]))
        params["wind_noise"] = dict(zip(("threshold", "ratio", "attack", "release", "sc_gain", "clipping_threshold"), vals),
                                    clipping=clip)
        snr = rs.uniform(wc["wind_noise_snr_low_bound"], wc["wind_noise_snr_high_bound"])
    else:
        noise_uid = _pick_source(fs, noise_table, rs)
        snr = rs.uniform(cfg.snr_low_bound, cfg.snr_high_bound)
    if noise_uid is None:
        raise ValueError("Noise sample not found for fs=%d+ Hz" % fs)
    # (the reference keeps the RIR when the draw EXCEEDS prob_reverberation, generate_data_param.py:337-345)
    if rir_table is None or cfg.prob_reverberation <= 0.0 or rs.rand() <= cfg.prob_reverberation:
        rir_uid = None
    else:
        rir_uid = _pick_source(fs, rir_table, rs)
    if len(chosen) == 0:
        if not wind:
            text = "none"
    for i, a in enumerate(chosen):
        spec = cfg.augmentations[a]
        if a == "bandwidth_limitation":
            lower = [r for r in BANDWIDTH_RATES if r < fs]
            if lower:
                method = rs.choice(BANDWIDTH_METHODS)
                fs_new = rs.choice(lower)
            else:
                method, fs_new = "none", fs
            text += "%s-%s->%s" % (a, method, fs_new)
            params[a] = dict(res_type=str(method), fs_new=int(fs_new))
        elif a == "clipping":
            lo, hi = rs.uniform(*spec["clipping_min_quantile"]), rs.uniform(*spec["clipping_max_quantile"])
            text += "%s(min=%s,max=%s)" % (a, lo, hi)
            params[a] = dict(min_quantile=float(lo), max_quantile=float(hi))
        elif a == "codec":
            c = rs.choice(spec["config"], 1)[0]
            enc, q = c["encoder"], c["qscale"]
            if isinstance(enc, list):
                enc = rs.choice(enc, 1)[0]
            if isinstance(q, list):
                q = rs.randint(*q)
            text += "%s(format=%s,encoder=%s,qscale=%s)" % (a, c["format"], enc, q)
            params[a] = dict(format=c["format"], encoder=enc, qscale=q)
        elif a == "packet_loss":
            idx = _lost_packets(speech_length, fs, spec, rs)
            text += "%s(packet_loss_indices=%s,packet_duration_ms=%s)" % (a, idx, spec["packet_duration_ms"])
            params[a] = dict(packet_loss_indices=[int(v) for v in idx], packet_duration_ms=spec["packet_duration_ms"])
        else:
            raise NotImplementedError(a)
        if i < len(chosen) - 1:
            text += "/"
    return dict(noise_uid=noise_uid, rir_uid="none" if rir_uid is None else rir_uid, snr=snr, augmentation=text, fs=fs,
                length=speech_length, params=params, order=[str(a) for a in chosen], wind=bool(wind))


class DynamicMixingDataset(torch.utils.data.Dataset):
    """Index -> (raw speech, raw noise, raw RIR or None, recipe, fs, length).  Constructor arguments as the reference's
    (:155); ``reader`` / ``frames`` replace the soundfile calls (tests and bench.py serve in-memory sources)."""

    def __init__(self, speech_source_scp, noise_source_scp, rir_scp, windnoise_scp, speech_length_file,
                 use_high_pass=True, retry_when_fails=False, max_duration=240000, reader=read_audio, frames=audio_frames):
        super().__init__()
        self.speech_source, self.speech_uids, self.speech_source_flt = read_source_scp(speech_source_scp)
        self.noise_source, self.noise_uids, self.noise_source_flt = read_source_scp(noise_source_scp)
        self.rirs, self.rir_uids, self.rirs_flt = read_source_scp(rir_scp)
        self.wind_noises, self.wind_noises_uids, self.wind_noises_flt = read_source_scp(windnoise_scp)
        self.all_noise_flt = dict(self.noise_source_flt, **self.wind_noises_flt)
        self.max_duration = max_duration
        self.source_length = {u: min(int(v), max_duration) for u, v in read_kv_scp(speech_length_file).items()}
        self.samplerates = list(self.speech_source)
        self._index = [(fs, j) for fs in self.samplerates for j in range(len(self.speech_source[fs]))]
        self.use_high_pass, self.retry_when_fails = use_high_pass, retry_when_fails
        self._read, self._frames = reader, frames
        self.skipped = defaultdict(int)          # augmentations drawn but not applicable on the device path

    def __len__(self):
        return len(self._index)

    def _get_from_index(self, index):
        return self._index[index]

    def get_srs(self):
        return [fs for fs, _ in self._index]

    def get_source_length(self):
        return [self.source_length[self.speech_uids[fs][j]] for fs, j in self._index]

    def _crop(self, wav):
        """the random crop of the simulator's ``read_audio(max_duration)`` (simulate_data_from_param.py:355-359)."""
        if 0 < self.max_duration < wav.shape[1]:
            start = random.randint(0, wav.shape[1] - self.max_duration)
            wav = wav[:, start:start + self.max_duration]
        return wav

    def _load(self, path, fs):
        """-> (mono wav [1, n], sampling rate of the file).  A file at the recipe's rate gets the simulator's random crop; a file at
        ANOTHER rate (`_pick_source` falls back to noise / wind / RIR files of a higher rate by design, as the reference's
        select_sample does) is handed on raw with its rate: the reference resamples it to fs with soxr_hq and returns BEFORE the
        crop (simulate_data_from_param.py:350-352); here that resampling runs on the device in ``RawMixBatch.materialise``."""
        wav, got = self._read(path)
        if got != fs:
            return wav[:1], int(got)
        return self._crop(wav[:1]), int(fs)

    @staticmethod
    def resampled_length(n, src_fs, fs):
        """length of n samples at src_fs after resampling to fs (librosa.resample: ceil(n * fs / src_fs))."""
        return n if src_fs == fs else -((-n * fs) // src_fs)

    @staticmethod
    def crop_for_resampling(x, src_fs, fs, offset, length, margin=4096):
        """A noise file at a higher rate is resampled as a whole by the reference and then cropped to [offset, offset + length)
        (read_audio + mix_noise); minutes of 48 kHz noise for a 4 s utterance need not travel to the device for that.  The polyphase
        resampler's output sample n depends on the input only through n * down - j * up, so cutting the SOURCE at a multiple of
        `down` (s0 = q * down <-> output index q * up) and `margin` source samples around the window (the soxr-HQ-specification filter
        reaches < 300 source samples) gives bit-identical samples inside the window.  -> (cropped source, offset inside its resampling)."""
        g = math.gcd(int(fs), int(src_fs))
        up, down = int(fs) // g, int(src_fs) // g
        n_src = x.shape[1]
        q = max(0, (offset * down // up - margin) // down)                 # source start s0 = q * down, output start o0 = q * up
        s0, o0 = q * down, q * up
        s1 = min(n_src, -((-(offset + length) * down) // up) + margin)
        if s0 == 0 and s1 == n_src:
            return x, offset
        return x[:, s0:s1], offset - o0

    def __getitem__(self, index):
        fs, j = self._index[index]
        uid = self.speech_uids[fs][j]
        path = self.speech_source[fs][uid]
        length = min(self.max_duration, self._frames(path))
        recipe = draw_recipe(length, fs, self.noise_source, self.rirs, self.wind_noises)
        recipe.update(speech_uid=uid, id=uid, snr_dB=recipe["snr"], highpass=self.use_high_pass)
        speech, speech_fs = self._load(path, fs)
        if speech_fs != fs:
            raise ValueError("%s is sampled at %d Hz but listed at %d Hz in the speech scp" % (path, speech_fs, fs))
        noise, noise_fs = self._load(self.all_noise_flt[recipe["noise_uid"]], fs)
        rir, rir_fs = self._load(self.rirs_flt[recipe["rir_uid"]], fs) if recipe["rir_uid"] != "none" else (None, fs)
        # the offset of mix_noise's wrap / crop (:108-119) comes from an unseeded default_rng() when mixing on the fly (:471);
        # it is drawn on the noise length AFTER the resampling to fs
        ls, ln = speech.shape[1], self.resampled_length(noise.shape[1], noise_fs, fs)
        recipe["noise_offset"] = int(np.random.default_rng().integers(0, abs(ls - ln))) if ls != ln else 0
        if noise_fs != fs and ln > ls:
            noise, recipe["noise_offset"] = self.crop_for_resampling(noise, noise_fs, fs, recipe["noise_offset"], ls)
        return dict(speech=speech, noise=noise, rir=rir, recipe=recipe, fs=fs, length=speech.shape[1], noise_fs=noise_fs,
                    rir_fs=rir_fs)


class RawMixBatch:
    """What ``collate_dynamic`` hands to the trainer: padded raw sources + recipes of one fs.  ``materialise(device)``
    runs the simulator on the GPU and returns the ``collate_fn`` tuple ``(clean[B,1,T], noisy[B,1,T], fs, lengths)``."""

    def __init__(self, items, pinned=False):
        """pinned: stack straight into page-locked memory (an in-process producer; DataLoader workers cannot, their batches
        cross a process boundary and the loader's pin thread calls ``pin_memory()`` instead)."""
        assert len({it["fs"] for it in items}) == 1, "mixed sampling rates in one batch"
        self.fs = items[0]["fs"]
        self.recipes = [it["recipe"] for it in items]
        self.lengths = [it["length"] for it in items]

        def stack(key, width):
            out = torch.zeros(len(items), width, dtype=torch.float32, pin_memory=pinned)
            lens = []
            for b, it in enumerate(items):
                a = it[key]
                n = 0 if a is None else a.shape[1]
                if n:
                    out[b, :n] = torch.as_tensor(a[0], dtype=torch.float32)
                lens.append(n)
            return out, lens
        self.speech, _ = stack("speech", max(self.lengths))
        self.noise, self.noise_lens = stack("noise", max(it["noise"].shape[1] for it in items))
        rmax = max([it["rir"].shape[1] for it in items if it["rir"] is not None] or [0])
        self.rir, self.rir_lens = stack("rir", rmax) if rmax else (None, [0] * len(items))
        self.noise_fs = [it.get("noise_fs", self.fs) for it in items]
        self.rir_fs = [it.get("rir_fs", self.fs) for it in items]
        self.rir_early = [0] * len(items)
        if rmax:
            from .mixing import early_rir_stop
            # (a RIR at another rate: its direct-path / early-part boundary is found after the resampling, on the device copy)
            self.rir_early = [early_rir_stop(it["rir"], self.fs) if it["rir"] is not None and self.rir_fs[b] == self.fs else 0
                              for b, it in enumerate(items)]

    def pin_memory(self):
        for k in ("speech", "noise", "rir"):
            t = getattr(self, k)
            if t is not None and not t.is_pinned():
                setattr(self, k, t.pin_memory())
        return self

    def _to_batch_rate(self, x, lens, src_fs):
        """rows of x [B, W] recorded at another rate than the batch's -> resampled to it on the device (soxr-HQ-specification
        polyphase filter, metrics.resample_soxr_hq), one launch per distinct source rate; -> (x', lens')."""
        if x is None or all(f == self.fs for f in src_fs):
            return x, lens
        from .metrics import resample_soxr_hq
        new_lens = [DynamicMixingDataset.resampled_length(n, f, self.fs) for n, f in zip(lens, src_fs)]
        out = torch.zeros(x.shape[0], max(max(new_lens), 1), dtype=x.dtype, device=x.device)
        for f in sorted(set(src_fs)):
            rows = [b for b, g in enumerate(src_fs) if g == f and lens[b] > 0]
            if not rows:
                continue
            if f == self.fs:
                w = max(lens[b] for b in rows)
                out[rows, :w] = x[rows, :w]
                continue
            w = max(lens[b] for b in rows)
            y = resample_soxr_hq(x[rows, :w], f, self.fs)           # zero padding behind a row resamples to zeros
            for i, b in enumerate(rows):
                out[b, :new_lens[b]] = y[i, :new_lens[b]]
        return out, new_lens

    def materialise(self, device, skipped=None):
        from . import mixing
        noise, noise_lens = self._to_batch_rate(self.noise.to(device, non_blocking=True), self.noise_lens, self.noise_fs)
        rir = None if self.rir is None else self.rir.to(device, non_blocking=True)
        rir, rir_lens = self._to_batch_rate(rir, self.rir_lens, self.rir_fs)
        rir_early = list(self.rir_early)
        for b, f in enumerate(self.rir_fs):
            if rir is not None and f != self.fs and rir_lens[b] > 0:
                rir_early[b] = mixing.early_rir_stop(rir[b:b + 1, :rir_lens[b]].cpu().numpy(), self.fs)
        out = mixing.simulate_recipes(self.speech.to(device, non_blocking=True), self.lengths, noise, noise_lens,
                                      rir, rir_lens, rir_early, self.fs, self.recipes, skipped)
        clean, noisy = out
        B, T = clean.shape
        return (clean.view(B, 1, T), noisy.view(B, 1, T), torch.tensor(self.fs, dtype=torch.int32),
                torch.tensor(self.lengths, dtype=torch.int32))


def collate_dynamic(items, pinned=False):
    return RawMixBatch(items, pinned)


# ------------------------------------------------------------------------------------------------------------------
# shard rule + batching
# ------------------------------------------------------------------------------------------------------------------
class GroupedBatchSampler(BatchSampler):
    """Per sampling rate: indices sorted by length, every ``world_size``-th one from ``rank`` on (the data-parallel
    shard), cut into buckets of ``bucket_size_mult * batch_size``.  An epoch shuffles the bucket list, each bucket and
    finally the batch list with the ``random`` module seeded by ``epoch + rank``; the shuffles are in place, so they
    accumulate over epochs exactly as in the reference (quirk C.3).

    ``max_batches``: ranks see different batch counts when the per-fs groups do not divide evenly (the reference then
    hangs in DDP); the trainer sets it to the minimum over ranks so that every rank runs the same number of steps."""

    def __init__(self, dataset, batch_size, rank, world_size, seed=0, drop_last=False, bucket_size_mult=100,
                 sampler=None):
        self.batch_size, self.drop_last = batch_size, drop_last
        self.bucket_size = batch_size * bucket_size_mult
        self.rank, self.world_size, self.seed, self.epoch = rank, world_size, seed, 0
        self.generator = torch.Generator().manual_seed(seed + rank)
        self.max_batches = None
        lengths = dataset.get_source_length()
        by_rate = defaultdict(list)
        for i, fs in enumerate(dataset.get_srs()):
            by_rate[fs].append(i)
        self.buckets = []
        for members in by_rate.values():
            shard = sorted(members, key=lengths.__getitem__)[rank::world_size]
            self.buckets += [shard[k:k + self.bucket_size] for k in range(0, len(shard), self.bucket_size)]

    def set_epoch(self, epoch):
        self.epoch = epoch
        self.generator.manual_seed(self.seed + self.rank + epoch)

    def _batches_of(self, bucket):
        full, rest = divmod(len(bucket), self.batch_size)
        n = full + (1 if rest and not self.drop_last else 0)
        return [bucket[k * self.batch_size:(k + 1) * self.batch_size] for k in range(n)]

    def __iter__(self):
        random.seed(self.epoch + self.rank)
        random.shuffle(self.buckets)
        plan = []
        for bucket in self.buckets:
            random.shuffle(bucket)
            plan += self._batches_of(bucket)
        random.shuffle(plan)
        return iter(plan if self.max_batches is None else plan[:self.max_batches])

    def __len__(self):
        n = sum(len(self._batches_of(b)) for b in self.buckets)
        return n if self.max_batches is None else min(n, self.max_batches)

    def state_dict(self):
        return {"seed": self.seed, "epoch": self.epoch}


def collate_fn(batch):
    """[(clean [1,Ti], noisy [1,Ti], fs, Ti)] -> (clean [B,1,Tmax], noisy [B,1,Tmax], fs int32 0-d, lengths int32 [B]),
    right-padded with zeros; the sample dtype is kept (the reference's ``torch.tensor(item)``)."""
    rates = {item[2] for item in batch}
    assert len(rates) == 1, "mixed sampling rates in one batch"
    width = max(np.shape(item[0])[1] for item in batch)
    out = []
    for col in (0, 1):
        first = torch.as_tensor(batch[0][col])
        padded = first.new_zeros((len(batch), first.shape[0], width))
        for b, item in enumerate(batch):
            a = torch.as_tensor(item[col])
            padded[b, :, :a.shape[1]] = a
        out.append(padded)
    return (out[0], out[1], torch.tensor(rates.pop(), dtype=torch.int32),
            torch.tensor([item[3] for item in batch], dtype=torch.int32))


class AudioDataModule:
    """``AudioDataModule(config)`` of the reference (:444-524) without Lightning; rank / world come from the trainer."""

    def __init__(self, config, rank=0, world_size=1):
        self.config, self.rank, self.world_size = config, rank, world_size
        self.num_worker, self.batch_size = config.num_worker, config.batch_size
        td, vd = config.train_set_path, config.valid_set_path
        self.dynamic = False
        if str(td).startswith("synthetic"):            # "synthetic[:n_items]" -> on-the-fly SURVEY 8(d) pairs
            n = int(str(td).split(":")[1]) if ":" in str(td) else 64
            fs_list = tuple(getattr(config, "synthetic_fs", (48000,)))
            self.train_dataset = SyntheticPairDataset(n, fs_list=fs_list, seconds=getattr(config, "synthetic_seconds", 4.0),
                                                      seed=config.seed)
            self.val_dataset = SyntheticPairDataset(max(self.batch_size, n // 8), fs_list=fs_list[:1],
                                                    seconds=getattr(config, "synthetic_seconds", 4.0), seed=config.seed + 1)
        else:
            if config.train_set_dynamic_mixing:
                self.dynamic = True
                self.train_dataset = DynamicMixingDataset(
                    speech_source_scp="%s/speech_sources.scp" % td, noise_source_scp="%s/noise_scoures.scp" % td,
                    speech_length_file="%s/source_length.scp" % td, rir_scp="%s/rirs.scp" % td,
                    windnoise_scp="%s/wind_noise_scoures.scp" % td, retry_when_fails=False,
                    max_duration=config.max_duration, use_high_pass=config.use_high_pass)
            else:
                self.train_dataset = self._presimulated(td, config.max_duration)
            self.val_dataset = self._presimulated(vd, -1)
        self.train_batch_sampler = None

    @staticmethod
    def _presimulated(d, max_duration):
        return PreSimulatedDataset("%s/spk1.scp" % d, "%s/wav.scp" % d, "%s/utt2fs" % d, "%s/speech_length.scp" % d,
                                   max_duration=max_duration)

    def _loader(self, ds, sampler, collate):
        nw = self.num_worker
        return DataLoader(ds, batch_sampler=sampler, num_workers=nw, pin_memory=False, persistent_workers=nw > 0,
                          collate_fn=collate)

    def train_dataloader(self):
        self.train_batch_sampler = GroupedBatchSampler(self.train_dataset, self.batch_size, self.rank, self.world_size,
                                                       drop_last=True)
        return self._loader(self.train_dataset, self.train_batch_sampler, collate_dynamic if self.dynamic else collate_fn)

    def val_dataloader(self):       # not sharded: every rank validates the full set (dataset.py:507-516)
        return self._loader(self.val_dataset, GroupedBatchSampler(self.val_dataset, self.batch_size, 0, 1, drop_last=True),
                            collate_fn)
