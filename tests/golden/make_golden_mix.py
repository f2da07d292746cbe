"""Golden vectors produced BY THE REFERENCE'S OWN CODE for the data path (run in the build container only, where
/root/reference exists): the on-the-fly simulator (``simulation/simulate_data_from_param.py``), the recipe draw
(``baseline_code/dataset.py::DynamicMixingDataset.run_simulation`` -> ``simulation/generate_data_param.py``), the
data-parallel shard rule (``dataset.py::GroupedBatchSampler``) and ``collate_fn``.

The reference modules import packages this image lacks (soundfile, librosa, torchaudio, espnet2, pytorch_lightning).
They are registered as EMPTY stand-in modules so that the reference files import; the only behaviour supplied is
  * ``soundfile.read`` / ``soundfile.SoundFile``: serve seeded in-memory arrays by "path" (the reference's file reads),
  * ``espnet2.train.preprocessor.detect_non_silence``: oracle/mix_ref.py's restatement (espnet is absent: that one
    function stays unpinned, everything around it is the reference's own code).
Codec / bandwidth-limitation / wind-noise branches need ffmpeg / librosa and are never taken by the recipes stored here.
Only data is stored (inputs, recipes, outputs), never reference source.  The script also asserts that oracle/mix_ref.py
agrees with the reference's outputs, which is what pins that oracle.
"""
import importlib
import io
import os
import random
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import mix_ref  # noqa: E402

AUDIO = {}      # "path" -> (float64 [T] array, fs): what the stand-in soundfile serves


def install_stand_ins():
    import torch.nn as nn

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    def sf_read(path, always_2d=False, **kw):
        x, fs = AUDIO[path]
        x = np.array(x, dtype=np.float64)
        return (x[:, None] if always_2d else x), fs

    class SoundFile:
        def __init__(self, path):
            self.frames = len(AUDIO[path][0])

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def unavailable(*a, **k):
        raise RuntimeError("this branch needs a package the image lacks")
    mod("soundfile", read=sf_read, SoundFile=SoundFile, write=unavailable)
    mod("librosa", resample=unavailable)
    mod("torchaudio")
    mod("torchaudio.io", AudioEffector=unavailable, CodecConfig=unavailable)
    mod("espnet2"); mod("espnet2.train"); mod("espnet2.utils")
    mod("espnet2.train.preprocessor", detect_non_silence=mix_ref.detect_non_silence)
    import argparse
    mod("espnet2.utils.config_argparse", ArgumentParser=argparse.ArgumentParser)
    mod("espnet2.utils.types", str2bool=lambda s: str(s).lower() in ("1", "true", "yes"))
    mod("pytorch_lightning", LightningDataModule=type("LightningDataModule", (), {}))
    if REF not in sys.path:
        sys.path.insert(0, REF)


def synth(rng, n, fs, kind="speech"):
    x = rng.standard_normal(n)
    if kind == "speech":
        y = np.zeros(n)
        acc = 0.0
        for i in range(n):           # one-pole low-pass + syllabic envelope + silent edges (SURVEY 8d generator, small n)
            acc = 0.95 * acc + x[i]
            y[i] = acc
        t = np.arange(n) / fs
        y *= 0.55 + 0.45 * np.sin(2 * np.pi * 4 * t + rng.uniform(0, 6.28))
        e = n // 10
        y[:e] *= 1e-3
        y[n - e:] *= 1e-3
        return 0.5 * y / np.abs(y).max()
    if kind == "rir":
        h = rng.standard_normal(n) * np.exp(-np.arange(n) / (0.05 * fs))
        h[:int(0.002 * fs)] *= 0.02
        return h / np.abs(h).max()
    return 0.1 * x


def dsp_fixtures(sim, rir_utils):
    """per-function and whole-sample outputs of the reference simulator."""
    out = {}
    rng = np.random.default_rng(2026)
    fs = 16000
    sp = synth(rng, 6000, fs)[None]
    nz_short = synth(rng, 2500, fs, "noise")[None]
    nz_long = synth(rng, 9000, fs, "noise")[None]
    rir = synth(rng, 1400, fs, "rir")[None]
    out.update(sp=sp, nz_short=nz_short, nz_long=nz_long, rir=rir, fs=np.int64(fs))

    class FixedRng:                     # mix_noise draws its offset from rng.integers: make the draw an input
        def __init__(self, v):
            self.v = v

        def integers(self, lo, hi):
            assert lo <= self.v < hi
            return self.v
    for tag, nz, off in (("short", nz_short, 1234), ("long", nz_long, 2321)):
        noisy, noise = sim.mix_noise(sp.copy(), nz.copy(), snr=3.5, rng=FixedRng(off))
        out["mix_%s_noisy" % tag], out["mix_%s_noise" % tag], out["mix_%s_offset" % tag] = noisy, noise, np.int64(off)
        o_noisy, o_noise = mix_ref.mix_noise(sp, nz, 3.5, off)
        assert np.array_equal(o_noisy, noisy) and np.array_equal(o_noise, noise), tag
    out["reverb"] = sim.add_reverberation(sp, rir)
    assert np.allclose(mix_ref.add_reverberation(sp, rir), out["reverb"], rtol=0, atol=1e-14)
    early = rir_utils.estimate_early_rir(rir, fs=fs)
    out["early_rir"] = early
    out["reverb_early"] = sim.add_reverberation(sp, early)
    out["clip"] = sim.clipping(sp, min_quantile=0.07, max_quantile=0.93)
    assert np.array_equal(mix_ref.clipping(sp, 0.07, 0.93), out["clip"])
    idx = [3, 4, 11, 17]
    out["ploss_idx"] = np.array(idx)
    out["ploss"] = sim.packet_loss(sp.copy(), fs, idx, 20)
    assert np.array_equal(mix_ref.packet_loss(sp, fs, idx), out["ploss"])
    for f in (8000, 16000, 48000):
        out["hp_taps_%d" % f] = sim.high_pass_taps[f]
        assert np.array_equal(mix_ref.filter_designs(f), sim.high_pass_taps[f])
    from scipy.signal import filtfilt
    out["hp"] = filtfilt(sim.high_pass_taps[fs], 1.0, sp.flatten()).reshape(sp.shape)
    assert np.array_equal(mix_ref.high_pass(sp, fs), out["hp"])

    # whole samples through the reference's process_one_sample(on_the_fly=True), equal-length noise so that the
    # unseeded default_rng() of the on-the-fly branch (:471) draws nothing
    AUDIO["sp.wav"], AUDIO["rir.wav"] = (sp[0], fs), (rir[0], fs)
    AUDIO["nz.wav"] = (synth(rng, sp.shape[1], fs, "noise"), fs)
    out["nz_equal"] = AUDIO["nz.wav"][0][None]
    recipes = [("none", "none"), ("rir.wav", "none"), ("none", "clipping(min=0.05,max=0.95)"),
               ("rir.wav", "packet_loss(packet_loss_indices=[2, 3, 11],packet_duration_ms=20)/clipping(min=0.02,max=0.9)")]
    for i, (rir_uid, aug) in enumerate(recipes):
        info = dict(id="utt_%d" % i, fs=fs, snr_dB=7.25 - 3 * i, speech_uid="sp.wav", noise_uid="nz.wav", rir_uid=rir_uid,
                    augmentation=aug, length=sp.shape[1])
        ident = {k: k for k in AUDIO}
        s, n, f = sim.process_one_sample(info, speech_dic=ident, noise_dic=ident, rir_dic=ident, highpass=True,
                                         on_the_fly=True, max_duration=-1)
        out["sample%d_speech" % i], out["sample%d_noisy" % i] = s, n
        out["sample%d_snr" % i] = np.float64(info["snr_dB"])
        out["sample%d_recipe" % i] = np.array([rir_uid, aug])
        # the oracle's composition of the same steps
        o_s = mix_ref.high_pass(sp, fs)
        o_n = o_s
        if rir_uid != "none":
            o_n = mix_ref.add_reverberation(o_s, rir)
            o_s = mix_ref.add_reverberation(o_s, early)
        o_n, o_noise = mix_ref.mix_noise(o_n, out["nz_equal"], info["snr_dB"], 0)
        for a in aug.split("/"):
            if a.startswith("clipping"):
                lo, hi = a[len("clipping(min="):-1].split(",max=")
                o_n = mix_ref.clipping(o_n, float(lo), float(hi))
            elif a.startswith("packet_loss"):
                import ast
                o_n = mix_ref.packet_loss(o_n, fs, ast.literal_eval(a[a.index("=[") + 1:a.index("]") + 1]))
        o_s, o_n, _ = mix_ref.joint_peak_normalise(o_s, o_n, o_noise)
        assert np.allclose(o_s, s, rtol=0, atol=1e-13) and np.allclose(o_n, n, rtol=0, atol=1e-13), i
    return out


def recipe_fixtures(ds_mod):
    """what DynamicMixingDataset.run_simulation draws (np.random global stream) for seeded calls."""
    tmp = tempfile.mkdtemp()
    rng = np.random.default_rng(7)
    lines = {"speech": [], "noise": [], "rir": [], "wind": [], "length": []}
    for fs in (16000, 48000):
        for i in range(6):
            uid = "sp%d_%d" % (fs, i)
            n = int(rng.integers(fs, 3 * fs))
            AUDIO["/a/%s.wav" % uid] = (np.zeros(n), fs)
            lines["speech"].append("%s %d /a/%s.wav" % (uid, fs, uid))
            lines["length"].append("%s %d" % (uid, n))
        for i in range(5):
            lines["noise"].append("nz%d_%d %d /a/nz%d_%d.wav" % (fs, i, fs, fs, i))
            lines["rir"].append("rir%d_%d %d /a/rir%d_%d.wav" % (fs, i, fs, fs, i))
        lines["wind"].append("wind_noise%d_0 %d /a/wn%d.wav" % (fs, fs, fs))
    paths = {}
    for k, v in lines.items():
        paths[k] = os.path.join(tmp, k + ".scp")
        with open(paths[k], "w") as f:
            f.write("\n".join(v) + "\n")
    ds = ds_mod.DynamicMixingDataset(paths["speech"], paths["noise"], paths["rir"], paths["wind"], paths["length"],
                                     max_duration=40000)
    captured = []

    def capture(info, **kw):
        captured.append(dict(info))
        return None, None, info["fs"]
    ds_mod.process_one_sample = capture
    rows = []
    for seed in range(40):
        np.random.seed(seed)
        fs, real = ds._get_from_index(seed % len(ds))
        uid = ds.speech_uids[fs][real]
        L = min(ds.max_duration, len(AUDIO[ds.speech_source[fs][uid]][0]))
        ds.run_simulation(uid, L, fs)
        info = captured[-1]
        rows.append([seed, seed % len(ds), fs, L, info["noise_uid"], info["rir_uid"], repr(float(info["snr"])),
                     info["augmentation"]])
    scp = {k: open(p).read() for k, p in paths.items()}
    return dict(recipe_rows=np.array(rows, dtype=object).astype(str), recipe_scp_keys=np.array(list(scp.keys())),
                recipe_scp_text=np.array(list(scp.values())),
                recipe_srs=np.array(ds.get_srs()), recipe_lengths=np.array(ds.get_source_length()))


def sampler_fixtures(ds_mod):
    import torch

    class FakeDataset:
        def __init__(self, n, seed):
            r = np.random.default_rng(seed)
            self.srs = [int(v) for v in r.choice([8000, 16000, 48000], size=n, p=[0.2, 0.5, 0.3])]
            self.lens = [int(v) for v in r.integers(4000, 96000, size=n)]

        def get_srs(self):
            return self.srs

        def get_source_length(self):
            return self.lens
    out = {}
    ds = FakeDataset(700, 11)
    out["sampler_srs"], out["sampler_lens"] = np.array(ds.srs), np.array(ds.lens)
    for rank, world, bs, drop in ((0, 1, 4, True), (0, 2, 4, True), (1, 2, 4, True), (3, 8, 2, True), (0, 1, 3, False)):
        s = ds_mod.GroupedBatchSampler(ds, batch_size=bs, rank=rank, world_size=world, drop_last=drop, bucket_size_mult=10)
        for it in range(2):              # the in-place bucket shuffles accumulate over epochs (quirk C.3)
            batches = list(iter(s))
            flat = np.array([i for b in batches for i in b] + [-1] * (len(batches) * bs - sum(len(b) for b in batches)))
            key = "sampler_r%d_w%d_b%d_d%d_it%d" % (rank, world, bs, int(drop), it)
            out[key] = np.array([len(b) for b in batches])
            out[key + "_idx"] = np.array([i for b in batches for i in b])
            assert len(batches) == len(s)
    # collate_fn
    r = np.random.default_rng(5)
    items = [(r.standard_normal((1, n)), r.standard_normal((1, n)), 16000, n) for n in (700, 512, 900)]
    a, b, fs, lens = ds_mod.collate_fn(items)
    out["collate_in_lens"] = np.array([700, 512, 900])
    out["collate_seed"] = np.int64(5)
    out["collate_clean"], out["collate_noisy"] = a.numpy(), b.numpy()
    out["collate_fs"], out["collate_lens"] = fs.numpy(), lens.numpy()
    out["collate_dtypes"] = np.array([str(a.dtype), str(fs.dtype), str(lens.dtype)])
    return out


def main():
    install_stand_ins()
    sim = importlib.import_module("simulation.simulate_data_from_param")
    rir_utils = importlib.import_module("simulation.rir_utils")
    ds_mod = importlib.import_module("baseline_code.dataset")
    out = {}
    out.update(dsp_fixtures(sim, rir_utils))
    out.update(sampler_fixtures(ds_mod))
    out.update(recipe_fixtures(ds_mod))
    path = os.path.join(HERE, "ref_mix.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
