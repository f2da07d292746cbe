"""CPU: the data path against vectors produced by the reference's OWN classes (tests/golden/ref_mix.npz, written by
tests/golden/make_golden_mix.py in the build container): shard rule / batch order of GroupedBatchSampler, collate_fn, the
recipe draw of DynamicMixingDataset, and the CPU oracle of the simulator DSP (oracle/mix_ref.py)."""
import ast
import os

import math

import numpy as np
import pytest
import torch

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_mix.npz"), allow_pickle=False)


class _Fake:
    def __init__(self, srs, lens):
        self.srs, self.lens = [int(v) for v in srs], [int(v) for v in lens]

    def get_srs(self):
        return self.srs

    def get_source_length(self):
        return self.lens


@pytest.mark.parametrize("rank,world,bs,drop", [(0, 1, 4, True), (0, 2, 4, True), (1, 2, 4, True), (3, 8, 2, True), (0, 1, 3, False)])
def test_sampler_batches_equal_the_reference_classes(rank, world, bs, drop):
    """same shard, same buckets, same three shuffles (random.seed(epoch + rank); in place, accumulating over epochs)."""
    from urgent2026_challenge_track1_amd.dataset import GroupedBatchSampler
    ds = _Fake(GOLD["sampler_srs"], GOLD["sampler_lens"])
    s = GroupedBatchSampler(ds, batch_size=bs, rank=rank, world_size=world, drop_last=drop, bucket_size_mult=10)
    for it in range(2):
        key = "sampler_r%d_w%d_b%d_d%d_it%d" % (rank, world, bs, int(drop), it)
        batches = list(iter(s))
        assert len(batches) == len(s)
        assert [len(b) for b in batches] == GOLD[key].tolist()
        assert [i for b in batches for i in b] == GOLD[key + "_idx"].tolist()


def test_sampler_max_batches_truncates_every_rank_to_the_same_count():
    from urgent2026_challenge_track1_amd.dataset import GroupedBatchSampler
    ds = _Fake(GOLD["sampler_srs"], GOLD["sampler_lens"])
    ss = [GroupedBatchSampler(ds, batch_size=3, rank=r, world_size=4, drop_last=True, bucket_size_mult=7) for r in range(4)]
    counts = [len(s) for s in ss]
    assert len(set(counts)) > 1          # the shard rule does give ranks different batch counts
    for s in ss:
        s.max_batches = min(counts)
    assert {len(s) for s in ss} == {min(counts)} and {len(list(iter(s))) for s in ss} == {min(counts)}


def test_collate_equals_the_reference_function():
    from urgent2026_challenge_track1_amd.dataset import collate_fn
    r = np.random.default_rng(int(GOLD["collate_seed"]))
    items = [(r.standard_normal((1, n)), r.standard_normal((1, n)), 16000, n) for n in GOLD["collate_in_lens"].tolist()]
    a, b, fs, lens = collate_fn(items)
    assert [str(a.dtype), str(fs.dtype), str(lens.dtype)] == GOLD["collate_dtypes"].tolist()
    assert np.array_equal(a.numpy(), GOLD["collate_clean"]) and np.array_equal(b.numpy(), GOLD["collate_noisy"])
    assert int(fs) == int(GOLD["collate_fs"]) and fs.dim() == 0 and lens.tolist() == GOLD["collate_lens"].tolist()


def _dm_dataset(tmp_path, reader=None, frames=None, **kw):
    from urgent2026_challenge_track1_amd.dataset import DynamicMixingDataset
    paths = {}
    for k, text in zip(GOLD["recipe_scp_keys"].tolist(), GOLD["recipe_scp_text"].tolist()):
        paths[k] = str(tmp_path / (k + ".scp"))
        with open(paths[k], "w") as f:
            f.write(text)
    return DynamicMixingDataset(paths["speech"], paths["noise"], paths["rir"], paths["wind"], paths["length"],
                                reader=reader, frames=frames, **kw)


def test_recipe_draw_equals_the_reference(tmp_path):
    """np.random.seed(s) -> the reference's run_simulation + generate_data_param.process_one_sample and ours draw the same
    noise / RIR / SNR / augmentation string (incl. wind-noise parameters, codec and bandwidth draws, packet indices)."""
    from urgent2026_challenge_track1_amd.dataset import draw_recipe
    ds = _dm_dataset(tmp_path, max_duration=40000)
    assert ds.get_srs() == GOLD["recipe_srs"].tolist() and ds.get_source_length() == GOLD["recipe_lengths"].tolist()
    kinds = set()
    for seed, index, fs, L, noise_uid, rir_uid, snr, aug in GOLD["recipe_rows"].tolist():
        np.random.seed(int(seed))
        assert ds._get_from_index(int(index))[0] == int(fs)
        r = draw_recipe(int(L), int(fs), ds.noise_source, ds.rirs, ds.wind_noises)
        assert (r["noise_uid"], r["rir_uid"], r["augmentation"]) == (noise_uid, rir_uid, aug), seed
        assert float(r["snr"]) == float(snr)
        kinds.update(a.split("(")[0].split("-")[0] for a in aug.split("/") if a)
        if "packet_loss" in r["params"]:
            txt = [a for a in aug.split("/") if a.startswith("packet_loss")][0]
            assert r["params"]["packet_loss"]["packet_loss_indices"] == ast.literal_eval(txt[txt.index("=[") + 1:txt.index("]") + 1])
    assert {"none", "clipping", "packet_loss", "codec", "bandwidth_limitation", "wind_noise"} <= kinds, kinds


def test_dynamic_mixing_dataset_serves_raw_sources_and_recipe(tmp_path):
    from urgent2026_challenge_track1_amd.dataset import collate_dynamic
    lens = dict(zip([l.split()[2] for l in GOLD["recipe_scp_text"].tolist()[0].strip().splitlines()],
                    GOLD["recipe_lengths"].tolist()))

    def reader(path):
        fs = 48000 if "48000" in path else 16000
        n = lens.get(path, 30000 if "nz" in path or "wn" in path else 3000)
        r = np.random.default_rng(abs(hash(path)) % 1000)
        return r.standard_normal((1, n)).astype(np.float32), fs
    ds = _dm_dataset(tmp_path, reader=reader, frames=lambda p: reader(p)[0].shape[1], max_duration=40000)
    np.random.seed(3)
    items = [ds[i] for i in (0, 1, 2)]
    for it in items:
        assert it["speech"].shape == (1, it["length"]) and it["length"] <= 40000 and it["fs"] == 16000
        assert it["noise"].shape[1] <= 40000 and (it["rir"] is None) == (it["recipe"]["rir_uid"] == "none")
    batch = collate_dynamic(items)
    assert batch.speech.shape == (3, max(batch.lengths)) and batch.fs == 16000 and len(batch.recipes) == 3


def test_mix_oracle_equals_the_reference_simulator():
    """oracle/mix_ref.py (the checker of the HIP mixing kernels) against outputs of the reference's own functions."""
    from oracle import mix_ref
    sp, rir, fs = GOLD["sp"], GOLD["rir"], int(GOLD["fs"])
    for tag, nz in (("short", GOLD["nz_short"]), ("long", GOLD["nz_long"])):
        noisy, noise = mix_ref.mix_noise(sp, nz, 3.5, int(GOLD["mix_%s_offset" % tag]))
        assert np.array_equal(noisy, GOLD["mix_%s_noisy" % tag]) and np.array_equal(noise, GOLD["mix_%s_noise" % tag])
    assert np.allclose(mix_ref.add_reverberation(sp, rir), GOLD["reverb"], rtol=0, atol=1e-14)
    assert np.array_equal(mix_ref.clipping(sp, 0.07, 0.93), GOLD["clip"])
    assert np.array_equal(mix_ref.packet_loss(sp, fs, GOLD["ploss_idx"].tolist()), GOLD["ploss"])
    assert np.array_equal(mix_ref.high_pass(sp, fs), GOLD["hp"])
    for f in (8000, 16000, 48000):
        assert np.array_equal(mix_ref.filter_designs(f), GOLD["hp_taps_%d" % f])
    from urgent2026_challenge_track1_amd.mixing import early_rir_stop, filter_designs
    stop = early_rir_stop(rir, fs)
    assert np.array_equal(np.where(np.arange(rir.shape[1]) < stop, rir, 0.0), GOLD["early_rir"])
    assert np.array_equal(filter_designs(48000), GOLD["hp_taps_48000"])


def _speechy(rng, n, ch, bits):
    x = np.cumsum(rng.standard_normal((n, ch)), axis=0)
    x = x / np.abs(x).max() * (0.7 * (1 << (bits - 1)))
    return np.round(x).astype(np.int64)


def test_flac_decoder_reads_every_subframe_and_stereo_mode(tmp_path):
    """the library's host-side FLAC decoder (csrc/flac.hip) against streams written by tests/flac_writer.py: every subframe
    type, Rice with 4- / 5-bit parameters, partition orders, an escape partition, wasted bits, all stereo decorrelations,
    16- and 24-bit, an odd last block.  (No FLAC file or encoder exists in the image: encoder and decoder both follow the
    published format specification; agreement is self-consistency, not a pin against libFLAC.)"""
    from tests import flac_writer as fw
    from urgent2026_challenge_track1_amd.audio_io import audio_frames, read_audio
    from urgent2026_challenge_track1_amd.flac import decode_flac
    rng = np.random.default_rng(0)
    # mono, 16 bit
    x = _speechy(rng, 4096 + 1152 + 256 + 192 + 100, 1, 16)
    x[4096 + 1152:4096 + 1152 + 256] = 1234                          # a constant block
    x[4096 + 1152 + 256:4096 + 1152 + 256 + 192] &= ~0x7                # three wasted bits
    frames = [(4096, "indep", [("fixed2", dict(method=0, part_order=3))]),
              (1152, "indep", [("lpc", dict(coefs=[1700, -700], shift=10, prec=12, method=1, part_order=2, escape_part=1))]),
              (256, "indep", [("constant", {})]),
              (192, "indep", [("fixed1", dict(wasted=3, method=0, part_order=0))]),
              (100, "indep", [("verbatim", {})])]
    data = fw.encode(x, 16000, 16, frames)
    y, fs = decode_flac(data)
    assert fs == 16000 and y.shape == (len(x), 1) and np.array_equal(np.round(y[:, 0] * 32768).astype(np.int64), x[:, 0])
    # stereo, 24 bit, every channel mode and the remaining fixed orders
    s = _speechy(rng, 4 * 576 + 77, 2, 24)
    s[:, 1] = s[:, 0] // 2 + _speechy(rng, len(s), 1, 24)[:, 0] // 8
    frames = [(576, "mid_side", [("fixed4", dict(method=0, part_order=2)), ("fixed3", dict(method=0, part_order=0))]),
              (576, "left_side", [("fixed0", dict(method=1, part_order=1)), ("fixed2", dict(method=0, part_order=1))]),
              (576, "right_side", [("fixed1", dict(method=0, part_order=0)), ("verbatim", {})]),
              (576, "indep", [("fixed3", dict(method=0, part_order=4)), ("lpc", dict(coefs=[900, 60, -100], shift=10, prec=11, method=0,
                                                                                       part_order=0))]),
              (77, "mid_side", [("fixed1", dict(method=0, part_order=0)), ("fixed1", dict(method=0, part_order=0))])]
    data = fw.encode(s, 48000, 24, frames)
    y, fs = decode_flac(data)
    assert fs == 48000 and np.array_equal(np.round(y * (1 << 23)).astype(np.int64), s)
    # through the file API the datasets use
    p = tmp_path / "a.flac"
    p.write_bytes(data)
    z, fs2 = read_audio(str(p))
    assert fs2 == 48000 and z.shape == (1, len(s)) and np.array_equal(z[0], y[:, 0]) and audio_frames(str(p)) == len(s)


def test_wav_reader_agrees_with_scipy(tmp_path):
    from scipy.io import wavfile
    from urgent2026_challenge_track1_amd.audio_io import audio_frames, read_audio
    rng = np.random.default_rng(1)
    x16 = (rng.standard_normal((3000, 2)) * 8000).astype(np.int16)
    wavfile.write(str(tmp_path / "s16.wav"), 22050, x16)
    y, fs = read_audio(str(tmp_path / "s16.wav"))
    assert fs == 22050 and np.array_equal(y[0], x16[:, 0].astype(np.float32) / 32768.0) and audio_frames(str(tmp_path / "s16.wav")) == 3000
    xf = rng.standard_normal(2000).astype(np.float32) * 0.1
    wavfile.write(str(tmp_path / "f32.wav"), 48000, xf)
    y, fs = read_audio(str(tmp_path / "f32.wav"))
    assert fs == 48000 and np.array_equal(y[0], xf)
    x32 = (rng.standard_normal(1000) * 1e8).astype(np.int32)
    wavfile.write(str(tmp_path / "s32.wav"), 8000, x32)
    y, fs = read_audio(str(tmp_path / "s32.wav"))
    assert np.array_equal(y[0], x32.astype(np.float32) / 2147483648.0)


def test_sources_at_a_higher_rate_are_served_raw_with_their_rate(tmp_path):
    """`_pick_source` (the reference's select_sample) falls back to noise / RIR files of a HIGHER rate when none exists at the speech
    rate - the common case on the real corpus.  The item then carries the source at ITS rate (no max_duration crop: the
    reference's read_audio returns before it on that path, simulate_data_from_param.py:350-352); the noise offset is drawn on the
    length after resampling (ceil(n fs / src)) and the source is cut to the window that offset selects plus filter margins
    (crop_for_resampling: bit-identical samples inside the window)."""
    from urgent2026_challenge_track1_amd.dataset import DynamicMixingDataset, collate_dynamic
    rows = {"sp": ["s%d 16000 sp16_%d" % (i, i) for i in range(3)], "nz": ["n%d 48000 nz48_%d" % (i, i) for i in range(2)],
            "rir": ["r0 48000 rir48_0"], "wn": ["wind_noise0 48000 wn48_0"], "len": ["s%d 20000" % i for i in range(3)]}
    paths = {}
    for k, v in rows.items():
        paths[k] = str(tmp_path / (k + ".scp"))
        (tmp_path / (k + ".scp")).write_text("\n".join(v) + "\n")

    def reader(path):
        fs = 48000 if "48" in path else 16000
        n = {"sp": 20000, "nz": 150001, "ri": 9001, "wn": 150001}[path[:2]]
        return np.random.default_rng(len(path)).standard_normal((1, n)).astype(np.float32), fs
    ds = DynamicMixingDataset(paths["sp"], paths["nz"], paths["rir"], paths["wn"], paths["len"], max_duration=40000, reader=reader,
                              frames=lambda p: reader(p)[0].shape[1])
    np.random.seed(5)
    items = [ds[i] for i in range(3)]
    for it in items:
        assert it["fs"] == 16000 and it["noise_fs"] == 48000
        assert DynamicMixingDataset.resampled_length(150001, 48000, 16000) == 50001
        n_src = it["noise"].shape[1]                       # 20,000 output samples need 60,000 source samples + margins
        assert 60000 <= n_src <= min(150001, 60000 + 2 * 4096 + 8)
        assert 0 <= it["recipe"]["noise_offset"] and it["recipe"]["noise_offset"] + it["length"] <= DynamicMixingDataset.resampled_length(n_src, 48000, 16000)
        if it["rir"] is not None:
            assert it["rir_fs"] == 48000 and it["rir"].shape[1] == 9001
    batch = collate_dynamic(items)
    assert batch.noise_fs == [48000] * 3 and batch.noise.shape[0] == 3 and batch.noise.shape[1] <= 60000 + 2 * 4096 + 8


def test_cropping_a_source_before_resampling_changes_nothing_inside_the_window():
    """DynamicMixingDataset.crop_for_resampling: resample(whole file)[offset : offset + L] == resample(cropped file)[offset' : offset' + L]
    for the polyphase resampler (checked on the float64 oracle filter; the device kernel evaluates the same taps)."""
    from oracle import metrics_ref
    from urgent2026_challenge_track1_amd.dataset import DynamicMixingDataset
    rng = np.random.default_rng(8)
    for src_fs, fs, n, off, L in ((48000, 16000, 600001, 150000, 20000), (44100, 16000, 400000, 91234, 16000), (48000, 16000, 90000, 5, 20000),
                                  (22050, 16000, 300000, 190000, 24000)):
        x = rng.standard_normal((1, n))
        full = metrics_ref.resample_soxr_hq_spec(x[0], src_fs, fs)
        assert off + L <= len(full)
        xc, off2 = DynamicMixingDataset.crop_for_resampling(x, src_fs, fs, off, L)
        part = metrics_ref.resample_soxr_hq_spec(xc[0], src_fs, fs)
        assert xc.shape[1] < n or (off2 == off and xc.shape[1] == n)
        assert np.abs(part[off2:off2 + L] - full[off:off + L]).max() <= 1e-12, (src_fs, fs, np.abs(part[off2:off2 + L] - full[off:off + L]).max())
        assert xc.shape[1] <= (L * src_fs) // fs + 2 * 4096 + 2 * (src_fs // math.gcd(src_fs, fs)) + 2
