"""GPU parity: on-device dynamic mixing kernels (csrc/mix.hip) vs the float64 numpy / scipy oracle (oracle/mix_ref.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _signals(B, L, seed, silent=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, L))
    for b in range(B):                      # one-pole low-pass + envelope with near-silent stretches
        for i in range(1, L):
            x[b, i] = 0.9 * x[b, i - 1] + x[b, i]
    if silent:
        env = 0.55 + 0.45 * np.sin(2 * np.pi * 3 * np.arange(L) / L + rng.uniform(0, 6, (B, 1)))
        env[:, :L // 7] *= 1e-3
        x = x * env
    return (x / np.abs(x).max(-1, keepdims=True) * 0.8).astype(np.float32)


def test_nonsilence_power_and_mix_noise(lib):
    from urgent2026_challenge_track1_amd import mixing
    from oracle import mix_ref
    B, L = 5, 20000
    sp = _signals(B, L, 1)
    lens = [20000, 19999, 1536, 1000, 17000]                       # < frame_length, exact frame multiples, ragged
    nlens = [9000, 30000, 1536, 400, 17000]
    offs = [3000, 1234, 0, 77, 0]
    snr = [5.0, -3.5, 20.0, 0.0, 12.25]
    nz = _signals(B, 30000, 2, silent=False)
    for b in range(B):
        sp[b, lens[b]:] = 0
    noisy, noise = mixing.mix_noise(torch.tensor(sp).cuda(), torch.tensor(nz).cuda(), nlens, lens, snr, torch.tensor(offs))
    pw = mixing.nonsilence_power(torch.tensor(sp).cuda(), lens).cpu().numpy()
    for b in range(B):
        s64 = sp[b:b + 1, :lens[b]].astype(np.float64)
        ref_p = (s64[mix_ref.detect_non_silence(s64)] ** 2).mean()
        assert abs(pw[b] - ref_p) <= 1e-12 * max(1.0, ref_p), (b, pw[b], ref_p)
        rn, rz = mix_ref.mix_noise(s64, nz[b:b + 1, :nlens[b]].astype(np.float64), snr[b], offs[b])
        assert np.abs(noisy[b, :lens[b]].cpu().numpy() - rn[0]).max() <= 2e-6 * max(1.0, np.abs(rn).max())
        assert np.abs(noise[b, :lens[b]].cpu().numpy() - rz[0]).max() <= 2e-6 * max(1.0, np.abs(rz).max())
        assert torch.all(noisy[b, lens[b]:] == 0)


def test_reverberation_and_high_pass(lib):
    from urgent2026_challenge_track1_amd import mixing
    from oracle import mix_ref
    B, L = 3, 24000
    sp = _signals(B, L, 3)
    lens = [24000, 18001, 9000]
    rng = np.random.default_rng(4)
    rl = [3000, 700, 12000]                                         # one RIR longer than the 1024-tap LDS chunk x several
    rir = np.zeros((B, 12000), np.float32)
    for b in range(B):
        rir[b, :rl[b]] = (rng.standard_normal(rl[b]) * np.exp(-np.arange(rl[b]) / (rl[b] / 6))).astype(np.float32)
        sp[b, lens[b]:] = 0
    out = mixing.add_reverberation(torch.tensor(sp).cuda(), lens, torch.tensor(rir).cuda(), rl).cpu().numpy()
    for b in range(B):
        ref = mix_ref.add_reverberation(sp[b:b + 1, :lens[b]].astype(np.float64), rir[b:b + 1, :rl[b]].astype(np.float64))
        assert np.abs(out[b, :lens[b]] - ref[0]).max() <= 2e-6 * np.abs(ref).max(), b
    for fs in (8000, 16000):
        hp = mixing.high_pass(torch.tensor(sp).cuda(), lens, fs).cpu().numpy()
        for b in range(B):
            ref = mix_ref.high_pass(sp[b:b + 1, :lens[b]].astype(np.float64), fs)
            assert np.abs(hp[b, :lens[b]] - ref[0]).max() <= 5e-6 * max(1.0, np.abs(ref).max()), (fs, b)
            assert np.all(hp[b, lens[b]:] == 0)
    np.testing.assert_allclose(mixing.filter_designs(16000), mix_ref.filter_designs(16000), rtol=0, atol=0)


def test_fir_tap_counts_and_ragged_tiles(lib):
    """the FIR kernel works in blocks of four taps and 1024-tap chunks on 1024-output tiles: tap counts around every boundary,
    a row pitch that is not a tile multiple, lengths that end inside a tile - against numpy's f64 convolution."""
    from urgent2026_challenge_track1_amd import mixing
    rng = np.random.default_rng(11)
    counts = [1, 2, 3, 5, 1023, 1024, 1025, 2051]
    B, L = len(counts), 5003
    sp = rng.standard_normal((B, L)).astype(np.float32)
    lens = [L, 4097, 1024, 1025, 3000, 17, L, 4999]
    rir = np.zeros((B, max(counts)), np.float32)
    for b, n in enumerate(counts):
        rir[b, :n] = rng.standard_normal(n).astype(np.float32)
        sp[b, lens[b]:] = 0
    out = mixing.add_reverberation(torch.tensor(sp).cuda(), lens, torch.tensor(rir).cuda(), counts).cpu().numpy()
    for b, n in enumerate(counts):
        ref = np.convolve(sp[b, :lens[b]].astype(np.float64), rir[b, :n].astype(np.float64))[:lens[b]]
        assert np.abs(out[b, :lens[b]] - ref).max() <= 2e-6 * np.abs(ref).max(), (b, n)
        assert np.all(out[b, lens[b]:] == 0), b


@pytest.mark.parametrize("L,counts", [(50000, [5000, 4096, 100, 1]), (192000, [48000, 70001])])
def test_fft_reverberation_matches_exact_convolution(lib, monkeypatch, L, counts):
    """RIRs of 4,096 taps and more take the FFT form (urse_fft_convolve, float32 transforms of 2^16 .. 2^19 points): within 1e-5 of
    the output's peak of the float64 convolution (the reference's scipy.signal.convolve runs its own float32 FFTs at these sizes),
    zeros behind each utterance's length, and the same result as the direct form to that tolerance."""
    from scipy.signal import fftconvolve
    from urgent2026_challenge_track1_amd import mixing
    rng = np.random.default_rng(21)
    B = len(counts)
    sp = rng.standard_normal((B, L)).astype(np.float32)
    lens = [L - 37 * b for b in range(B)]
    rir = np.zeros((B, max(counts)), np.float32)
    for b, n in enumerate(counts):
        rir[b, :n] = (rng.standard_normal(n) * np.exp(-np.arange(n) / max(1.0, n / 6))).astype(np.float32)
        sp[b, lens[b]:] = 0
    x, h = torch.tensor(sp).cuda(), torch.tensor(rir).cuda()
    assert max(counts) >= mixing.FFT_CONV_MIN_TAPS
    out = mixing.add_reverberation(x, lens, h, counts).cpu().numpy()
    monkeypatch.setattr(mixing, "FFT_CONV_MIN_TAPS", 1 << 30)
    direct = mixing.add_reverberation(x, lens, h, counts).cpu().numpy()
    for b, n in enumerate(counts):
        ref = fftconvolve(sp[b, :lens[b]].astype(np.float64), rir[b, :n].astype(np.float64))[:lens[b]]
        peak = np.abs(ref).max()
        assert np.abs(out[b, :lens[b]] - ref).max() <= 1e-5 * peak, (b, n, np.abs(out[b, :lens[b]] - ref).max() / peak)
        assert np.abs(out[b, :lens[b]] - direct[b, :lens[b]]).max() <= 1e-5 * peak, (b, n)
        assert np.all(out[b, lens[b]:] == 0), b


def test_fft_convolve_shared_filter_and_argument_checks(lib):
    """urse_fft_convolve through the C ABI with ONE filter for the whole batch (taps_per_utt = 0), and its refusals: a workspace that
    is too small, x == y, a length bound above the row pitch."""
    import ctypes
    from urgent2026_challenge_track1_amd import _lib
    from urgent2026_challenge_track1_amd._lib import call, stream_ptr
    rng = np.random.default_rng(5)
    B, L, nt = 3, 20000, 4500
    sp = rng.standard_normal((B, L)).astype(np.float32)
    h = (rng.standard_normal(nt) * np.exp(-np.arange(nt) / 700.0)).astype(np.float32)
    lens = [L, L - 1, 12345]
    x = torch.tensor(sp).cuda()
    taps = torch.tensor(h[None]).cuda()
    y = torch.full_like(x, 7.0)
    li = torch.tensor(lens, dtype=torch.int32).cuda()
    ni = torch.tensor([nt], dtype=torch.int32).cuda()
    nb = ctypes.c_int64()
    assert _lib.load().urse_fft_convolve_workspace_bytes(B, L, nt, ctypes.addressof(nb)) == 0
    ws = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
    call("fft_convolve", x, li, B, x.stride(0), taps, ni, nt, 0, y, L, nt, ws, ws.numel(), stream_ptr())
    out = y.cpu().numpy()
    for b in range(B):
        ref = np.convolve(sp[b, :lens[b]].astype(np.float64), h.astype(np.float64))[:lens[b]]
        assert np.abs(out[b, :lens[b]] - ref).max() <= 1e-5 * np.abs(ref).max(), b
        assert np.all(out[b, lens[b]:] == 0), b
    for bad in (lambda: call("fft_convolve", x, li, B, x.stride(0), taps, ni, nt, 0, y, L, nt, ws, 1024, stream_ptr()),
                lambda: call("fft_convolve", x, li, B, x.stride(0), taps, ni, nt, 0, x, L, nt, ws, ws.numel(), stream_ptr()),
                lambda: call("fft_convolve", x, li, B, x.stride(0), taps, ni, nt, 0, y, L + 1, nt, ws, ws.numel(), stream_ptr())):
        with pytest.raises(_lib.UrseError):
            bad()


def test_clipping_packet_loss_peak_norm(lib):
    from urgent2026_challenge_track1_amd import mixing
    from oracle import mix_ref
    B, L, fs = 4, 48000, 16000
    sp = _signals(B, L, 5)
    lens = [48000, 31111, 5, 20000]
    qmin, qmax = [0.0, 0.05, 0.1, 0.02], [0.9, 0.95, 0.75, 1.0]
    for b in range(B):
        sp[b, lens[b]:] = 0
    x = torch.tensor(sp).cuda()
    bounds = mixing.clipping(x, lens, qmin, qmax).cpu().numpy()
    for b in range(B):
        s64 = sp[b:b + 1, :lens[b]].astype(np.float64)
        ref = mix_ref.clipping(s64, qmin[b], qmax[b])
        mn, mx = np.quantile(s64, [qmin[b], qmax[b]], axis=-1)
        assert abs(bounds[b, 0] - mn[0]) <= 1e-6 and abs(bounds[b, 1] - mx[0]) <= 1e-6, (b, bounds[b], mn, mx)
        assert np.abs(x[b, :lens[b]].cpu().numpy() - ref[0]).max() <= 1e-6
    idx = [[0, 3, 7], [], [1], [149]]
    y = torch.tensor(sp).cuda()
    mixing.packet_loss(y, fs, idx)
    for b in range(B):
        assert np.array_equal(y[b].cpu().numpy(), mix_ref.packet_loss(sp[b:b + 1], fs, idx[b])[0])
    a, n_, z = torch.tensor(sp).cuda(), torch.tensor(sp * 1.7).cuda(), torch.tensor(sp * 0.3).cuda()
    mixing.joint_peak_normalise(a, n_, z)
    for b in range(B):
        ra, rn, rz = mix_ref.joint_peak_normalise(sp[b].astype(np.float64), sp[b].astype(np.float64) * 1.7, sp[b].astype(np.float64) * 0.3)
        assert np.abs(n_[b].cpu().numpy() - rn).max() <= 1e-6 and np.abs(a[b].cpu().numpy() - ra).max() <= 1e-6
        assert abs(float(n_[b].abs().max()) - 0.9) <= 1e-6


def test_simulate_batch_matches_oracle_chain(lib):
    """the reference's per-utterance order (high-pass, reverberation + early-RIR target, additive noise, clipping,
    packet loss, joint peak normalisation) on a ragged batch."""
    from urgent2026_challenge_track1_amd import mixing
    from oracle import mix_ref
    B, L, fs = 3, 20000, 8000
    sp, nz = _signals(B, L, 7), _signals(B, 26000, 8, silent=False)
    lens, nlens, offs, snr = [20000, 15000, 12001], [26000, 4000, 12001], [1500, 321, 0], [3.0, 10.0, -2.0]
    rng = np.random.default_rng(9)
    rl = [900, 2500, 400]
    rir = np.zeros((B, 2500), np.float32)
    for b in range(B):
        sp[b, lens[b]:] = 0
        h = rng.standard_normal(rl[b]) * np.exp(-np.arange(rl[b]) / (rl[b] / 5))
        h[:20] *= 0.01
        rir[b, :rl[b]] = h
    stops = [mixing.early_rir_stop(rir[b, :rl[b]], fs) for b in range(B)]
    qmin, qmax = [0.02, 0.0, 0.1], [0.95, 0.9, 0.99]
    pidx = [[1, 5], [], [0, 2, 30]]
    s, n, fs_out, z = mixing.simulate_batch(torch.tensor(sp).cuda(), lens, torch.tensor(nz).cuda(), nlens, fs, snr, torch.tensor(offs),
                                            rir=torch.tensor(rir).cuda(), rir_lens=rl, rir_early_stops=stops,
                                            clip_quantiles=(qmin, qmax), packet_loss_indices=pidx)
    assert fs_out == fs
    for b in range(B):
        x = mix_ref.high_pass(sp[b:b + 1, :lens[b]].astype(np.float64), fs)
        h = rir[b:b + 1, :rl[b]].astype(np.float64)
        he = h.copy()
        he[:, stops[b]:] = 0
        noisy, clean = mix_ref.add_reverberation(x, h), mix_ref.add_reverberation(x, he)
        noisy, noise = mix_ref.mix_noise(noisy, nz[b:b + 1, :nlens[b]].astype(np.float64), snr[b], offs[b])
        noisy = mix_ref.packet_loss(mix_ref.clipping(noisy, qmin[b], qmax[b]), fs, pidx[b])
        clean, noisy, noise = mix_ref.joint_peak_normalise(clean, noisy, noise)
        for got, ref in ((s, clean), (n, noisy), (z, noise)):
            assert np.abs(got[b, :lens[b]].cpu().numpy() - ref[0]).max() <= 2e-5, b


def test_simulate_recipes_matches_the_reference_simulator(lib):
    """``mixing.simulate_recipes`` (what DynamicMixingDataset batches go through on the device) against samples the
    reference's own ``process_one_sample(on_the_fly=True)`` produced (tests/golden/ref_mix.npz): the four recipes as ONE
    batch - none, RIR, clipping, RIR + packet loss then clipping."""
    import ast
    import os
    from urgent2026_challenge_track1_amd import mixing
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_mix.npz"))
    fs, sp, nz, rir = int(g["fs"]), g["sp"], g["nz_equal"], g["rir"]
    L, R = sp.shape[1], rir.shape[1]
    recipes, rir_lens = [], []
    for i in range(4):
        rir_uid, aug = g["sample%d_recipe" % i].tolist()
        params, order = {}, []
        for a in aug.split("/"):
            if a.startswith("clipping"):
                lo, hi = a[len("clipping(min="):-1].split(",max=")
                params["clipping"] = dict(min_quantile=float(lo), max_quantile=float(hi))
                order.append("clipping")
            elif a.startswith("packet_loss"):
                params["packet_loss"] = dict(packet_loss_indices=ast.literal_eval(a[a.index("=[") + 1:a.index("]") + 1]),
                                             packet_duration_ms=20)
                order.append("packet_loss")
        recipes.append(dict(snr=float(g["sample%d_snr" % i]), params=params, order=order, noise_offset=0, highpass=True))
        rir_lens.append(R if rir_uid != "none" else 0)
    B = 4
    t = lambda a: torch.tensor(np.repeat(a.astype(np.float32), B, axis=0)).cuda()
    rir_b = t(rir)
    for b in range(B):
        if rir_lens[b] == 0:
            rir_b[b] = 0
    stops = [mixing.early_rir_stop(rir, fs)] * B
    skipped = {}
    speech, noisy = mixing.simulate_recipes(t(sp), [L] * B, t(nz), [L] * B, rir_b, rir_lens, stops, fs, recipes, skipped)
    assert skipped == {}
    for i in range(4):
        rs, rn = g["sample%d_speech" % i][0], g["sample%d_noisy" % i][0]
        es = np.abs(speech[i].cpu().numpy() - rs).max()
        en = np.abs(noisy[i].cpu().numpy() - rn).max()
        # inputs are rounded to f32 on the way in (the reference computes in f64 from f64 files); 0.9-peak signals
        assert es <= 2e-5 and en <= 2e-5, (i, es, en)


def test_bandwidth_limitation_polyphase_matches_scipy(lib):
    """the polyphase branch of the reference's bandwidth limitation (librosa.resample(res_type="polyphase") = scipy.signal
    .resample_poly + fix_length, down and back up) on the device vs scipy on the CPU."""
    import math
    from scipy.signal import resample_poly
    from urgent2026_challenge_track1_amd import mixing
    x = _signals(1, 24000, 7)[0]
    for fs, fs_new in ((48000, 16000), (48000, 22050), (16000, 8000)):
        g = math.gcd(fs, fs_new)
        down = resample_poly(x.astype(np.float64), fs_new // g, fs // g)
        n1 = int(math.ceil(len(x) * fs_new / fs))
        down = down[:n1] if len(down) >= n1 else np.pad(down, (0, n1 - len(down)))
        up = resample_poly(down, fs // g, fs_new // g)
        n2 = int(math.ceil(n1 * fs / fs_new))
        up = up[:n2] if len(up) >= n2 else np.pad(up, (0, n2 - len(up)))
        ref = up[:len(x)] if len(up) >= len(x) else np.pad(up, (0, len(x) - len(up)))
        got = mixing.bandwidth_limitation_polyphase(torch.tensor(x[None]).cuda(), fs, fs_new)[0].cpu().numpy()
        assert got.shape == x.shape and np.abs(got - ref).max() <= 2e-5, (fs, fs_new, np.abs(got - ref).max())


def test_bandwidth_limitation_fft_matches_scipy(lib):
    """the scipy branch of the reference's bandwidth limitation (librosa.resample(res_type="scipy") = scipy.signal.resample to
    ceil(n * ratio) samples, down and back up, cropped) on the device vs scipy on the CPU - odd, even and prime lengths."""
    import math
    from scipy.signal import resample
    from urgent2026_challenge_track1_amd import mixing
    for L, fs, fs_new in ((24000, 48000, 16000), (24001, 48000, 22050), (9973, 16000, 8000), (12000, 44100, 32000), (192000, 48000, 16000),
                          (131071, 48000, 44100)):
        x = _signals(1, L, 9)[0]
        n1 = int(math.ceil(L * float(fs_new) / fs))
        down = resample(x.astype(np.float64), n1)
        up = resample(down, int(math.ceil(n1 * float(fs) / fs_new)))[:L]
        got = mixing.bandwidth_limitation_fft(torch.tensor(x[None]).cuda(), fs, fs_new)[0].cpu().numpy()
        assert got.shape == x.shape and np.abs(got - up).max() <= 2e-5, (L, fs, fs_new, np.abs(got - up).max())
    # a batch of rows through one launch, and the single transform against scipy (up- and down-sampling, odd / even / prime lengths)
    xs = _signals(3, 9973, 10)
    for num in (4987, 9973, 12001, 19946):
        got = mixing._fft_resample(torch.tensor(xs).cuda(), num).cpu().numpy()
        ref = resample(xs.astype(np.float64), num, axis=1)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-5, (num, np.abs(got - ref).max())


def test_sources_at_another_rate_are_resampled_on_the_device(lib):
    """RawMixBatch.materialise with noise / RIR rows recorded at 48 kHz in a 16 kHz batch (`_load` used to raise):
    the rows are resampled with the soxr-HQ-specification filter on the device and the mix equals the oracle chain run on
    oracle-resampled sources (oracle/metrics_ref.resample_soxr_hq_spec + oracle/mix_ref)."""
    from oracle import metrics_ref, mix_ref
    from urgent2026_challenge_track1_amd.dataset import RawMixBatch
    from urgent2026_challenge_track1_amd.mixing import early_rir_stop
    fs, L = 16000, 12000
    sp = _signals(2, L, 21)
    nz48 = _signals(2, 60001, 22, silent=False)
    rng = np.random.default_rng(23)
    rir48 = (rng.standard_normal(4801) * np.exp(-np.arange(4801) / 900.0)).astype(np.float32)[None]
    rir16 = (rng.standard_normal(1500) * np.exp(-np.arange(1500) / 300.0)).astype(np.float32)[None]
    rec = lambda snr, off, rir: dict(snr=snr, noise_offset=off, order=[], params={}, highpass=True, rir_uid=rir, wind=False)
    items = [dict(speech=sp[0:1], noise=nz48[0:1], rir=rir48, recipe=rec(5.0, 1000, "r48"), fs=fs, length=L, noise_fs=48000, rir_fs=48000),
             dict(speech=sp[1:2], noise=nz48[1:2, :30000], rir=rir16, recipe=rec(-2.0, 0, "r16"), fs=fs, length=L, noise_fs=16000, rir_fs=16000)]
    clean, noisy, fs_t, lens = RawMixBatch(items).materialise("cuda")
    assert int(fs_t) == fs and lens.tolist() == [L, L]
    for b, it in enumerate(items):
        n64 = it["noise"][0].astype(np.float64)
        r64 = it["rir"][0].astype(np.float64)
        if it["noise_fs"] != fs:
            n64 = metrics_ref.resample_soxr_hq_spec(n64, it["noise_fs"], fs)[:-(-len(n64) * fs // it["noise_fs"])]
            r64 = metrics_ref.resample_soxr_hq_spec(r64, it["rir_fs"], fs)[:-(-len(r64) * fs // it["rir_fs"])]
        s = mix_ref.high_pass(it["speech"].astype(np.float64), fs)
        early = r64.copy()
        early[early_rir_stop(r64, fs):] = 0
        rev, s_early = mix_ref.add_reverberation(s, r64[None]), mix_ref.add_reverberation(s, early[None])
        mixed, noise = mix_ref.mix_noise(rev, n64[None], it["recipe"]["snr"], it["recipe"]["noise_offset"])
        s_early, mixed, _ = mix_ref.joint_peak_normalise(s_early, mixed, noise)
        e_n = np.abs(noisy[b, 0].cpu().numpy() - mixed[0]).max()
        e_c = np.abs(clean[b, 0].cpu().numpy() - s_early[0]).max()
        assert e_n <= 5e-5 and e_c <= 5e-5, (b, e_n, e_c)


def test_bandwidth_limitation_resampy_matches_oracle(lib):
    """the kaiser_best / kaiser_fast branches of the bandwidth limitation (librosa -> resampy.resample, table-interpolated windowed
    sinc; resampy itself is absent: oracle/mix_ref.resampy_resample restates its loop, unpinned) on the device vs the float64 oracle,
    down and back up for the rate pairs the recipes draw, odd and even lengths."""
    from oracle import mix_ref
    from urgent2026_challenge_track1_amd import mixing
    for L, fs, fs_new, name in ((24000, 48000, 16000, "kaiser_best"), (24001, 48000, 22050, "kaiser_fast"), (9973, 16000, 8000, "kaiser_best"),
                                (12000, 44100, 32000, "kaiser_fast"), (30001, 48000, 8000, "kaiser_fast"), (20000, 22050, 16000, "kaiser_best")):
        x = _signals(1, L, 11)[0]
        ref = mix_ref.bandwidth_limitation_resampy(x, fs, fs_new, name)
        got = mixing.bandwidth_limitation_resampy(torch.tensor(x[None]).cuda(), fs, fs_new, name)[0].cpu().numpy()
        assert got.shape == x.shape and np.abs(got - ref).max() <= 2e-6, (L, fs, fs_new, name, np.abs(got - ref).max())
    # a recipe that draws it is now applied, not counted as skipped
    sp = _signals(2, 16000, 12)
    nz = _signals(2, 20000, 13, silent=False)
    rec = lambda rt: dict(snr=10.0, noise_offset=100, order=["bandwidth_limitation"], highpass=True, rir_uid="none", wind=False, length=16000,
                          params={"bandwidth_limitation": dict(res_type=rt, fs_new=8000)})
    skipped = {}
    clean, noisy = mixing.simulate_recipes(torch.tensor(sp).cuda(), [16000, 16000], torch.tensor(nz).cuda(), [20000, 20000], None, [0, 0],
                                           [0, 0], 16000, [rec("kaiser_best"), rec("kaiser_fast")], skipped)
    assert not skipped and torch.isfinite(noisy).all()
    X = torch.fft.rfft(noisy.cpu().double(), dim=1).abs()          # (test-side check only) nothing left above 4.2 kHz
    assert float(X[:, 4300:].max()) <= 1e-3 * float(X.max())
