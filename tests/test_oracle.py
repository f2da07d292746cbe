"""CPU: pins of the oracle itself (architecture counts from the reference yaml, independent numpy DFT,
committed golden vectors)."""
import os

import numpy as np
import torch

from oracle import bsrnn_ref, losses_ref, stft_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_parameter_counts_match_reference_yaml():
    """conf/models/BSRNN_baseline.yaml:30-32: 32.0456657409668 Mi (16 kHz, 27 bands), 36.01795196533203 Mi (48 kHz),
    '~38M' in total; counts exclude norm layers."""
    m = bsrnn_ref.BSRNN_SE(196, 6)
    total = sum(p.numel() for p in m.parameters())
    assert total == 37800844
    import re
    is_norm = lambda n: ("norm" in n) or re.search(r"mlp_(mask|residual)\.\d+\.0\.", n) is not None
    non_norm = sum(p.numel() for n, p in m.named_parameters() if not is_norm(n))
    assert non_norm == 37767560 and abs(non_norm / 2 ** 20 - 36.01795196533203) < 1e-12

    def used(n, K):   # parameters touched when only the first K bands run
        for tag in ("band_split.norm.", "band_split.fc.", "mlp_mask.", "mlp_residual."):
            if tag in n:
                return int(n.split(tag)[1].split(".")[0]) < K
        return True
    K16 = bsrnn_ref.num_bands_for(161, bsrnn_ref.SUBBANDS_481)
    assert K16 == 27
    n16 = sum(p.numel() for n, p in m.named_parameters() if not is_norm(n) and used(n, K16))
    assert n16 == 33602316 and abs(n16 / 2 ** 20 - 32.0456657409668) < 1e-12


def test_band_rule():
    exp = {41: 9 + 1, 81: 20, 161: 27, 221: 28, 241: 29, 321: 31, 442: 34, 481: 34}   # F -> K  
    for F, K in exp.items():
        got = bsrnn_ref.num_bands_for(F, bsrnn_ref.SUBBANDS_481)
        if F in (81, 161, 221, 241, 321, 481):
            assert got == K, (F, got)


def test_stft_oracle_matches_numpy_dft():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3000, generator=g, dtype=torch.float64)
    for n_fft, hop, win in ((320, 160, "hann"), (441, 220, "hann"), (256, 128, "rect")):
        X, _ = stft_ref.stft(x, n_fft, hop, win)
        Xn = stft_ref.stft_numpy(x.numpy(), n_fft, hop, win)
        assert np.abs(X.numpy() - Xn).max() < 1e-9
        y = stft_ref.istft(X, n_fft, hop, 3000, win).numpy()
        yn = stft_ref.istft_numpy(Xn, n_fft, hop, 3000, win)
        assert np.abs(y - yn).max() < 1e-9
    X, _ = stft_ref.stft(x, 320, 160)
    assert (stft_ref.istft(X, 320, 160, 3000) - x).abs().max() < 1e-9
    # frames past olens are zeroed
    X, ol = stft_ref.stft(x, 320, 160, "hann", torch.tensor([3000, 1000]))
    assert ol.tolist() == [19, 7] and torch.all(X[1, 7:] == 0) and X[1, 6].abs().sum() > 0


def test_golden_vectors_reproduce():
    from tests.golden.make_golden import SMALL, small_inputs, small_model
    g = np.load(os.path.join(GOLD, "oracle_small.npz"))
    clean, noisy, lens = small_inputs()
    assert np.array_equal(clean.numpy(), g["clean"]) and np.array_equal(noisy.numpy(), g["noisy"])
    m = small_model()
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    assert abs(flat.double().sum().item() - g["param_checksum"][0]) < 1e-6
    wav, spec = m(noisy, lens, SMALL["fs"])
    assert np.abs(wav.detach().numpy() - g["wav"]).max() <= 1e-5 * np.abs(g["wav"]).max()
    loss = losses_ref.mr_l1_loss(clean, wav)
    assert np.allclose(loss.detach().numpy(), g["loss"], rtol=1e-5)
    assert np.allclose(losses_ref.si_snr_loss(clean, wav.detach()).numpy(), g["sisnr"], atol=1e-4)


def test_emulated_bf16_is_close_to_f32():
    from tests.golden.make_golden import SMALL, small_inputs, small_model
    clean, noisy, lens = small_inputs()
    m = small_model()
    with torch.no_grad():
        a = m(noisy, lens, SMALL["fs"])[0]
        b = m(noisy, lens, SMALL["fs"], True)[0]
    assert (a - b).abs().max() < 2e-2 * a.abs().max()


def test_mrl1_matches_manual_formula():
    g = torch.Generator().manual_seed(3)
    t = torch.randn(2, 2000, generator=g)
    e = 0.5 * t + 0.1 * torch.randn(2, 2000, generator=g)
    ref = losses_ref.mr_l1_loss(t, e)
    tn, en = t / t.std(1, keepdim=True), e / e.std(1, keepdim=True)
    a = (en * tn).sum(1, keepdim=True) / ((en ** 2).sum(1, keepdim=True) + 1e-6)
    man = 0.5 * (a * en - tn).abs().sum(1)
    for w in (256, 512, 768, 1024):
        E = np.abs(stft_ref.stft_numpy((a * en).numpy(), w, w // 2, "rect"))
        T = np.abs(stft_ref.stft_numpy(tn.numpy(), w, w // 2, "rect"))
        man = man + 0.125 * torch.from_numpy(np.abs(E - T).sum((1, 2))).float()
    assert torch.allclose(ref, man, rtol=1e-4)


def _load_flow_golden():
    from oracle import flow_ref
    g = np.load(os.path.join(GOLD, "ref_flow.npz"))
    N = g["w:condition_fc.weight"].shape[0]
    L = sum(1 for k in g["keys"].tolist() if k.startswith("norm_time.") and k.endswith(".weight"))
    m = flow_ref.BSRNNFlow(769, N, L)
    m.load_state_dict({k: torch.from_numpy(g["w:" + k]) for k in g["keys"].tolist()})
    c = lambda a: torch.view_as_complex(torch.from_numpy(a))
    return g, m, c


def test_flow_oracle_matches_reference_goldens():
    """oracle/flow_ref.py vs vectors produced by the REFERENCE's own bsrnn_flowse.py / odes.py / sampling
    (tests/golden/make_golden_flow.py): DNN forward bitwise, ODE marginals, 4-step Euler trajectory."""
    from oracle import flow_ref
    g, m, c = _load_flow_golden()
    x, y, t = c(g["x"]), c(g["y"]), torch.from_numpy(g["t"])
    with torch.no_grad():
        out = m(torch.cat([x, y], 1), t)
    assert torch.equal(out, c(g["out"]))
    ode = flow_ref.FlowMatching(0.05, 0.5)
    mean, std = ode.marginal_prob(x, t, y)
    assert torch.equal(mean, c(g["mean"])) and torch.equal(std, torch.from_numpy(g["std"]))
    s = flow_ref.euler_sample(lambda xx, tt, yy: -m(torch.cat([xx, yy], 1), tt), ode, y, c(g["z"]), 1.0, 0.03, 4)
    assert torch.allclose(s, c(g["sample"]), atol=1e-6)
    full = flow_ref.BSRNNFlow(769, 384, 6)
    assert sum(p.numel() for p in full.parameters()) == int(g["n_params_full"]) == 103245488


def test_euler_schedule():
    from oracle import flow_ref
    ts, steps = flow_ref.euler_timesteps(1.0, 0.03, 15)
    assert abs(float(ts[0]) - 1.0) < 1e-7 and abs(float(ts[-1]) - 0.03) < 1e-7
    assert abs(steps[-1] - 0.03) < 1e-7 and abs(sum(steps) - 1.0) < 1e-6     # last step integrates down to t = 0


def test_mix_oracle_detect_non_silence_and_scipy_calls():
    """a20 oracle: the espnet detect_non_silence restatement behaves as SURVEY A.5 states; the rest is scipy / numpy."""
    import numpy as np
    from oracle import mix_ref
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 5000))
    x[:, :2048] *= 1e-3                                     # two silent frames' worth
    m = mix_ref.detect_non_silence(x)
    assert m.shape == x.shape and m.dtype == bool
    assert not m[0, :1024].any() and m[0, 3000:].all()
    assert mix_ref.detect_non_silence(x[:, :1000]).all()   # shorter than one frame: everything counts
    assert mix_ref.detect_non_silence(np.zeros((1, 4096))).all()
    # SNR of the mix equals the request when measured with the same power rule
    sp, nz = rng.standard_normal((1, 8000)), rng.standard_normal((1, 3000))
    noisy, noise = mix_ref.mix_noise(sp, nz, 7.0, offset=100)
    ps = (sp[mix_ref.detect_non_silence(sp)] ** 2).mean()
    pn = (noise[mix_ref.detect_non_silence(noise)] ** 2).mean()
    assert abs(10 * np.log10(ps / pn) - 7.0) < 1e-9 and np.allclose(noisy, sp + noise)
    assert noise.shape == sp.shape and np.allclose(noise[0, 100:3100] / noise[0, 100], nz[0] / nz[0, 0])
    assert len(mix_ref.filter_designs(48000)) == 1455 and len(mix_ref.filter_designs(8000)) == 243


def _twin_models(g, fold=False):
    """oracle modules loaded with the weights the reference twin ran with (tests/golden/make_golden_bsrnn.py)."""
    bs = bsrnn_ref.BandSplit(481, 48000, 16)
    bs.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("bsw:")}, strict=True)
    net = bsrnn_ref.BSRNN(481, 16, 2, 48000, False, 1)
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}
    if fold:
        sd.update({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("wfold:")})
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith(("band_split", "mask_decoder")) for m in missing)
    return bs, net


def test_bsrnn_oracle_equals_reference_twin():
    """ref_bsrnn.npz holds outputs of the REFERENCE'S OWN bsrnn_flowse.BandSplit(481) and BSRNN loop (bsrnn_flowse.py:16-86,
    288-307), bit-equal to this oracle where it was generated; here (another CPU / BLAS) the bound is 1e-6 relative."""
    g = np.load(os.path.join(GOLD, "ref_bsrnn.npz"))
    bs, net = _twin_models(g)
    with torch.no_grad():
        for fs in (48000, 22050, 16000, 8000):
            z = bs(torch.from_numpy(g["bs_x_%d" % fs])).numpy()
            ref = g["bs_z_%d" % fs]
            assert z.shape == ref.shape and np.abs(z - ref).max() <= 1e-6 * np.abs(ref).max(), fs
        skip = net.dual_path(torch.from_numpy(g["z"])).numpy()
        assert np.abs(skip - g["skip_zero_temb"]).max() <= 1e-6 * np.abs(g["skip_zero_temb"]).max()
        _, folded = _twin_models(g, fold=True)
        skip_t = folded.dual_path(torch.from_numpy(g["z"])).numpy()
        assert np.abs(skip_t - g["skip_folded_temb"]).max() <= 4e-6 * np.abs(g["skip_folded_temb"]).max()


def test_soxr_hq_specification_of_the_resampling_filter():
    """the 48 -> 16 kHz (and 44.1 / 32 / 22.05 -> 16) filter standing in for soxr HQ meets soxr HQ's own specification
    (soxr.c soxr_quality_spec quality 4): pass band to 0.9136 x Nyquist flat within 2^-20, stop band from the Nyquist frequency
    of the lower rate down >= 120.4 dB, linear phase (symmetric taps), unity DC gain.  That - not bit equality with libsoxr, which
    is not reproducible - is the tolerance the stand-in is held to."""
    from oracle import metrics_ref
    for fs_in, fs_out in ((48000, 16000), (32000, 16000), (48000, 8000), (22050, 16000)):
        h, up, down = metrics_ref.soxr_hq_design(fs_in, fs_out)
        assert np.allclose(h, h[::-1]) and len(h) % 2 == 1
        fsw, n, nyq = fs_in * up, len(h), 0.5 * min(fs_in, fs_out)
        t = np.arange(n) - (n - 1) / 2
        resp = lambda f: abs(np.sum(h * np.exp(-2j * np.pi * f / fsw * t)))
        assert abs(resp(0.0) - 1.0) <= 2.0 ** -20
        for f in np.linspace(0, 0.9136 * nyq, 60):
            assert abs(resp(f) - 1.0) <= 2.0 ** -20, (fs_in, f)
        for f in np.linspace(nyq, 0.5 * fsw, 200):
            assert 20 * np.log10(resp(f) + 1e-30) <= -120.4, (fs_in, f)
